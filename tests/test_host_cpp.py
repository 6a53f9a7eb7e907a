"""C++ host layer (include/poasta_amd.hpp + poasta_amd/host/lasagna_amd.cpp): the `lasagna align`-shaped
driver.  CPU: GFA import and node->segment resolution against the values the reference's tests assert.
GPU: GAF records (path, CIGAR, AS:i) against an independent restatement of `alignment_to_gaf` applied to
the oracle's A* alignments."""
import os
import subprocess

import numpy as np
import pytest

from poasta_amd.graph import GraphBuilder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "poasta_amd", "lasagna_amd")
GFA = os.path.join(ROOT, "tests", "golden", "test.gfa")


@pytest.fixture(scope="module")
def driver():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "poasta_amd", "csrc")], stdout=subprocess.DEVNULL)
    assert os.path.exists(DRIVER)
    return DRIVER


def _dump(driver, gfa):
    out = subprocess.check_output([driver, "align", "--dump-graph", gfa]).decode()
    nodes, segs = {}, []
    n = None
    for line in out.splitlines():
        f = line.split("\t")
        if f[0] == "nodes":
            n = int(f[1])
        elif f[0] == "segment":
            segs.append((f[1], int(f[2]), int(f[3]), int(f[4])))
        elif f[0] == "node":
            v = int(f[1])
            si, pi = f.index("succ"), f.index("pred")
            seg = f.index("seg") if "seg" in f else len(f)
            nodes[v] = dict(sym=f[2], succ=[int(x) for x in f[si + 1:pi]], pred=[int(x) for x in f[pi + 1:seg]],
                            seg=(int(f[seg + 1]), int(f[seg + 2])) if seg < len(f) else None)
    return n, segs, nodes


def _parse_gfa(path):
    segs, links = [], []
    for line in open(path):
        f = line.strip().split("\t")
        if f[0] == "S":
            segs.append((f[1], f[2].upper()))
        elif f[0] == "L":
            links.append((f[1], f[3]))
    return segs, links


def _python_graph(path):
    segs, links = _parse_gfa(path)
    b = GraphBuilder()
    ids = {}
    for name, seq in segs:
        ids[name] = b.add_path(np.frombuffer(seq.encode(), np.uint8))
    for a, c in links:
        b.add_edge(ids[a][-1], ids[c][0])
    return b.finish(), segs, ids


def test_gfa_import_and_segment_resolution(driver):
    """src/io/graph.rs:617-628: tests/test.gfa loads to 35 nodes; src/io/gaf.rs:313-355: resolver positions."""
    n, segs, nodes = _dump(driver, GFA)
    assert n - 2 == 35
    assert [s[0] for s in segs] == ["s1", "s2", "s3", "s4"] and [s[3] for s in segs] == [20, 8, 4, 3]
    s1_start, s1_end = segs[0][1], segs[0][2]
    assert nodes[s1_start]["seg"] == (0, 0)
    s1_second = [x for x in nodes[s1_start]["succ"] if x != 1][0]
    assert nodes[s1_second]["seg"] == (0, 1)
    assert nodes[s1_end]["seg"] == (0, 19)
    s2_start, s2_end = segs[1][1], segs[1][2]
    assert nodes[s2_start]["seg"] == (1, 0) and nodes[s2_end]["seg"] == (1, 7)
    assert nodes[segs[3][1]]["seg"] == (3, 0) and nodes[segs[3][2]]["seg"] == (3, 2)
    assert nodes[0]["seg"] is None and nodes[1]["seg"] is None
    # adjacency (iteration order!) equals the Python host-side builder's
    g, _, _ = _python_graph(GFA)
    for v in range(g.n):
        assert nodes[v]["succ"] == g.successors(v).tolist() and nodes[v]["pred"] == g.predecessors(v).tolist()
        assert ord(nodes[v]["sym"]) == g.symbol[v]


# ---- independent restatement of alignment_to_gaf (src/io/gaf.rs:152-304) for the check -----------------
def _gaf_line(g, segs, ids, name, seq, aln, score, NONE=0xFFFFFFFF):
    node_seg = {}
    for si, (nm, _) in enumerate(segs):
        for pos, v in enumerate(ids[nm]):
            node_seg[v] = (si, pos)
    if not aln:
        return None
    qstart, pstart, path, ops = 0, 0, [], []
    at_start, last_ix, last_pos, nmatch = True, 0, 0, 0
    eq = lambda r, q: r == g.end or g.symbol[r] == seq[q]
    for r, q in aln:
        if at_start:
            if r != NONE and q == NONE:
                qstart += 1
            elif r != NONE and q != NONE:
                si, pos = node_seg[r]
                pstart = pos
                path.append(si)
                m = eq(r, q); nmatch += m; ops.append("=" if m else "X")
                at_start = False
                last_ix, last_pos = len(path) - 1, pos
        elif r != NONE and q != NONE:
            si, pos = node_seg[r]
            if path[-1] != si:
                path.append(si)
            m = eq(r, q); nmatch += m; ops.append("=" if m else "X")
            last_ix, last_pos = len(path) - 1, pos
        elif r != NONE:
            si, _ = node_seg[r]
            if path[-1] != si:
                path.append(si)
            ops.append("D")
        else:
            ops.append("I")
    gpath = "".join(">" + segs[s][0] for s in path[:last_ix + 1])
    plen = sum(len(segs[s][1]) for s in path[:last_ix + 1])
    pend = plen - len(segs[path[last_ix]][1]) + last_pos
    qend = [q for r, q in aln if r != NONE and q != NONE][-1]
    rle = []
    for o in ops:
        if rle and rle[-1][0] == o:
            rle[-1][1] += 1
        else:
            rle.append([o, 1])
    if rle and rle[-1][0] in "ID":
        rle.pop()
    block = sum(c for _, c in rle)
    cigar = "".join("%d%s" % (c, o) for o, c in rle)
    return "\t".join(str(x) for x in [name, len(seq), qstart, qend, "+", gpath, plen, pstart, pend, nmatch, block, 60,
                                      "cg:Z:" + cigar, "AS:i:%d" % score])


@pytest.mark.gpu
def test_lasagna_amd_gaf_matches_restated_reference(driver, oracle, tmp_path):
    g, segs, ids = _python_graph(GFA)
    s1, s2, s3, s4 = (s[1] for s in segs)
    reads = [("r_path134", s1 + s2 + s4), ("r_path1234", s1 + s2 + s3 + s4), ("r_sub", s1[:9] + "T" + s1[10:] + s2 + s4),
             ("r_del", s1[:5] + s1[8:] + s2 + s3 + s4), ("r_ins", s1 + "ACGT" + s2 + s4), ("r_short", s2 + s3),
             ("r_tail", s1 + s2 + s4 + "GG")]
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(">%s\n%s\n" % r for r in reads))
    out = subprocess.check_output([driver, "align", "--mode", "exact", GFA, str(fa)]).decode().strip().splitlines()
    og = oracle.OracleGraph.from_csr(g.as_dict())
    want = []
    for name, seq in reads:
        a = og.astar_align(seq.encode(), oracle.Costs(4, 6, 2), oracle.H_MINGAP, True)
        line = _gaf_line(g, segs, ids, name, seq.encode(), a["alignment"], a["score"])
        if line is not None:
            want.append(line)
    assert out == want
    # dense mode: same scores (AS:i) for every read
    dense = subprocess.check_output([driver, "align", GFA, str(fa)]).decode().strip().splitlines()
    assert [l.split("\t")[-1] for l in dense] == [l.split("\t")[-1] for l in want]


# ------------------------------------------------------------------------------------------------
# Graph update + MSA import / export of the C++ host mirror: POAGraph::add_alignment_with_weights
# (src/graphs/poa.rs:171-321), load_graph_from_fasta_msa (src/io/graph.rs:36-103), poa_graph_to_fasta (src/io/fasta.rs:69-156).
ALIGN_DRIVER = os.path.join(ROOT, "poasta_amd", "poasta_align_amd")
GOLD = os.path.join(ROOT, "tests", "golden")


def _replay(driver_dir, tmp_path, alignments, msa=None):
    path = str(tmp_path / "replay.txt")
    with open(path, "w") as f:
        if msa:
            f.write("msa\t%d\n" % len(msa))
            for n, r in msa:
                f.write("%s\t%s\n" % (n, r))
        for name, seq, aln in alignments:
            if aln is None:
                f.write("%s\t%s\t-\n" % (name, seq))
            else:
                f.write("%s\t%s\t%d\n" % (name, seq, len(aln)))
                for r, q in aln:
                    f.write("%d %d\n" % (-1 if r == 0xFFFFFFFF else r, -1 if q == 0xFFFFFFFF else q))
    return subprocess.check_output([ALIGN_DRIVER, "replay", path]).decode()


def test_cpp_graph_update_and_export_kats(driver, tmp_path):
    """The reference's own asserted strings (tests/io_fasta.rs:4-34, src/io/fasta.rs:165-209) through the C++ mirror."""
    assert _replay(driver, tmp_path, [("seq1", "AC", None), ("seq2", "ACGT", None)]) == ">seq1\n---AC\n>seq2\nACGT--\n"
    assert _replay(driver, tmp_path, [("empty", "", None)]) == ">empty\n"
    N = 0xFFFFFFFF
    assert _replay(driver, tmp_path, [("seq1", "ACG", None), ("seq2", "AG", [(2, 0), (3, N), (4, 1)])]) == ">seq1\nACG\n>seq2\nA-G\n"


def test_cpp_graph_update_matches_oracle_on_fixture_builds(driver, oracle, tmp_path):
    """Sequential POA builds of the reference's fixtures: the alignments come from the oracle, the graph update and the MSA
    export run in the C++ mirror; the MSA must equal the oracle's own (same alignments, independent restatement)."""
    for fa in ("test_from_abpoa.fa", "test2_from_abpoa.fa"):
        alns = []
        g, _ = oracle.sequential_poa(oracle.read_fasta(os.path.join(GOLD, fa)), oracle.Costs(4, 6, 2), alignments=alns)
        assert _replay(driver, tmp_path, alns) == g.to_fasta()
    msa = oracle.read_fasta(os.path.join(GOLD, "small_test.input.fa"))
    alns = []
    g = oracle.OracleGraph.from_fasta_msa(msa)
    g, _ = oracle.sequential_poa(oracle.read_fasta(os.path.join(GOLD, "small_test.query.fa")), oracle.Costs(4, 6, 2), graph=g, alignments=alns)
    out = _replay(driver, tmp_path, alns, msa=msa)
    assert out == g.to_fasta()
    assert out.splitlines()[5] == "-----TTGTCAACATCAGTA"   # small_test.truth.fa row 3 up to the export's leading-gap quirk
    # random reads against a growing graph: co-optimal tie-breaks, SNP columns (aligned_nodes), insertions
    rng = np.random.Generator(np.random.PCG64(11))
    base = rng.integers(0, 4, size=60)
    recs = []
    for i in range(8):
        s = base.copy()
        for _ in range(4):
            p = int(rng.integers(0, len(s)))
            k = int(rng.integers(0, 3))
            s = np.concatenate([s[:p], [int(rng.integers(0, 4))], s[p + (k != 1):]]) if k else np.delete(s, p)
        recs.append(("r%d" % i, "".join("ACGT"[x] for x in s)))
    alns = []
    g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2), alignments=alns)
    assert _replay(driver, tmp_path, alns) == g.to_fasta()


@pytest.mark.gpu
def test_gpu_sequential_poa_build_equals_oracle_msa(driver, oracle, tmp_path):
    """BASELINE.json configs[0] end to end on the engine: `poasta align`-shaped driver, every read aligned on the GPU in
    hybrid mode (the reference's own tie-breaks), graph updated on the host, MSA exported — byte-identical to the MSA the
    oracle's sequential build gives (tests/test_oracle_msa.py says what that MSA is pinned by)."""
    for fa in ("test_from_abpoa.fa", "test2_from_abpoa.fa"):
        g, _ = oracle.sequential_poa(oracle.read_fasta(os.path.join(GOLD, fa)), oracle.Costs(4, 6, 2))
        out = subprocess.check_output([ALIGN_DRIVER, "align", os.path.join(GOLD, fa)], stderr=subprocess.DEVNULL).decode()
        assert out == g.to_fasta(), fa
    g = oracle.OracleGraph.from_fasta_msa(oracle.read_fasta(os.path.join(GOLD, "small_test.input.fa")))
    g, _ = oracle.sequential_poa(oracle.read_fasta(os.path.join(GOLD, "small_test.query.fa")), oracle.Costs(4, 6, 2), graph=g)
    out = subprocess.check_output([ALIGN_DRIVER, "align", "-I", os.path.join(GOLD, "small_test.input.fa"),
                                   os.path.join(GOLD, "small_test.query.fa")], stderr=subprocess.DEVNULL).decode()
    assert out == g.to_fasta()
    truth = [l for l in open(os.path.join(GOLD, "small_test.truth.fa")).read().splitlines() if not l.startswith(">")]
    rows = [l for l in out.splitlines() if not l.startswith(">")]
    assert rows[:2] == truth[:2] and rows[2] == truth[2][1:]   # the reference-held MSA, up to the export's leading-gap quirk


@pytest.mark.gpu
def test_gpu_sequential_poa_cli_options_two_piece_heuristic_span(driver, oracle, tmp_path):
    """The CLI's other aligner selections (src/bin/poasta.rs:275-445) on the engine: `-g 6,24 -e 2,1` (Affine2PieceMinGapCost:
    every read through the replay of the five-state search), `-H dijkstra`, `-m ends-free`, and the fallback to the one-piece
    model when extend1 <= extend2 — MSAs byte-identical to the oracle's sequential builds under the same configuration."""
    fa = os.path.join(GOLD, "test2_from_abpoa.fa")
    recs = oracle.read_fasta(fa)
    run = lambda *opts: subprocess.check_output([ALIGN_DRIVER, "align", *opts, fa], stderr=subprocess.DEVNULL).decode()
    with oracle.two_piece(24, 1):
        g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2))
    assert run("-g", "6,24", "-e", "2,1") == g.to_fasta()
    with oracle.two_piece(24, 1):
        g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2), heuristic=oracle.H_DIJKSTRA)
    assert run("-g", "6,24", "-e", "2,1", "-H", "dijkstra") == g.to_fasta()
    g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2), heuristic=oracle.H_DIJKSTRA)
    assert run("-H", "dijkstra") == g.to_fasta()
    g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2))
    assert run("-g", "6,24", "-e", "2,2") == g.to_fasta()          # poasta.rs:339-342: standard affine with the first values
    ef = oracle.ends_free(oracle.UNBOUNDED, oracle.UNBOUNDED, oracle.UNBOUNDED, oracle.UNBOUNDED)
    for two in (False, True):
        opts = ("-m", "ends-free") + (("-g", "6,24", "-e", "2,1") if two else ())
        try:
            with oracle.alignment_type(ef):
                if two:
                    with oracle.two_piece(24, 1):
                        g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2))
                else:
                    g, _ = oracle.sequential_poa(recs, oracle.Costs(4, 6, 2))
        except (oracle.RefPanic, RuntimeError):
            # the reference's build stops here (a panic, or add_alignment_with_weights returns InvalidAlignment for what an
            # all-Unbounded search hands it, poa.rs:171-180): so does the twin
            assert subprocess.run([ALIGN_DRIVER, "align", *opts, fa], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL).returncode != 0
            continue
        assert run(*opts) == g.to_fasta(), opts
    assert subprocess.run([ALIGN_DRIVER, "align", "-H", "path", fa], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL).returncode == 1

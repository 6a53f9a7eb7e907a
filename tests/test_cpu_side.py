"""CPU-only tests: oracle self-consistency (dense restatement == literal A* restatement), host logic,
the C ABI's symbol table and error behaviour without a GPU, and the world_size-2 gloo rehearsal of
the multi-GPU sharding + gather."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

from poasta_amd import workloads as W
from poasta_amd.graph import FlatGraph, GraphBuilder, pack_queries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_UNCERTAIN = 2 | 8  # START_QUIRK | SHORT_QUERY


# ------------------------------------------------------------------------------------------------
# oracle: the dense recurrences (what the GPU computes) against the literal A* restatement
def _compare_dense_astar(oracle, g, qs, costs):
    og = oracle.OracleGraph.from_csr(g.as_dict())
    qseq, qoff = pack_queries(qs)
    oc = oracle.Costs(*costs)
    A = og.astar_batch(qseq, qoff, oc, oracle.H_MINGAP, True, threads=4)
    Dj = og.astar_batch(qseq, qoff, oc, oracle.H_DIJKSTRA, False, threads=4)
    D = og.dense_batch(qseq, qoff, oc, threads=4)
    stats = dict(n=0, certified=0, equal=0)
    for i in range(len(qs)):
        if A["status"][i] != 0:  # the reference would panic (u32 wrap) — nothing to compare
            continue
        f = int(D["flags"][i])
        stats["n"] += 1
        if not f & SCORE_UNCERTAIN:
            assert int(D["score"][i]) == int(A["score"][i]), "score, query %d" % i
            if Dj["status"][i] == 0:
                assert int(D["score"][i]) == int(Dj["score"][i])
        same = oracle.batch_alignment(A, i) == oracle.batch_alignment(D, i)
        stats["equal"] += same
        if f == 0:
            stats["certified"] += 1
            assert same, "certified-unique alignment differs from the A* restatement, query %d" % i
    return stats


def test_dense_equals_astar_random_dags(oracle):
    tot = dict(n=0, certified=0, equal=0)
    for seed in range(120):
        rng = np.random.Generator(np.random.PCG64(1000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(20)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (3, 1, 1), (4, 4, 2)][seed % 5]
        st = _compare_dense_astar(oracle, g, qs, costs)
        for k in tot:
            tot[k] += st[k]
    assert tot["n"] > 2000 and tot["certified"] > 1000


def test_dense_equals_astar_linearish(oracle):
    g, (qseq, qoff) = W.scaled_linearish(300, 15, 8, 60, 330)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(60)]
    st = _compare_dense_astar(oracle, g, qs, (4, 6, 2))
    assert st["n"] == 60


def test_dense_equals_astar_bubbles_and_msa(oracle):
    poa = W.LayeredPOA(n_layers=40, width=4, indeg=4, seed=5)
    _compare_dense_astar(oracle, poa.graph, poa.queries(20, length=0), (4, 6, 2))
    pg = W.PangenomePOA(ref_len=300, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
    _compare_dense_astar(oracle, pg.graph, pg.queries(12, length=120), (4, 6, 2))


def test_sequential_poa_build_fixture(oracle):
    """BASELINE.json configs[0]: the reference's fixture reads (tests/test_from_abpoa.fa, 4 reads; the
    10-read file is tests/test2_from_abpoa.fa), POA built from scratch through the restated
    add_alignment_with_weights (graphs/poa.rs:171-321) with every read aligned by the A* restatement.
    Checks the plumbing invariant (every read is spelled by a path from its recorded start node); the
    *.truth.fa files are unasserted by the reference's own tests."""
    for name in ("test_from_abpoa", "test2_from_abpoa"):
        reads = _read_fasta(os.path.join(ROOT, "tests", "golden", name + ".fa"))
        g = oracle.OracleGraph.new_poa()
        for i, (nm, seq) in enumerate(reads):
            if g.n == 2:
                g.add_alignment(nm, seq, None)
            else:
                r = g.astar_align(seq, oracle.Costs(4, 6, 2), oracle.H_MINGAP)
                d = g.dense_align(seq, oracle.Costs(4, 6, 2))
                assert d["score"] == r["score"] or d["flags"] & SCORE_UNCERTAIN
                g.add_alignment(nm, seq, r["alignment"])
        csr = g.export_csr()
        fg = FlatGraph.from_dict(csr)
        starts = g.seq_start_nodes()
        assert len(starts) == len(reads)
        for (nm, seq), st in zip(reads, starts):
            assert _spells_path(fg, seq, st), nm


def _read_fasta(path):
    out, name, buf = [], None, []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(buf).encode()))
            name, buf = line[1:], []
        elif line:
            buf.append(line)
    if name is not None:
        out.append((name, "".join(buf).encode()))
    return out


def _spells_path(g, seq, first_node):
    if g.symbol[first_node] != seq[0]:
        return False
    cur = {int(first_node)}
    for c in seq[1:]:
        nxt = set()
        for v in cur:
            for s in g.successors(v):
                if s != g.end and g.symbol[s] == c:
                    nxt.add(int(s))
        if not nxt:
            return False
        cur = nxt
    return True


# ------------------------------------------------------------------------------------------------
# host logic
def test_graph_builder_matches_oracle_poa(oracle):
    """GraphBuilder (product side) reproduces the adjacency order of the restated POAGraph."""
    seqs = [b"ACGTACGT", b"ACGAACGT", b"TTACG"]
    og = oracle.OracleGraph.new_poa()
    b = GraphBuilder()
    for i, s in enumerate(seqs):
        og.add_alignment("s%d" % i, s, None)
        b.add_path(np.frombuffer(s, np.uint8))
        b.finish()
    g = b.finish()
    csr = og.export_csr()
    for k in ("symbol", "succ_off", "succ", "pred_off", "pred"):
        assert np.array_equal(getattr(g, k), csr[k]), k


def test_workloads_are_deterministic():
    g1, (q1, o1) = W.config2(n_queries=8)
    g2, (q2, o2) = W.config2(n_queries=8)
    assert np.array_equal(q1, q2) and np.array_equal(o1, o2) and np.array_equal(g1.succ, g2.succ)
    assert g1.n == 1002 and all(int(o1[i + 1] - o1[i]) == 1000 for i in range(8))
    # shards of the seeded stream agree with the whole
    _, (q3, o3) = W.config2(n_queries=4, first=4)
    assert np.array_equal(q3, q1[int(o1[4]):])
    # in-degree of the config-5 family
    poa = W.LayeredPOA(n_layers=50, width=4, indeg=4)
    g = poa.graph
    indeg = np.diff(g.pred_off)[2:]
    assert indeg[4:].mean() >= 4.0


def test_msa_import_rule():
    """src/io/graph.rs:36-103: one node per (column, symbol), '-' skipped; tests/io_fasta.rs:4-34 rows."""
    g = W.msa_to_graph([b"---AC", b"ACGT--"[:5]])
    assert g.n == 2 + 2 + 4  # AC in cols 3,4 and ACGT in cols 0..3 share nothing -> 6 real nodes
    g = W.msa_to_graph([b"ACGT", b"AC-T", b"AGGT"])
    assert g.n == 2 + 5  # A, C|G, G, T


# ------------------------------------------------------------------------------------------------
# C ABI: loads, exports every declared symbol, fails loudly without a device
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "poasta_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(poa_[a-z0-9_]+)\s*\(", hdr)))


def test_abi_exports_every_declared_symbol():
    from poasta_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 15
    for s in declared:
        assert hasattr(L, s), "libpoasta_amd.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == declared
    assert b"gfx950" in _lib.lib().poa_version()


def test_abi_graph_validation_and_no_device_error():
    from poasta_amd import _lib, aligner
    g, (qseq, qoff) = W.scaled_linearish(30, 2, 1, 2, 30)
    dg = aligner.DeviceGraph(g)
    assert _lib.lib().poa_graph_rows(dg.handle) == g.n
    rows = dg.node_rows()
    assert sorted(rows.tolist()) == list(range(g.n)) and rows[g.start] == 0 and rows[g.end] == g.n - 1
    for v in range(g.n):  # topological
        for s in g.successors(v):
            assert rows[v] < rows[s]
    # malformed graph: predecessor lists do not mirror successor lists
    bad = FlatGraph(g.n, g.start, g.end, g.symbol, g.succ_off, g.succ, g.pred_off, g.pred[::-1].copy())
    with pytest.raises(_lib.PoaError):
        aligner.DeviceGraph(bad)
    if _lib.lib().poa_device_count() == 0:
        # no GPU here: the product path must refuse, never compute on the CPU
        al = aligner.PoastaAligner(aligner.AffineMinGapCost(aligner.GapAffine(4, 2, 6)))
        with pytest.raises(_lib.PoaError) as ei:
            al.align_batch(g, qseq=qseq, qoff=qoff)
        assert ei.value.code == -3  # POA_ERR_NO_DEVICE
        # the two-piece entry points refuse alike, dense and replayed (no CPU search behind them either)
        for mode in ("dense", "exact"):
            a2 = aligner.PoastaAligner(aligner.Affine2PieceMinGapCost(aligner.GapAffine2Piece(4, 2, 6, 1, 24)), mode=mode)
            with pytest.raises(_lib.PoaError) as ei:
                a2.align_batch(g, qseq=qseq, qoff=qoff)
            assert ei.value.code == -3
    # argument checks of the two-piece entry point come before any device work: ends-free needs the replay
    a3 = aligner.PoastaAligner(aligner.Affine2PieceMinGapCost(aligner.GapAffine2Piece(4, 2, 6, 1, 24)), aligner.AlignmentType.EndsFree(), mode="dense")
    with pytest.raises(ValueError):
        a3.align_batch(g, qseq=qseq, qoff=qoff)
    with pytest.raises(ValueError):
        aligner.GapAffine2Piece(4, 1, 6, 2, 24)   # extend1 < extend2: the reference's constructor panics (gap_affine_2piece.rs:28-33)


def test_product_path_does_not_touch_the_oracle():
    """poasta_amd/ and bench.py's GPU leg must not import, link or execute anything under oracle/."""
    bad = re.compile(r"(from\s+oracle|import\s+oracle|oracle/|libpoa_oracle|pyoracle)")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "poasta_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(src), "%s/%s references the oracle" % (dirpath, f)
    assert not bad.search(open(os.path.join(ROOT, "include", "poasta_amd.h")).read())
    # bench.py may use the oracle only inside cpu_baseline()
    b = open(os.path.join(ROOT, "bench.py")).read()
    head, tail = b.split("def cpu_baseline", 1)
    assert not bad.search(head)


# ------------------------------------------------------------------------------------------------
# multi-GPU rehearsal on CPU: world_size 2, gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_worker(rank, world, port, n_total, out_path):
    import torch.distributed as dist
    from oracle import pyoracle
    from poasta_amd import dist as pdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    poa = W.LinearishPOA(60, 5, 3, seed=1)
    first, count = pdist.shard_range(n_total, rank, world)
    qs = poa.queries(count, length=70, seed=2, first=first)
    qseq, qoff = pack_queries(qs)
    og = pyoracle.OracleGraph.from_csr(poa.graph.as_dict())
    D = og.dense_batch(qseq, qoff, pyoracle.Costs(4, 6, 2))  # stands in for the rank's GPU results
    npairs = D["n_pairs"].astype(np.uint64)
    off = np.zeros(count + 1, np.uint64)
    off[1:] = np.cumsum(npairs)
    pairs = np.concatenate([D["pairs"][int(D["pair_off"][i]):int(D["pair_off"][i]) + int(npairs[i])] for i in range(count)]) \
        if count else np.zeros((0, 2), np.uint32)
    gs, gf, goff, gp = pdist.gather_results(D["score"], D["flags"], off, pairs)
    if rank == 0:
        np.savez(out_path, score=gs, flags=gf, off=goff, pairs=gp)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_shard_and_gather(oracle, tmp_path):
    import torch.multiprocessing as mp
    from poasta_amd import dist as pdist
    assert pdist.shard_range(10, 0, 4) == (0, 3) and pdist.shard_range(10, 3, 4) == (9, 1) and pdist.shard_range(2, 3, 4) == (2, 0)
    n_total = 11  # uneven shards: 6 + 5
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_dist_worker, args=(2, _free_port(), n_total, out), nprocs=2, join=True)
    z = np.load(out)
    poa = W.LinearishPOA(60, 5, 3, seed=1)
    qs = poa.queries(n_total, length=70, seed=2)
    qseq, qoff = pack_queries(qs)
    og = oracle.OracleGraph.from_csr(poa.graph.as_dict())
    D = og.dense_batch(qseq, qoff, oracle.Costs(4, 6, 2))
    assert np.array_equal(z["score"], D["score"]) and np.array_equal(z["flags"], D["flags"])
    for i in range(n_total):
        got = [tuple(x) for x in z["pairs"][int(z["off"][i]):int(z["off"][i + 1])].tolist()]
        assert got == oracle.batch_alignment(D, i)


def test_mingap_with_pruning_can_be_suboptimal(oracle):
    """Found while testing (parity unpinned: no reference fixture covers it, the reference cannot be built here):
    restated literally, the reference's default search (min-gap heuristic + superbubble pruning) returns 501 where
    Dijkstra order, or the same heuristic without pruning, or the dense recurrences return 500 — costs 8/3/1 on a
    20 %-divergent read.  Query 9 differs through the offset-0 DFA quirk instead (dense flag START_QUIRK).  With
    the CLI default costs 4/6/2 no such case appeared in 10 000 config-2 queries."""
    g, (qseq, qoff) = W.scaled_linearish(420, 20, 10, 12, 400, p_sub=0.2, p_ins=0.05, p_del=0.05)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    oc = oracle.Costs(8, 3, 1)
    D = og.dense_batch(qseq, qoff, oc, threads=4)
    dij = og.astar_batch(qseq, qoff, oc, oracle.H_DIJKSTRA, False, threads=4)
    dij_p = og.astar_batch(qseq, qoff, oc, oracle.H_DIJKSTRA, True, threads=4)
    mg = og.astar_batch(qseq, qoff, oc, oracle.H_MINGAP, False, threads=4)
    mg_p = og.astar_batch(qseq, qoff, oc, oracle.H_MINGAP, True, threads=4)
    ok = [i for i in range(12) if dij["status"][i] == 0]
    assert len(ok) == 10
    for i in ok:
        assert int(D["score"][i]) == int(dij["score"][i]) == int(dij_p["score"][i])
    assert int(D["score"][4]) == 500 and int(mg["score"][4]) == 500 and int(mg_p["score"][4]) == 501
    assert int(D["score"][9]) == 488 and int(mg["score"][9]) == 489 and int(D["flags"][9]) & 2
    assert all(int(mg_p["score"][i]) == int(D["score"][i]) for i in ok if i not in (4, 9))


def test_bench_spawns_its_own_launcher_for_n_gpus():
    """`python bench.py --gpus 2` with no launcher in the environment starts torch.distributed.run itself (as a child, before
    anything touches a GPU) and hands back its exit code.  Here, without a GPU, both ranks must get as far as the engine's
    "needs a GPU" exit — which proves the spawn, the rendezvous arguments and that rank discovery reads the launcher's env."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode(errors="replace")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the rehearsal itself runs (covered on the GPU box)")
    assert p.returncode != 0
    assert err.count("bench.py needs a GPU") >= 1, err[-2000:]
    assert "--nproc-per-node" not in err or "error: unrecognized" not in err

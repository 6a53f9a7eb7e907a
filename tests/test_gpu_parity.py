"""Parity of the HIP path (through the C ABI) against the oracle.  Bit-exact: integer scores,
(rpos, qpos) pairs, flags and — for small cases — every cell of the M/I/D planes."""
import numpy as np
import pytest

from poasta_amd import workloads as W
from poasta_amd.graph import GraphBuilder, pack_queries

pytestmark = pytest.mark.gpu

SCORE_UNCERTAIN = 2 | 8  # POA_FLAG_START_QUIRK | POA_FLAG_SHORT_QUERY


def _costs(engine, m=4, o=6, e=2):
    return engine.GapAffine(m, e, o)  # reference ctor order: (mismatch, extend, open)


def _linear_graph(seq):
    b = GraphBuilder()
    b.add_path(np.frombuffer(seq, np.uint8))
    return b.finish()


def _check_against_dense(engine, oracle, g, qs, costs=(4, 6, 2), planes=False):
    og = oracle.OracleGraph.from_csr(g.as_dict())
    oc = oracle.Costs(*costs)
    qseq, qoff = pack_queries(qs)
    al = engine.PoastaAligner(engine.AffineMinGapCost(_costs(engine, *costs)))
    if planes:
        rb = engine.ResidentBatch(g, qseq, qoff)
        rb.run(_costs(engine, *costs), None, engine.make_config(full_planes=True))  # all three planes kept for the comparison
        res = rb.fetch()
        node_rows = rb.dg.node_rows()
        orank = og.export_csr()["rank"]
    else:
        res = al.align_batch(g, qseq=qseq, qoff=qoff)
    D = og.dense_batch(qseq, qoff, oc, threads=4)
    for i in range(len(qs)):
        assert int(res.score[i]) == int(D["score"][i]), "score of query %d" % i
        assert res.raw_alignment(i) == oracle.batch_alignment(D, i), "alignment of query %d" % i
        assert int(res.flags[i]) == int(D["flags"][i]), "flags of query %d" % i
    if planes:
        for i in range(len(qs)):
            m, ii, d = rb.planes(i)
            od = og.dense_align(qs[i], oc, planes=True)
            for name, gp, op in (("M", m, od["M"]), ("I", ii, od["I"]), ("D", d, od["D"])):
                # rows are ranks; map node -> row on both sides
                assert np.array_equal(gp[node_rows], op[orank]), "plane %s of query %d" % (name, i)
        rb.close()
    return res, D


def _check_against_astar(oracle, g, qs, res, costs=(4, 6, 2), heuristic=None, pruning=True):
    og = oracle.OracleGraph.from_csr(g.as_dict())
    qseq, qoff = pack_queries(qs)
    A = og.astar_batch(qseq, qoff, oracle.Costs(*costs), oracle.H_MINGAP if heuristic is None else heuristic, pruning, threads=4)
    n_certified = 0
    for i in range(len(qs)):
        if A["status"][i] != 0:
            continue
        f = int(res.flags[i])
        if not f & SCORE_UNCERTAIN:
            assert int(res.score[i]) == int(A["score"][i]), "score vs A* of query %d" % i
        if f == 0:
            n_certified += 1
            assert res.raw_alignment(i) == oracle.batch_alignment(A, i), "certified alignment vs A* of query %d" % i
    return n_certified


def test_hand_traced_known_answers(engine, oracle):
    """SURVEY.md appendix C (hand-traced from the reference source)."""
    cases = [(b"ACGT", b"ACGT", (4, 6, 2), 0), (b"ACGT", b"AC", (1, 10, 2), 14), (b"AAAC", b"AAC", (4, 6, 2), 8),
             (b"AC", b"AAC", (4, 6, 2), 8), (b"AAAA", b"TTTT", (2, 8, 1), 8)]
    for gs, q, c, score in cases:
        g = _linear_graph(gs)
        res, _ = _check_against_dense(engine, oracle, g, [np.frombuffer(q, np.uint8)], c, planes=True)
        assert int(res.score[0]) == score
        _check_against_astar(oracle, g, [np.frombuffer(q, np.uint8)], res, c)
    # AAAC x AAC: the deletion is right-aligned (3rd A): nodes 2,3,4,5 = A,A,A,C
    g = _linear_graph(b"AAAC")
    al = engine.PoastaAligner(engine.AffineMinGapCost(_costs(engine)))
    r = al.align(g, b"AAC")
    assert r.pairs() == [(2, 0), (3, 1), (4, None), (5, 2)] and r.flags == 0


def test_planes_small_linearish(engine, oracle):
    g, (qseq, qoff) = W.scaled_linearish(60, 5, 3, 12, 70)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(12)]
    res, _ = _check_against_dense(engine, oracle, g, qs, planes=True)
    _check_against_astar(oracle, g, qs, res)


def test_random_dags(engine, oracle):
    total_cert = 0
    for seed in range(40):
        rng = np.random.Generator(np.random.PCG64(1000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(16)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (3, 1, 1), (4, 4, 2)][seed % 5]
        res, _ = _check_against_dense(engine, oracle, g, qs, costs, planes=(seed < 8))
        total_cert += _check_against_astar(oracle, g, qs, res, costs)
    assert total_cert > 100


def test_edge_cases(engine, oracle):
    g = _linear_graph(b"ACGTACGTAC")
    qs = [np.zeros(0, np.uint8), np.frombuffer(b"A", np.uint8), np.frombuffer(b"T", np.uint8),
          np.frombuffer(b"ACGTACGTAC" * 4, np.uint8), np.frombuffer(b"AC", np.uint8),
          np.frombuffer(b"GGGGGGGGGGGGGGGG", np.uint8), np.frombuffer(b"ACGTACGTAC", np.uint8)]
    res, _ = _check_against_dense(engine, oracle, g, qs, planes=True)
    _check_against_astar(oracle, g, qs, res)
    assert int(res.score[6]) == 0 and int(res.flags[6]) == 0
    # empty batch
    al = engine.PoastaAligner(engine.AffineMinGapCost(_costs(engine)))
    r = al.align_batch(g, [])
    assert len(r) == 0
    # empty graph: PoastaAligner::align shortcut (mod.rs:124-142): score 4 * len, no pairs
    r = al.align_batch(GraphBuilder().finish(), [b"ACGT", b""])
    assert r.score.tolist() == [16, 0] and r.alignment(0) == []


def test_multi_strip_long_queries(engine, oracle):
    """Queries longer than one 1024-column strip exercise the strip carry."""
    g, (qseq, qoff) = W.scaled_linearish(1500, 40, 20, 4, 0)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(4)]
    qs.append(qs[0][:1023])   # pitch exactly 1024
    qs.append(qs[1][:1024])   # pitch 1056: second strip with a single active lane group
    qs.append(np.concatenate([qs[2], qs[2][:700]]))  # 3 strips
    res, _ = _check_against_dense(engine, oracle, g, qs, planes=False)
    _check_against_astar(oracle, g, qs, res)
    rb_res, _ = _check_against_dense(engine, oracle, g, qs[4:6], planes=True)


def test_one_strip_kernel_shapes(engine, oracle):
    """The pairs-across-quads kernel (one strip, 512 < widest pitch <= 1024) on a batch of mixed lengths: queries whose
    second quad is partly or wholly outside their row, short ones whose first quad is too; with both flag encodings
    (scores below 0x3FFF: two flags ride in the M value; larger gap costs: bit-planes only)."""
    g, (qseq, qoff) = W.scaled_linearish(880, 40, 20, 6, 1000, p_sub=0.04, p_ins=0.02, p_del=0.02)
    full = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(6)]
    qs = [full[0], full[1][:513], full[2][:600], full[3][:777], full[4][:900], full[5][:100], full[0][:1023], full[1][:64], full[2][:960]]
    for costs in ((4, 6, 2), (9, 40, 12), (3, 1, 1)):
        res, _ = _check_against_dense(engine, oracle, g, qs, costs=costs)
        _check_against_astar(oracle, g, qs, res, costs=costs, heuristic=oracle.H_DIJKSTRA, pruning=False)
    poa = W.LayeredPOA(n_layers=800, width=4, indeg=4, seed=3)     # every row reads several predecessors from the planes
    qs = poa.queries(5, length=0) + [q[:600] for q in poa.queries(2, length=0, seed=8)]
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs)
    _check_against_astar(oracle, poa.graph, qs, res)


def test_symbols_beyond_acgt(engine, oracle):
    """Graph and reads over A C G T N a c g t (and a few IUPAC codes): the one-strip kernel fetches the masks of A/C/G/T
    rows from its LDS tables and computes the others; symbols are compared as bytes, exactly as the reference does."""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGTNacgtRYKM", np.uint8)
    probs = np.array([6, 6, 6, 6, 2, 1, 1, 1, 1, .5, .5, .5, .5]); probs = probs / probs.sum()
    backbone = rng.choice(alpha, 820, p=probs)
    b = GraphBuilder()
    ids = b.add_path(backbone)
    for _ in range(40):                                   # SNP-like side branches and skip edges
        i = int(rng.integers(1, 800))
        v = b.add_node(int(rng.choice(alpha, p=probs)))
        b.add_edge(ids[i - 1], v); b.add_edge(v, ids[i + 1])
        j = int(rng.integers(1, 790))
        b.add_edge(ids[j], ids[j + int(rng.integers(2, 6))])
    g = b.finish()
    qs = []
    for k in range(8):
        q = backbone.copy()
        pos = rng.choice(len(q), 40, replace=False)
        q[pos] = rng.choice(alpha, 40, p=probs)
        cut = sorted(rng.choice(len(q), 2, replace=False))
        qs.append(np.concatenate([q[:cut[0]], q[cut[0] + int(rng.integers(0, 4)):]])[:int(rng.integers(530, 820))])
    qs.append(backbone[:300].copy())                       # short: the 512-column kernel in its own batch below
    for costs in ((4, 6, 2), (3, 9, 1)):
        res, _ = _check_against_dense(engine, oracle, g, qs, costs=costs)
        _check_against_astar(oracle, g, qs, res, costs=costs, heuristic=oracle.H_DIJKSTRA, pruning=False)
    res, _ = _check_against_dense(engine, oracle, g, qs[-1:])


def test_multi_wave_pipeline(engine, oracle):
    """Long queries run one workgroup per query with the strips pipelined over its waves: mixed lengths in one launch
    (waves beyond a query's last strip exit early), more strips than waves (a second group through the carry array),
    and a bubble-rich graph where every row reads several predecessors from the planes."""
    g, (qseq, qoff) = W.scaled_linearish(260, 12, 6, 3, 9000, p_sub=0.03, p_ins=0.02, p_del=0.02)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(3)]
    qs = [qs[0], qs[1][:700], qs[2][:3000], qs[1][:8191], qs[2][:8192], qs[0][:1500]]   # 18, 2, 6, 16, 17 (512-col) strips
    res, _ = _check_against_dense(engine, oracle, g, qs)
    poa = W.LayeredPOA(n_layers=120, width=4, indeg=4, seed=9)
    qs = poa.queries(5, length=2600)
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs)
    _check_against_astar(oracle, poa.graph, qs, res)
    # scores beyond u16 (a 1.5 kbp read against a 36 k-node graph, Global: > 65 k of deletions): u32 planes, same pipeline
    g, (qseq, qoff) = W.scaled_linearish(36000, 300, 150, 3, 1500)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(3)] + [qseq[:1100]]
    res, _ = _check_against_dense(engine, oracle, g, qs)
    assert int(res.score.min()) > 65534


def test_deep_bubbles(engine, oracle):
    poa = W.LayeredPOA(n_layers=60, width=4, indeg=4, seed=5)
    qs = poa.queries(10, length=0)
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs, planes=True)
    _check_against_astar(oracle, poa.graph, qs, res)


def test_msa_graph(engine, oracle):
    poa = W.PangenomePOA(ref_len=400, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
    qs = poa.queries(8, length=150)
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs, planes=True)
    _check_against_astar(oracle, poa.graph, qs, res)


def test_medium_members_of_config4_and_config5(engine, oracle):
    """BASELINE.json configs[3] / configs[4] graph families at sizes the oracle still finishes in seconds:
    u32 planes (scores no longer fit u16), multi-predecessor rows everywhere, long rows."""
    poa = W.LayeredPOA(n_layers=1200, width=4, indeg=4, seed=5)       # 4802 rows, in-degree 4
    qs = poa.queries(6, length=1500)                                    # 2 strips
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs)
    _check_against_astar(oracle, poa.graph, qs, res)
    pg = W.PangenomePOA(ref_len=4000, n_hap=12, seed=4)                # ~4.1k rows
    qs = pg.queries(4, length=1500)                                     # 2 strips, Global against the whole graph
    res, _ = _check_against_dense(engine, oracle, pg.graph, qs)
    _check_against_astar(oracle, pg.graph, qs, res)


def test_u16_planes_by_score_upper_bound(engine, oracle):
    """Mismatch cost 255: the crude bound (rows + L + 2) * max(x, o + e) says u32, the bound on the OPTIMAL score
    (insert the query + delete the shortest path) says u16 — cells only reachable through long mismatch runs
    saturate to INF in u16, and nothing the result depends on may change."""
    g, (qseq, qoff) = W.scaled_linearish(420, 20, 10, 12, 400, p_sub=0.2, p_ins=0.05, p_del=0.05)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(12)]
    rng = np.random.default_rng(5)
    qs.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), 400))       # unrelated query: mismatch-heavy cells everywhere
    # A* side: Dijkstra order without pruning (`AffineDijkstra` + `align_no_pruning`), which is optimal by construction;
    # at this divergence the min-gap + pruning search can return a score 1 above the optimum of its own graph
    # (tests/test_cpu_side.py::test_mingap_with_pruning_can_be_suboptimal) — the exact-replay mode reproduces that.
    for costs in ((255, 3, 1), (200, 0, 1), (255, 6, 2)):
        res, _ = _check_against_dense(engine, oracle, g, qs, costs=costs)
        _check_against_astar(oracle, g, qs, res, costs=costs, heuristic=oracle.H_DIJKSTRA, pruning=False)
    poa = W.LayeredPOA(n_layers=150, width=4, indeg=4, seed=7)
    qs = poa.queries(6, length=140)
    res, _ = _check_against_dense(engine, oracle, poa.graph, qs, costs=(255, 2, 1))
    _check_against_astar(oracle, poa.graph, qs, res, costs=(255, 2, 1), heuristic=oracle.H_DIJKSTRA, pruning=False)


def test_chunked_workspace_uses_every_plan(engine, oracle):
    """A workspace too small for the batch: the compact layout is packed at its real size (M + flag codes + the kept D rows,
    plan 2), three full u16 planes take half the u32 space (plan 1), the exact replay runs on the u32 plan (plan 0); results
    equal the unchunked run's every way."""
    g, (qseq, qoff) = W.scaled_linearish(200, 10, 5, 24, 180)
    costs = _costs(engine)
    whole = engine.ResidentBatch(g, qseq, qoff)
    whole.run(costs)
    ref = whole.fetch()
    per_query_u32 = 3 * (g.n) * 192 * 4
    rb = engine.ResidentBatch(g, qseq, qoff, workspace_bytes=5 * per_query_u32)
    rb.run(costs)
    a = rb.fetch()
    import os
    if os.environ.get("POA_PLANES") != "32" and os.environ.get("POA_COMPACT") != "0":   # (debug overrides that force another plan)
        assert a.stats["n_chunks"] == 2                  # compact layout: < 3 of the 12 bytes per cell of the u32 planes
    rb.run(costs, None, engine.make_config("exact"))
    e = rb.fetch()
    assert e.stats["n_chunks"] == 5                      # 5 per chunk in 4-byte elements
    rb.run(costs, None, engine.make_config(full_planes=True))
    f = rb.fetch()
    if os.environ.get("POA_PLANES") != "32":
        assert f.stats["n_chunks"] == 3                  # 10 queries per chunk in 2-byte elements, three planes
    assert np.array_equal(a.score, ref.score) and np.array_equal(f.score, ref.score) and np.array_equal(e.score, ref.score)
    assert np.array_equal(a.flags, ref.flags) and np.array_equal(f.flags, ref.flags)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    A = og.astar_batch(qseq, qoff, oracle.Costs(4, 6, 2), oracle.H_MINGAP, True, threads=4)
    for i in range(24):
        assert a.raw_alignment(i) == ref.raw_alignment(i) and f.raw_alignment(i) == ref.raw_alignment(i)
        assert e.raw_alignment(i) == oracle.batch_alignment(A, i)
    whole.close(); rb.close()


def test_config2_sample_vs_astar(engine, oracle):
    """BASELINE.json configs[1] shape, a 64-query sample: every score equals the A* restatement's."""
    g, (qseq, qoff) = W.config2(n_queries=64)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(64)]
    res, _ = _check_against_dense(engine, oracle, g, qs)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    A = og.astar_batch(qseq, qoff, oracle.Costs(4, 6, 2), oracle.H_MINGAP, True, threads=8)
    assert np.array_equal(res.score, A["score"])
    _check_against_astar(oracle, g, qs, res)
    assert res.stats["cells"] == 64 * 1002 * 1001


def test_error_codes(engine):
    from poasta_amd import _lib
    # cycle
    sym = np.array([35, 36, 65, 67], np.uint8)
    succ_off = np.array([0, 1, 1, 2, 4], np.uint32)
    succ = np.array([2, 3, 1, 2], np.uint32)
    pred_off = np.array([0, 0, 1, 3, 4], np.uint32)
    pred = np.array([3, 0, 3, 2], np.uint32)
    from poasta_amd.graph import FlatGraph
    with pytest.raises(_lib.PoaError) as ei:
        engine.DeviceGraph(FlatGraph(4, 0, 1, sym, succ_off, succ, pred_off, pred))
    assert ei.value.code == -2


def test_randomised_differential_run(engine):
    """scripts/fuzz_parity.py: random graphs / costs / lengths, dense mode vs the oracle's dense restatement and exact mode
    vs its A*.  Seeds 81 and 409 are the two differences the first long run found (flag plane of a < 64-column row inside
    a one-strip batch; what is emitted after a step at which the reference panics), then a fresh block of seeds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["--first", "81", "--seeds", "1"], ["--first", "409", "--seeds", "1"], ["--first", "2000", "--seeds", "48", "--seconds", "60"]):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py")] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_flag_encodings_of_the_one_strip_kernel_agree(engine, oracle):
    """poa_forward_px_kernel stores the four traceback flags as bit-planes (code_fmt 1), two of them (code_fmt 2) or all four
    (code_fmt 3) beside the score, as the bound on the optimal score allows; POA_MF caps the variant.  Same results each way."""
    import os
    g, (qseq, qoff) = W.scaled_linearish(600, 30, 15, 40, 700)
    costs = _costs(engine)
    runs = []
    for cap in ("0", "1", "2"):
        os.environ["POA_MF"] = cap
        try:
            rb = engine.ResidentBatch(g, qseq, qoff)
            rb.run(costs)
            runs.append(rb.fetch())
            rb.close()
        finally:
            del os.environ["POA_MF"]
    for r in runs[1:]:
        assert np.array_equal(r.score, runs[0].score) and np.array_equal(r.flags, runs[0].flags)
        assert np.array_equal(r.pair_off, runs[0].pair_off) and np.array_equal(r.pairs, runs[0].pairs)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    D = og.dense_batch(qseq, qoff, oracle.Costs(4, 6, 2), threads=4)
    assert np.array_equal(runs[2].score, D["score"]) and np.array_equal(runs[2].flags, D["flags"])
    for i in range(40):
        assert runs[2].raw_alignment(i) == oracle.batch_alignment(D, i)


def test_traceback_is_the_same_walk_at_every_speculation_width(engine):
    """Lanes per walk and speculation depth only change how many steps a round of the traceback takes at once (diagonal runs,
    deletion runs, insertion runs): pairs and flags must not move — including depth 64, where a whole wave is accepted."""
    import os
    g, (qseq, qoff) = W.config2(n_queries=600)        # reads that end in ~75 inserted bases
    costs = _costs(engine)
    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            rb = engine.ResidentBatch(g, qseq, qoff)
            rb.run(costs)
            r = rb.fetch()
            rb.close()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        return r
    ref = run({"POA_TB_GROUP": "16", "POA_TB_DEPTH": "1"})     # one step per round: the sequential rule
    assert int((ref.pairs[:, 0] == 0xFFFFFFFF).sum()) > 20000   # the insertion runs are there
    for grp, depth in ((64, 64), (64, 32), (32, 32), (16, 16), (8, 8), (64, 2)):
        r = run({"POA_TB_GROUP": str(grp), "POA_TB_DEPTH": str(depth)})
        assert np.array_equal(r.flags, ref.flags) and np.array_equal(r.pair_off, ref.pair_off) and np.array_equal(r.pairs, ref.pairs), (grp, depth)
    r = run({})
    assert np.array_equal(r.flags, ref.flags) and np.array_equal(r.pairs, ref.pairs)

"""Exact-replay mode (poasta_amd/csrc/poa_exact.hpp): the product's flat restatement of the reference's
A* search.  CPU: compiled for the host and diffed against the oracle — visited table, score and the
three search counters must be identical.  GPU: alignments must be bit-identical to the oracle's for
EVERY query, ties included."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from poasta_amd import workloads as W
from poasta_amd.graph import GraphBuilder, pack_queries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(ROOT, "tests", "exact_host", "exact_host.cpp")
    out = os.path.join(ROOT, "tests", "exact_host", "libexact_host.so")
    deps = [src] + [os.path.join(ROOT, "poasta_amd", "csrc", f) for f in ("poa_exact.hpp", "poa_graph.cpp", "poa_graph.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src,
                               os.path.join(ROOT, "poasta_amd", "csrc", "poa_graph.cpp")])
    X = C.CDLL(out)
    X.exact_host_run.argtypes = [C.c_uint32] * 3 + [vp] * 5 + [C.c_uint8] * 3 + [C.c_int, C.c_int, vp, C.c_uint32, vp, vp, vp, vp, vp]
    X.exact_host_bubbles.argtypes = [C.c_uint32] * 3 + [vp] * 5 + [vp] * 5 + [C.c_uint32]
    X.exact_host_set_batch.argtypes = [C.c_uint32]
    X.exact_host_set_parallel.argtypes = [C.c_uint32] * 3
    X.exact_host_set_records.argtypes = [C.c_uint32]
    X.exact_host_run2.argtypes = [C.c_uint32] * 3 + [vp] * 5 + [vp, C.c_int, C.c_int, vp, C.c_uint32, vp, vp, vp]
    return X


@pytest.fixture(params=[0, 1, 7, 64, "records_batch7", (63, 4, 1), (5, 1, 1), (63, 8, 0), (8, 0, 1), (8, 0, 2), (63, 0, 2), (3, 0, 2), (16, 0, 0)],
                ids=["linked_list_queue", "buckets_batch1", "buckets_batch7", "buckets_batch64", "buckets_batch7_row_records", "parallel_63x4", "parallel_5x1",
                     "parallel_63x8_generic_code", "flat_8", "flat_8_lean", "flat_63_lean", "flat_3_lean", "flat_16_generic_code"])
def queue_variant(request, harness):
    """0: ExactSearch::run (linked-list queue, the one-search-per-lane kernel); n: ExactSearch::run_buckets(n), the
    step schedule of the wave-per-query kernel (poa_wsearch.hpp) over the bucket queue; (lanes, rmax, fast):
    ExactSearch::run_parallel (rmax > 0: the top entries of a stack expanded at once in log mode, each lane following what its
    entry puts in front of the next) or ExactSearch::run_flat (rmax == 0: the schedule of poa_fsearch.hpp — the next entries in
    pop order, one per lane; fast == 2: the lean step, greedy extensions resumed from step to step)."""
    if isinstance(request.param, tuple):
        harness.exact_host_set_parallel(*request.param)
    elif request.param == "records_batch7":
        # the wave kernels' instantiation whose one-round-trip path reads the per-row records (successors, symbols, bubbles)
        harness.exact_host_set_batch(7)
        harness.exact_host_set_records(1)
    else:
        harness.exact_host_set_batch(request.param)
    yield request.param
    harness.exact_host_set_records(0)
    harness.exact_host_set_batch(0)
    harness.exact_host_set_parallel(0, 4, 1)


def _oracle_table(oracle, og, q, costs, heur, prune, n):
    L = oracle.lib()
    L.oracle_astar_table.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_int, vp, C.c_uint64, vp, vp, vp, vp]
    m, i, d = (np.zeros((n, len(q) + 1), np.uint32) for _ in range(3))
    out = np.zeros(4, np.uint64)
    rc = L.oracle_astar_table(og.h, *costs, heur, prune, _p(q), len(q), _p(m), _p(i), _p(d), _p(out))
    return rc, m, i, d, out


def _span_args(oracle, span):
    """span = None (Global) or dict(qry_free_end=, graph_free_begin=, graph_free_end=) with oracle-style bounds."""
    if span is None:
        return None, None
    b = lambda v: (v, 0) if isinstance(v, int) else v
    qfe, gfb, gfe = b(span.get("qry_free_end", 0)), b(span.get("graph_free_begin", 0)), b(span.get("graph_free_end", 0))
    arr = np.array([1, qfe[0], qfe[1], gfb[0], gfe[0], gfe[1]], np.uint32)
    return arr, oracle.ends_free(0, span.get("qry_free_end", 0), span.get("graph_free_begin", 0), span.get("graph_free_end", 0))


def _compare(oracle, X, g, qs, costs, heur, prune, span=None):
    og = oracle.OracleGraph.from_csr(g.as_dict())
    n_ok = 0
    sarr, ospec = _span_args(oracle, span)
    for q in qs:
        q = np.ascontiguousarray(q, np.uint8)
        if ospec is None:
            rc1, om, oi, od, oo = _oracle_table(oracle, og, q, costs, heur, prune, g.n)
        else:
            with oracle.alignment_type(ospec):
                rc1, om, oi, od, oo = _oracle_table(oracle, og, q, costs, heur, prune, g.n)
        xm, xi, xd = (np.zeros((g.n, len(q) + 1), np.uint32) for _ in range(3))
        xo = np.zeros(6 if sarr is not None else 4, np.uint32)
        rc2 = X.exact_host_run(g.n, g.start, g.end, _p(g.symbol), _p(g.succ_off), _p(g.succ), _p(g.pred_off), _p(g.pred),
                               *costs, heur, prune, _p(q), len(q), _p(xo), _p(xm), _p(xi), _p(xd),
                               _p(sarr) if sarr is not None else None)
        xo = xo[:4]
        if rc1 == 1 and rc2 == 0:
            continue  # the oracle's panic came from the BACKTRACE (u32 wrap), which the search does not include
        assert rc1 == 0 and rc2 == 0, (rc1, rc2)
        assert oo.tolist() == xo.tolist(), "score / num_queued / num_visited / num_pruned"
        assert np.array_equal(om, xm) and np.array_equal(oi, xi) and np.array_equal(od, xd), "visited table"
        n_ok += 1
    return n_ok


def _compare2(oracle, X, g, qs, costs5, heur, prune, span=None):
    """Two-piece model (gap_affine_2piece.rs): costs5 = (mismatch, open1, extend1, open2, extend2).  The product's search
    object instantiated with EX_AS_TWO_PIECE against the oracle's literal two-piece search: counters, end cell and all
    five planes of the visited table."""
    og = oracle.OracleGraph.from_csr(g.as_dict())
    L = oracle.lib()
    L.oracle_astar_table2.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_int, vp, C.c_uint64, vp, vp]
    sarr, ospec = _span_args(oracle, span)
    x, o1, e1, o2, e2 = costs5
    c5 = np.array(costs5, np.uint8)
    n_ok = 0
    for q in qs:
        q = np.ascontiguousarray(q, np.uint8)
        op = [np.zeros((g.n, len(q) + 1), np.uint32) for _ in range(5)]
        xp = [np.zeros((g.n, len(q) + 1), np.uint32) for _ in range(5)]
        optr = (vp * 5)(*[a.ctypes.data for a in op])
        xptr = (vp * 5)(*[a.ctypes.data for a in xp])
        oo = np.zeros(4, np.uint64)
        with oracle.two_piece(o2, e2):
            if ospec is None:
                rc1 = L.oracle_astar_table2(og.h, x, o1, e1, heur, prune, _p(q), len(q), optr, _p(oo))
            else:
                with oracle.alignment_type(ospec):
                    rc1 = L.oracle_astar_table2(og.h, x, o1, e1, heur, prune, _p(q), len(q), optr, _p(oo))
        xo = np.zeros(6, np.uint32)
        rc2 = X.exact_host_run2(g.n, g.start, g.end, _p(g.symbol), _p(g.succ_off), _p(g.succ), _p(g.pred_off), _p(g.pred),
                                _p(c5), heur, prune, _p(q), len(q), _p(xo), xptr, _p(sarr) if sarr is not None else None)
        if rc1 == 1 and rc2 == 0:
            continue  # the oracle's panic came from the backtrace, which the search does not include
        if rc1 == 1 and rc2 == 2:
            n_ok += 1   # "Could not align sequence!" on both sides (EX_PANIC)
            continue
        assert rc1 == 0 and rc2 == 0, (rc1, rc2)
        assert oo.tolist() == xo[:4].tolist(), "score / num_queued / num_visited / num_pruned"
        for a, b in zip(op, xp):
            assert np.array_equal(a, b), "visited table"
        n_ok += 1
    return n_ok


def test_two_piece_replay_equals_oracle_search_cpu(oracle, harness):
    """`poasta align -g 6,24 -e 2,1` builds Affine2PieceMinGapCost with pruning (config.rs:215-272): every heuristic /
    pruning combination, random DAGs, ends-free spans; plus the structured workloads."""
    n_ok = 0
    for seed in range(60):
        rng = np.random.Generator(np.random.PCG64(7000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(6)]
        costs5 = [(4, 6, 2, 24, 1), (1, 3, 12, 6, 1), (2, 8, 2, 12, 2), (3, 1, 1, 1, 1), (4, 4, 3, 10, 1)][seed % 5]
        for heur, prune in ((1, 1), (0, 1), (1, 0), (0, 0)):
            n_ok += _compare2(oracle, harness, g, qs, costs5, heur, prune)
        if seed % 3 == 0:
            for span in (dict(qry_free_end=oracle.UNBOUNDED, graph_free_begin=oracle.UNBOUNDED, graph_free_end=oracle.UNBOUNDED),
                         dict(qry_free_end=(oracle.INCLUDED, 2), graph_free_end=(oracle.EXCLUDED, 3)),
                         dict(graph_free_begin=oracle.UNBOUNDED, graph_free_end=(oracle.INCLUDED, 1))):
                n_ok += _compare2(oracle, harness, g, qs, costs5, 1, 1, span)
    assert n_ok > 1300, n_ok
    g, (qseq, qoff) = W.scaled_linearish(300, 15, 8, 16, 330)
    assert _compare2(oracle, harness, g, [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(16)], (4, 6, 2, 24, 1), 1, 1) == 16
    pg = W.PangenomePOA(ref_len=300, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
    assert _compare2(oracle, harness, pg.graph, pg.queries(6, length=120), (4, 6, 2, 24, 1), 1, 1) == 6


def test_product_bubble_index_matches_oracle(oracle, harness):
    for seed in range(40):
        g = W.random_dag(seed, n_nodes=12, p_edge=0.3)
        og = oracle.OracleGraph.from_csr(g.as_dict())
        bi = og.bubble_index()
        n = g.n
        dmin, dmax, ex = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        off, nbm = np.zeros(n + 1, np.uint32), np.zeros((64 * n, 3), np.uint32)
        rc = harness.exact_host_bubbles(g.n, g.start, g.end, _p(g.symbol), _p(g.succ_off), _p(g.succ), _p(g.pred_off),
                                        _p(g.pred), _p(dmin), _p(dmax), _p(ex), _p(off), _p(nbm), 64 * n)
        assert rc == 0
        assert list(zip(dmin.tolist(), dmax.tolist())) == bi["dist_to_end"]
        assert ex.astype(bool).tolist() == bi["is_exit"]
        for v in range(n):
            assert [tuple(x) for x in nbm[off[v]:off[v + 1]].tolist()] == bi["node_bubble_map"][v]


def test_replay_equals_oracle_search_cpu(oracle, harness, queue_variant):
    n_ok = 0
    for seed in range(60):
        rng = np.random.Generator(np.random.PCG64(1000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(8)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (3, 1, 1), (4, 4, 2)][seed % 5]
        for heur, prune in ((1, 1), (0, 1), (1, 0), (0, 0)):
            n_ok += _compare(oracle, harness, g, qs, costs, heur, prune)
    assert n_ok > 1500
    g, (qseq, qoff) = W.scaled_linearish(300, 15, 8, 16, 330)
    assert _compare(oracle, harness, g, [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(16)], (4, 6, 2), 1, 1) == 16
    poa = W.LayeredPOA(n_layers=40, width=4, indeg=4, seed=5)
    assert _compare(oracle, harness, poa.graph, poa.queries(6, length=0), (4, 6, 2), 1, 1) == 6
    pg = W.PangenomePOA(ref_len=300, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
    assert _compare(oracle, harness, pg.graph, pg.queries(6, length=120), (4, 6, 2), 1, 1) == 6


# ------------------------------------------------------------------------------------------------
def _gpu_exact_vs_astar(engine, oracle, g, qs, costs=(4, 6, 2), cfg_cls="AffineMinGapCost", pruning=True, mode="exact", allow_flags=0, queue_entries_per_cell=3.0):
    m, o, e = costs
    al = engine.PoastaAligner(getattr(engine, cfg_cls)(engine.GapAffine(m, e, o)), mode=mode, queue_entries_per_cell=queue_entries_per_cell)
    qseq, qoff = pack_queries(qs)
    res = al.align_batch(g, qseq=qseq, qoff=qoff, pruning=pruning)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    heur = oracle.H_MINGAP if cfg_cls == "AffineMinGapCost" else oracle.H_DIJKSTRA
    A = og.astar_batch(qseq, qoff, oracle.Costs(*costs), heur, pruning, threads=4)
    n = 0
    for i in range(len(qs)):
        if A["status"][i] != 0:
            assert int(res.flags[i]) & 4, "reference panics: REF_PANIC expected"
            continue
        assert int(res.flags[i]) & ~allow_flags == 0, "flags, query %d" % i
        assert int(res.score[i]) == int(A["score"][i]), "score, query %d" % i
        assert res.raw_alignment(i) == oracle.batch_alignment(A, i), "alignment, query %d" % i
        n += 1
    return n, res


def test_replay_equals_oracle_search_ends_free_cpu(oracle, harness, queue_variant):
    """AlignmentType::EndsFree: the replay's visited table, score and counters equal the oracle's for every kind of
    bound the reference distinguishes (gap_affine.rs:136-248)."""
    U, INC, EXC = oracle.UNBOUNDED, oracle.INCLUDED, oracle.EXCLUDED
    spans = [dict(), dict(graph_free_begin=(INC, 0)), dict(qry_free_end=(INC, 2)), dict(qry_free_end=(EXC, 3), graph_free_end=(INC, 2)),
             dict(graph_free_begin=(INC, 0), qry_free_end=(INC, 0), graph_free_end=(EXC, 3)), dict(graph_free_end=(INC, 0))]
    n_ok = 0
    for seed in range(40):
        rng = np.random.Generator(np.random.PCG64(3000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(6)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (3, 1, 1)][seed % 4]
        span = spans[seed % len(spans)]
        for heur, prune in ((1, 1), (0, 0)):
            n_ok += _compare(oracle, harness, g, qs, costs, heur, prune, span=span)
    assert n_ok > 300
    g, (qseq, qoff) = W.scaled_linearish(120, 6, 3, 6, 60)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(6)]
    assert _compare(oracle, harness, g, qs, (4, 6, 2), 1, 1, span=dict(qry_free_end=(INC, 0))) == 6


@pytest.mark.gpu
def test_gpu_exact_mode_is_bit_identical(engine, oracle):
    n = 0
    for seed in range(40):
        rng = np.random.Generator(np.random.PCG64(1000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(16)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (4, 4, 2)][seed % 4]
        cfg = "AffineMinGapCost" if seed % 3 else "AffineDijkstra"
        n += _gpu_exact_vs_astar(engine, oracle, g, qs, costs, cfg, pruning=(seed % 5 != 0))[0]
    assert n > 500
    g, (qseq, qoff) = W.scaled_linearish(300, 15, 8, 48, 330)
    assert _gpu_exact_vs_astar(engine, oracle, g, [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(48)])[0] == 48
    poa = W.LayeredPOA(n_layers=60, width=4, indeg=4, seed=5)
    assert _gpu_exact_vs_astar(engine, oracle, poa.graph, poa.queries(10, length=0))[0] == 10
    pg = W.PangenomePOA(ref_len=400, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
    assert _gpu_exact_vs_astar(engine, oracle, pg.graph, pg.queries(8, length=150))[0] == 8


@pytest.mark.gpu
def test_gpu_exact_reproduces_suboptimal_pruned_search(engine, oracle):
    """Costs 8/3/1, 20 % substitutions: the min-gap + pruning search returns 501 / 489 where the optimum is 500 / 488
    (tests/test_cpu_side.py::test_mingap_with_pruning_can_be_suboptimal).  The replay must return what the search returns."""
    g, (qseq, qoff) = W.scaled_linearish(420, 20, 10, 12, 400, p_sub=0.2, p_ins=0.05, p_del=0.05)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(12)]
    # 0x10 = POA_FLAG_TRUNCATED: informational (the reference drops leading insertions from the alignment it returns)
    n, res = _gpu_exact_vs_astar(engine, oracle, g, qs, costs=(8, 3, 1), allow_flags=0x10)
    assert n == 10 and int(res.score[4]) == 501 and int(res.score[9]) == 489
    n, res = _gpu_exact_vs_astar(engine, oracle, g, qs, costs=(8, 3, 1), cfg_cls="AffineDijkstra", pruning=False, allow_flags=0x10,
                                 queue_entries_per_cell=12.0)  # Dijkstra order without pruning queues every cell several times
    assert int(res.score[4]) == 500 and int(res.score[9]) == 488


def _gpu_ends_free_vs_astar(engine, oracle, g, qs, costs, cfg_cls, pruning, bounds):
    """bounds: dict of (kind, value) pairs keyed like AlignmentType::EndsFree's fields."""
    m, o, e = costs
    B = engine.Bound
    conv = lambda b: B.Unbounded if b[0] == 0 else (B.Included(b[1]) if b[0] == 1 else B.Excluded(b[1]))
    names = ("qry_free_begin", "qry_free_end", "graph_free_begin", "graph_free_end")
    at = engine.AlignmentType.EndsFree(**{k: conv(bounds.get(k, (0, 0))) for k in names})
    al = engine.PoastaAligner(getattr(engine, cfg_cls)(engine.GapAffine(m, e, o)), aln_type=at, queue_entries_per_cell=4.0)
    qseq, qoff = pack_queries(qs)
    res = al.align_batch(g, qseq=qseq, qoff=qoff, pruning=pruning)
    og = oracle.OracleGraph.from_csr(g.as_dict())
    heur = oracle.H_MINGAP if cfg_cls == "AffineMinGapCost" else oracle.H_DIJKSTRA
    spec = oracle.ends_free(*[bounds.get(k, (0, 0)) if bounds.get(k, (0, 0))[0] else 0 for k in names])
    n = 0
    with oracle.alignment_type(spec):
        A = og.astar_batch(qseq, qoff, oracle.Costs(*costs), heur, pruning, threads=2)
    for i in range(len(qs)):
        if A["status"][i] != 0:
            assert int(res.flags[i]) & 4, "reference panics: REF_PANIC expected (query %d)" % i
            continue
        assert int(res.flags[i]) & ~0x10 == 0, "flags, query %d" % i     # TRUNCATED: does not begin at the start node
        assert int(res.score[i]) == int(A["score"][i]), "score, query %d" % i
        assert res.raw_alignment(i) == oracle.batch_alignment(A, i), "alignment, query %d" % i
        n += 1
    return n


@pytest.mark.gpu
def test_gpu_ends_free_matches_the_literal_search(engine, oracle):
    """AlignmentType::EndsFree through the C ABI (the replay is implied): scores and alignments equal the oracle's
    literal restatement, which the reference's own ends-free known answers pin (tests/test_oracle_kat.py)."""
    INC, EXC = 1, 2
    spans = [dict(), dict(graph_free_begin=(INC, 0)), dict(qry_free_end=(INC, 2)), dict(qry_free_end=(EXC, 3), graph_free_end=(INC, 2)),
             dict(graph_free_begin=(INC, 0), qry_free_end=(INC, 0), graph_free_end=(EXC, 3)), dict(graph_free_end=(INC, 0))]
    n = 0
    for seed in range(36):
        rng = np.random.Generator(np.random.PCG64(5000 + seed))
        alpha = b"AC" if seed % 2 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 14)), p_edge=0.3, alphabet=alpha)
        qs = [W.random_walk_query(rng, g, 0.3, alpha) for _ in range(12)]
        costs = [(4, 6, 2), (2, 8, 1), (1, 10, 2), (3, 1, 1)][seed % 4]
        cfg = "AffineMinGapCost" if seed % 3 else "AffineDijkstra"
        n += _gpu_ends_free_vs_astar(engine, oracle, g, qs, costs, cfg, seed % 5 != 0, spans[seed % len(spans)])
    assert n > 300
    # the reference's own fixtures (gap_affine.rs:1124-1182, :1320-1352; edge_cases.rs:64-112, :225-262)
    b = GraphBuilder(); b.add_path(np.frombuffer(b"ATCG", np.uint8)); g = b.finish()
    al = engine.PoastaAligner(engine.AffineDijkstra(engine.GapAffine(1, 2, 8)), aln_type=engine.AlignmentType.EndsFree())
    r = al.align_batch(g, [b"TCG", b"ATCGAA", b""])
    assert r.score.tolist() == [0, 0, 0]
    b = GraphBuilder(); b.add_path(np.frombuffer(b"AAAA", np.uint8)); g = b.finish()
    al = engine.PoastaAligner(engine.AffineDijkstra(engine.GapAffine(2, 1, 8)), aln_type=engine.AlignmentType.EndsFree())
    r = al.align_batch(g, [b"TTTT"])
    assert int(r.score[0]) == 2 and r.raw_alignment(0) == []          # "chose not to align (cost 2)", edge_cases.rs:100-108
    b = GraphBuilder(); b.add_path(np.frombuffer(b"A", np.uint8)); g = b.finish()
    al = engine.PoastaAligner(engine.AffineDijkstra(engine.GapAffine(1, 2, 8)), aln_type=engine.AlignmentType.EndsFree())
    r = al.align_batch(g, [b"A", b"T"])
    assert int(r.score[0]) == 0 and int(r.score[1]) == 10 and r.raw_alignment(1) == [(0xFFFFFFFF, 0)]   # edge_cases.rs:246-256


@pytest.mark.gpu
def test_gpu_exact_config2_sample_and_hybrid(engine, oracle):
    g, (qseq, qoff) = W.config2(n_queries=96)
    qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(96)]
    n, res = _gpu_exact_vs_astar(engine, oracle, g, qs)
    assert n == 96 and res.stats["n_exact"] == 96
    n, res = _gpu_exact_vs_astar(engine, oracle, g, qs, mode="hybrid")
    assert n == 96 and res.stats["n_exact"] <= 96


@pytest.mark.gpu
def test_gpu_exact_overflow_keeps_dense_result(engine, oracle):
    g, (qseq, qoff) = W.scaled_linearish(300, 15, 8, 8, 330)
    al_d = engine.PoastaAligner(engine.AffineMinGapCost(engine.GapAffine(4, 2, 6)))
    al_x = engine.PoastaAligner(engine.AffineMinGapCost(engine.GapAffine(4, 2, 6)), mode="exact", queue_entries_per_cell=1e-6)
    d = al_d.align_batch(g, qseq=qseq, qoff=qoff)
    os.environ["POA_WS_CHUNK_CAP"] = "6"   # wave search: every live stack holds a chunk, so the ring adds to the pool; cap it
    try:
        x = al_x.align_batch(g, qseq=qseq, qoff=qoff)
    finally:
        del os.environ["POA_WS_CHUNK_CAP"]
    for i in range(8):
        if int(x.flags[i]) & 0x40:  # POA_FLAG_EXACT_OVERFLOW: dense result kept
            assert int(x.score[i]) == int(d.score[i]) and x.raw_alignment(i) == d.raw_alignment(i)
    assert any(int(f) & 0x40 for f in x.flags)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(POA_WS_RING_GLOBAL="1"), dict(POA_WS_GROUP="16"), dict(POA_WS_GROUP="8", POA_WS_WAVES="2"),
                                 dict(POA_WS_STATIC="1"), dict(POA_WS_LANES="1"), dict(POA_WS_LANES="63", POA_WS_WAVES="4"),
                                 dict(POA_WS_REC="0"), dict(POA_WS_ADAPT="0", POA_WS_LANES="16"), dict(POA_WS_ADAPT="1", POA_WS_LANES="63"),
                                 dict(POA_EXACT_LDS="0"), dict(POA_EXACT_IMPL="lane"),
                                 dict(POA_EXACT_IMPL="flat"), dict(POA_EXACT_IMPL="flat", POA_PS_LEAN="0"),
                                 dict(POA_EXACT_IMPL="flat", POA_PS_LANES="3", POA_WS_RING_GLOBAL="1"),
                                 dict(POA_EXACT_IMPL="flat", POA_EXACT_LDS="0", POA_WS_STATIC="1")],
                         ids=["ring_in_global_memory", "four_queries_per_wave", "eight_queries_per_wave", "static_schedule",
                              "one_entry_per_step", "63_entries_per_step", "test_over_graph_arrays_not_records", "fixed_test_width_16",
                              "adaptive_width_from_1_to_63", "graph_in_global_memory", "one_search_per_lane_kernel",
                              "flat_schedule_lean_step", "flat_schedule_generic_code", "flat_3_lanes_ring_global", "flat_records_in_global_memory"])
def test_gpu_replay_variants_are_bit_identical(engine, oracle, env):
    """Every schedule / placement variant of the replay kernels returns the reference's alignments: the descriptor ring in
    global memory (wide priority ranges), several queries per wave, the static schedule, single-entry and 63-entry steps,
    the graph read from global memory, the round-1 one-search-per-lane kernel, and the parallel-step kernel of round 3
    (poa_fsearch.hpp: eight queries per wave with the lean step, four with the generic code in log mode)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g, (qseq, qoff) = W.config2(n_queries=40)
        qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(40)]
        n, res = _gpu_exact_vs_astar(engine, oracle, g, qs)
        assert n == 40 and res.stats["n_exact"] == 40
        poa = W.LayeredPOA(n_layers=60, width=4, indeg=4, seed=5)
        assert _gpu_exact_vs_astar(engine, oracle, poa.graph, poa.queries(6, length=0))[0] == 6
        pg = W.PangenomePOA(ref_len=400, n_hap=6, p_snp=0.02, p_indel=0.01, max_indel=6, seed=4)
        assert _gpu_exact_vs_astar(engine, oracle, pg.graph, pg.queries(6, length=150))[0] == 6
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_gpu_search_counters_equal_the_reference(engine, oracle):
    """AstarResult::{num_queued, num_visited, num_pruned} of the replayed search (astar.rs:86-89) through the C ABI."""
    g, (qseq, qoff) = W.config2(n_queries=24)
    rb = engine.ResidentBatch(g, qseq, qoff)
    rb.run(engine.GapAffine(4, 2, 6), None, engine.make_config("exact", queue_entries_per_cell=0.25))
    sc = rb.search_counters()
    rb.fetch()
    og = oracle.OracleGraph.from_csr(g.as_dict())
    A = og.astar_batch(qseq, qoff, oracle.Costs(4, 6, 2), oracle.H_MINGAP, True, threads=8, want_counters=True)
    assert np.array_equal(sc[:, :3].astype(np.uint64), A["counters"])
    assert (sc[:, 3] > 0).all() and (sc[:, 3] <= sc[:, 0]).all()   # steps: at most one per queued state
    rb.close()

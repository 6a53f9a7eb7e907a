"""BASELINE.json configs at their full sizes on the GPU, checked through size-independent properties:
the checksum of all scores (equal to the restated reference's, see profiles/r01e_px_kernel/parity_config2.json),
random members against the oracle, independence of a result from the batch it was computed in, and the shape every
Global alignment must have (each query position exactly once and in order)."""
import numpy as np
import pytest

from poasta_amd import workloads as W
from poasta_amd.graph import pack_queries

pytestmark = pytest.mark.gpu

NONE = 0xFFFFFFFF
TRUNCATED = 0x10


def _costs(engine):
    return engine.GapAffine(4, 2, 6)


def _query(qseq, qoff, i):
    return qseq[int(qoff[i]):int(qoff[i + 1])]


def _check_shape(res, qoff, idx):
    for i in idx:
        qp = [q for (_, q) in res.raw_alignment(i) if q != NONE]
        L = int(qoff[i + 1] - qoff[i])
        # (insertions at the very start of the walk are not emitted by the reference: gap_affine.rs:891-893)
        assert qp == list(range(L - len(qp), L)), "query %d: query positions once each, in order, through to the last" % i
        assert len(qp) >= L - 64


def _check_members_against_oracle(engine, oracle, g, qseq, qoff, res, idx, threads=8):
    og = oracle.OracleGraph.from_csr(g.as_dict())
    qs = [_query(qseq, qoff, i) for i in idx]
    sseq, soff = pack_queries(qs)
    D = og.dense_batch(sseq, soff, oracle.Costs(4, 6, 2), threads=threads)
    for k, i in enumerate(idx):
        assert int(res.score[i]) == int(D["score"][k]), "score of query %d" % i
        assert res.raw_alignment(i) == oracle.batch_alignment(D, k), "alignment of query %d" % i
        assert int(res.flags[i]) == int(D["flags"][k]), "flags of query %d" % i
    return qs


def _check_batch_independence(engine, g, qs, res, idx):
    """The same queries as a batch of their own: identical scores, alignments, flags."""
    al = engine.PoastaAligner(engine.AffineMinGapCost(_costs(engine)))
    small = al.align_batch(g, qs)
    for k, i in enumerate(idx):
        assert int(small.score[k]) == int(res.score[i]) and int(small.flags[k]) == int(res.flags[i])
        assert small.raw_alignment(k) == res.raw_alignment(i)


def test_config2_full_size(engine, oracle):
    """configs[1]: 1002 rows x 10 000 queries x 1 kbp (the bench workload, pairs-across-quads kernel)."""
    g, (qseq, qoff) = W.config2(n_queries=10000)
    rb = engine.ResidentBatch(g, qseq, qoff)
    rb.run(_costs(engine))
    res = rb.fetch()
    assert res.stats["cells"] == 10000 * 1002 * 1001 and res.stats["n_chunks"] == 1
    # checksum of checksums: the restated reference (A*, min-gap, pruning) gives the same 10 000 scores
    assert int(res.score.astype(np.uint64).sum()) == 3720720
    assert int((res.flags != 0).sum()) == 9928
    rng = np.random.default_rng(11)
    idx = sorted(rng.choice(10000, 48, replace=False).tolist())
    qs = _check_members_against_oracle(engine, oracle, g, qseq, qoff, res, idx)
    _check_batch_independence(engine, g, qs, res, idx)
    _check_shape(res, qoff, range(0, 10000, 7))
    # a second pass over the resident batch reproduces the first bit for bit
    rb.run(_costs(engine))
    again = rb.fetch()
    assert np.array_equal(again.score, res.score) and np.array_equal(again.flags, res.flags) and np.array_equal(again.pairs, res.pairs)
    rb.close()


def test_config5_full_size(engine, oracle):
    """configs[4]: 20 002 rows (in-degree 4) x 2 000 queries x 5 kbp — the multi-wave pipeline, several chunks."""
    g, (qseq, qoff) = W.config5(n_queries=2000)
    rb = engine.ResidentBatch(g, qseq, qoff)
    rb.run(_costs(engine))
    res = rb.fetch()
    assert res.stats["n_chunks"] > 1 and int((res.score == NONE).sum()) == 0
    rng = np.random.default_rng(12)
    idx = sorted(rng.choice(2000, 6, replace=False).tolist())
    qs = _check_members_against_oracle(engine, oracle, g, qseq, qoff, res, idx, threads=6)
    _check_batch_independence(engine, g, qs, res, idx)
    _check_shape(res, qoff, range(0, 2000, 5))
    rb.close()


def test_config4_members(engine, oracle):
    """configs[3] graph (50 kbp x 32 haplotypes as an MSA, ~56 k rows), 10 kbp reads, Global: scores beyond u16 in 2-byte
    cells (relative encoding) through the multi-wave pipeline, 40 k-row deletion runs in the traceback.  512 of the 5 000
    queries in several plane chunks, 8 oracle members (the full 5 000: scripts/config_throughput.py --config 4,
    profiles/r02_relative/)."""
    g, (qseq, qoff) = W.config4(n_queries=512)
    rb = engine.ResidentBatch(g, qseq, qoff)
    rb.run(_costs(engine))
    res = rb.fetch()
    assert int(res.score.min()) > 65534 and int((res.score == NONE).sum()) == 0
    assert rb.layout() == {"u16", "compact", "relative"}
    assert res.stats["n_chunks"] > 1
    idx = [3, 57, 121, 200, 266, 349, 430, 511]
    qs = _check_members_against_oracle(engine, oracle, g, qseq, qoff, res, idx, threads=8)
    _check_batch_independence(engine, g, qs, res, idx)
    _check_shape(res, qoff, range(0, 512, 7))
    rb.close()


def test_config2_hybrid_full_size_is_the_reference(engine, oracle):
    """configs[1] at full size in hybrid mode (dense pass + replay of the reference's search for every query the dense pass
    cannot certify): ALL 10 000 scores and alignments equal the restated reference's (A*, min-gap heuristic, pruning), and so
    do its search counters for the replayed queries."""
    g, (qseq, qoff) = W.config2(n_queries=10000)
    rb = engine.ResidentBatch(g, qseq, qoff)
    rb.run(_costs(engine), None, engine.make_config("hybrid", queue_entries_per_cell=0.25))
    sc = rb.search_counters()
    res = rb.fetch()
    og = oracle.OracleGraph.from_csr(g.as_dict())
    A = og.astar_batch(qseq, qoff, oracle.Costs(4, 6, 2), oracle.H_MINGAP, True, threads=16, want_counters=True)
    assert int((A["status"] != 0).sum()) == 0
    assert np.array_equal(res.score, A["score"])
    assert int((res.flags != 0).sum()) == 0 and res.stats["n_exact"] == 9928
    bad = [i for i in range(10000) if res.raw_alignment(i) != oracle.batch_alignment(A, i)]
    assert bad == []
    replayed = sc[:, 3] > 0
    assert int(replayed.sum()) == 9928 and np.array_equal(sc[replayed, :3].astype(np.uint64), A["counters"][replayed])
    rb.close()

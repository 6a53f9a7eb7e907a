"""Committed golden vectors (tests/golden/golden_vectors.json, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path (through the C ABI) matches them."""
import json
import os

import numpy as np
import pytest

from poasta_amd.graph import FlatGraph, pack_queries

HERE = os.path.dirname(os.path.abspath(__file__))
SCORE_UNCERTAIN = 2 | 8


def _cases():
    with open(os.path.join(HERE, "golden", "golden_vectors.json")) as f:
        return json.load(f)["cases"]


def _graph(c):
    g = c["graph"]
    return FlatGraph(g["n"], g["start"], g["end"], np.frombuffer(g["symbol"].encode("latin1"), np.uint8),
                     g["succ_off"], g["succ"], g["pred_off"], g["pred"])


def test_oracle_reproduces_golden(oracle):
    n = 0
    for c in _cases():
        og = oracle.OracleGraph.from_csr(_graph(c).as_dict())
        costs = oracle.Costs(*c["costs"])
        for q, a, d in zip(c["queries"], c["astar"], c["dense"]):
            q = q.encode("latin1")
            if "panic" in a:
                with pytest.raises(oracle.RefPanic):
                    og.astar_align(q, costs)
            else:
                r = og.astar_align(q, costs)
                assert r["score"] == a["score"] and [list(p) for p in r["alignment"]] == a["pairs"], c["name"]
            dd = og.dense_align(q, costs)
            assert dd["score"] == d["score"] and [list(p) for p in dd["alignment"]] == d["pairs"] and dd["flags"] == d["flags"]
            n += 1
    assert n >= 30


def test_golden_scores_are_consistent():
    """A* and dense scores in the fixture agree unless the dense flags say the score is uncertain."""
    for c in _cases():
        for a, d in zip(c["astar"], c["dense"]):
            if "panic" in a:
                continue
            if not d["flags"] & SCORE_UNCERTAIN:
                assert a["score"] == d["score"], c["name"]
            if d["flags"] == 0:
                assert a["pairs"] == d["pairs"], c["name"]


@pytest.mark.gpu
def test_hip_path_matches_golden(engine):
    n_cert = 0
    for c in _cases():
        g = _graph(c)
        m, o, e = c["costs"]
        al = engine.PoastaAligner(engine.AffineMinGapCost(engine.GapAffine(m, e, o)))
        res = al.align_batch(g, [q.encode("latin1") for q in c["queries"]])
        for i, (a, d) in enumerate(zip(c["astar"], c["dense"])):
            assert int(res.score[i]) == d["score"] and int(res.flags[i]) == d["flags"], c["name"]
            assert [list(p) for p in res.raw_alignment(i)] == d["pairs"], c["name"]
            if "panic" in a:
                continue
            if not d["flags"] & SCORE_UNCERTAIN:
                assert int(res.score[i]) == a["score"], c["name"]
            if d["flags"] == 0:
                n_cert += 1
                assert [list(p) for p in res.raw_alignment(i)] == a["pairs"], c["name"]
    assert n_cert >= 10

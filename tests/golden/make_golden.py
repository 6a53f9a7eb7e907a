#!/usr/bin/env python3
"""Regenerates tests/golden/golden_vectors.json.

The reference (Rust) cannot be built or run in this container, so these vectors are NOT captures of the
reference binary: they are outputs of the oracle's literal A* restatement (oracle/astar.hpp, pinned by
the reference's own known-answer tests in tests/test_oracle_kat.py) on
  * the hand-traced cases of SURVEY.md appendix C,
  * the reference's fixture reads tests/test_from_abpoa.fa / tests/test2_from_abpoa.fa (copied here as
    data), aligned one by one while the POA is built (BASELINE.json configs[0]),
  * small members of the config-2 / config-4 / config-5 graph families.
Traceback parity vs the real reference stays "unpinned" (no reference test asserts an alignment).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from poasta_amd import workloads as W  # noqa: E402
from poasta_amd.graph import GraphBuilder  # noqa: E402


def graph_json(csr):
    return dict(n=int(csr["n"]), start=int(csr["start"]), end=int(csr["end"]),
                symbol=bytes(np.asarray(csr["symbol"], np.uint8)).decode("latin1"),
                succ_off=np.asarray(csr["succ_off"]).tolist(), succ=np.asarray(csr["succ"]).tolist(),
                pred_off=np.asarray(csr["pred_off"]).tolist(), pred=np.asarray(csr["pred"]).tolist())


def case(name, og, csr, queries, costs):
    c = O.Costs(*costs)
    out = dict(name=name, graph=graph_json(csr), costs=list(costs), queries=[], astar=[], dense=[])
    for q in queries:
        q = bytes(np.asarray(np.frombuffer(q, np.uint8) if isinstance(q, (bytes, bytearray)) else q, np.uint8))
        out["queries"].append(q.decode("latin1"))
        try:
            a = og.astar_align(q, c, O.H_MINGAP, True)
            out["astar"].append(dict(score=a["score"], pairs=[list(p) for p in a["alignment"]]))
        except O.RefPanic as ex:
            out["astar"].append(dict(panic=str(ex)))
        d = og.dense_align(q, c)
        out["dense"].append(dict(score=d["score"], pairs=[list(p) for p in d["alignment"]], flags=d["flags"]))
    return out


def read_fasta(path):
    return [l.strip().encode() for l in open(path) if l.strip() and not l.startswith(">")]


def main():
    cases = []
    for gs, q, c in [(b"ACGT", b"ACGT", (4, 6, 2)), (b"ACGT", b"AC", (1, 10, 2)), (b"AAAC", b"AAC", (4, 6, 2)),
                     (b"AC", b"AAC", (4, 6, 2)), (b"AAAA", b"TTTT", (2, 8, 1)), (b"GAAC", b"AC", (4, 6, 2))]:
        og = O.OracleGraph.new_poa()
        og.add_alignment("ref", gs, None)
        cases.append(case("appendixC_%s_x_%s" % (gs.decode(), q.decode()), og, og.export_csr(), [q], c))
    for fx in ("test_from_abpoa", "test2_from_abpoa"):
        reads = read_fasta(os.path.join(HERE, fx + ".fa"))
        og = O.OracleGraph.new_poa()
        og.add_alignment("1", reads[0], None)
        for i, r in enumerate(reads[1:], start=2):
            cases.append(case("%s_read%d" % (fx, i), og, og.export_csr(), [r], (4, 6, 2)))
            a = og.astar_align(r, O.Costs(4, 6, 2), O.H_MINGAP, True)
            og.add_alignment(str(i), r, a["alignment"])
    g, (qseq, qoff) = W.scaled_linearish(120, 8, 4, 6, 130)
    cases.append(case("linearish_120", O.OracleGraph.from_csr(g.as_dict()), g.as_dict(),
                      [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(6)], (4, 6, 2)))
    poa = W.LayeredPOA(n_layers=30, width=4, indeg=4, seed=5)
    cases.append(case("layered_30x4", O.OracleGraph.from_csr(poa.graph.as_dict()), poa.graph.as_dict(),
                      poa.queries(4, length=0), (4, 6, 2)))
    pg = W.PangenomePOA(ref_len=200, n_hap=5, p_snp=0.02, p_indel=0.01, max_indel=5, seed=4)
    cases.append(case("msa_200x5", O.OracleGraph.from_csr(pg.graph.as_dict()), pg.graph.as_dict(),
                      pg.queries(4, length=90), (4, 6, 2)))
    b = GraphBuilder()
    b.add_path(np.frombuffer(b"ACGTACGTAC", np.uint8))
    g = b.finish()
    cases.append(case("edge_cases", O.OracleGraph.from_csr(g.as_dict()), g.as_dict(),
                      [b"", b"A", b"T", b"ACGTACGTAC", b"AC", b"GGGGGGGGGGGGGGGG", b"ACGTACGTACACGTACGTAC"], (4, 6, 2)))
    with open(os.path.join(HERE, "golden_vectors.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py (oracle A* restatement; not a capture of the Rust reference)",
                       cases=cases), f, separators=(",", ":"))
    print("wrote %d cases, %d queries" % (len(cases), sum(len(c["queries"]) for c in cases)))


if __name__ == "__main__":
    main()

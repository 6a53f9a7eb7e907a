#!/usr/bin/env python3
"""Time the two-piece replay (poa_align_batch_2piece_ex, mode EXACT) on configs[1]-shaped work under the CLI's example
costs (`poasta align -g 6,24 -e 2,1`: Affine2PieceMinGapCost, pruning on); optionally check a sample against the oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poasta_amd import aligner as E   # noqa: E402
from poasta_amd import workloads as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--heuristic", default="mingap")
    a = ap.parse_args()
    g, (qseq, qoff) = W.config2(n_queries=a.queries)
    cls = E.Affine2PieceMinGapCost if a.heuristic == "mingap" else E.Affine2PieceDijkstra
    al = E.PoastaAligner(cls(E.GapAffine2Piece(4, 2, 6, 1, 24)), mode="exact")
    best = None
    for _ in range(a.reps):
        t0 = time.time()
        res = al.align_batch(g, qseq=qseq, qoff=qoff)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, res)
    dt, res = best
    cells = int(g.n) * int((np.diff(qoff.astype(np.int64)) + 1).sum())
    out = dict(queries=a.queries, wall_s=round(dt, 4), ms_exact=res.stats["ms_exact"], ms_traceback=res.stats["ms_traceback"],
               gcells_per_s=round(cells / dt / 1e9, 3), gcells_per_s_kernels=round(cells / ((res.stats["ms_exact"] + res.stats["ms_traceback"]) * 1e-3) / 1e9, 3),
               flagged=int((res.flags != 0).sum()), flag_bits=int(np.bitwise_or.reduce(res.flags)), chunks=res.stats["n_chunks"],
               queued_mean=float(res.search_counters[:, 0].mean()), visited_mean=float(res.search_counters[:, 1].mean()),
               pruned_mean=float(res.search_counters[:, 2].mean()), live_max=int(res.search_counters[:, 3].max()),
               cells_per_query=cells // a.queries)
    if a.check:
        from oracle import pyoracle as O
        og = O.OracleGraph.from_csr(g.as_dict())
        k = min(a.check, a.queries)
        t0 = time.time()
        with O.two_piece(24, 1):
            A = og.astar_batch(qseq[:int(qoff[k])], qoff[:k + 1], O.Costs(4, 6, 2), O.H_MINGAP if a.heuristic == "mingap" else O.H_DIJKSTRA, True, threads=8)
        out["oracle_s_per_query_8_threads"] = round((time.time() - t0) / k, 5)
        out["identical"] = int(sum(res.raw_alignment(i) == O.batch_alignment(A, i) and int(res.score[i]) == int(A["score"][i]) for i in range(k)))
        out["checked"] = k
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput of the two-piece dense pass (poa_align_batch_2piece) on config 2 (GPU box); scores checked on a sample."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poasta_amd import aligner, workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=2000)
ap.add_argument("--check", type=int, default=32)
args = ap.parse_args()
g, (qseq, qoff) = W.config2(n_queries=args.queries)
al = aligner.PoastaAligner(aligner.Affine2PieceDijkstra(aligner.GapAffine2Piece(4, 2, 6, 1, 24)))   # poasta align -g 6,24 -e 2,1
al.align_batch(g, qseq=qseq[:int(qoff[8])], qoff=qoff[:9])
t0 = time.time()
res = al.align_batch(g, qseq=qseq, qoff=qoff)
dt = time.time() - t0
st = res.stats
out = dict(queries=args.queries, wall_s=round(dt, 3), ms_forward=round(st["ms_forward"], 2), ms_traceback=round(st["ms_traceback"], 2),
           chunks=st["n_chunks"], gcells_per_s_kernels=round(st["cells"] / ((st["ms_forward"] + st["ms_traceback"]) * 1e-3) / 1e9, 2),
           gcells_per_s_forward=round(st["cells"] / (st["ms_forward"] * 1e-3) / 1e9, 2), flagged=int((res.flags != 0).sum()))
if args.check:
    from oracle import pyoracle as O  # test-side checker
    og = O.OracleGraph.from_csr(g.as_dict())
    k = args.check
    with O.two_piece(24, 1):
        D = og.dense_batch(qseq[:int(qoff[k])], qoff[:k + 1], O.Costs(4, 6, 2), threads=16)
    out.update(checked=k, score_equal=int((D["score"] == res.score[:k]).sum()),
               alignment_equal=int(sum(res.raw_alignment(i) == O.batch_alignment(D, i) for i in range(k))))
print(json.dumps(out))

#!/usr/bin/env python3
"""Dense-mode throughput on the other BASELINE.json configs (run on the GPU box).  Size-independent checks
only: every score finite, alignment consumes the whole query, scores of a small sample equal the oracle's dense
restatement.  Prints one JSON line per run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poasta_amd import aligner, workloads as W  # noqa: E402
from poasta_amd.graph import pack_queries  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="5")
ap.add_argument("--queries", type=int, default=0)
ap.add_argument("--oracle-sample", type=int, default=0)
ap.add_argument("--runs", type=int, default=2)
args = ap.parse_args()
t0 = time.time()
if args.config == "2":
    g, (qseq, qoff) = W.config2(n_queries=args.queries or 10000)
elif args.config == "5":
    g, (qseq, qoff) = W.config5(n_queries=args.queries or 2000)
elif args.config == "long":
    # long reads on a linear-ish graph: 10 kbp x 10 752 rows (config 2's family, ten times longer)
    g, (qseq, qoff) = W.scaled_linearish(10000, 500, 250, args.queries or 1500, 10000)
elif args.config == "4":
    g, (qseq, qoff) = W.config4(n_queries=args.queries or 5000)
else:
    raise SystemExit("unknown config")
t_gen = time.time() - t0
n = len(qoff) - 1
costs = aligner.GapAffine(4, 2, 6)
rb = aligner.ResidentBatch(g, qseq, qoff)
best = None
for _ in range(args.runs):
    t0 = time.time()
    rb.run(costs)
    st = rb.stats()
    dt = time.time() - t0
    if best is None or dt < best[0]:
        best = (dt, st)
dt, st = best
res = rb.fetch()
cells = st["cells"]
out = dict(config=args.config, rows=g.n, queries=n, gen_s=round(t_gen, 1), wall_s=round(dt, 4), gcells_per_s=round(cells / dt / 1e9, 2),
           ms_forward=round(st["ms_forward"], 2), ms_traceback=round(st["ms_traceback"], 2), chunks=st["n_chunks"],
           cells=int(cells), finite=int((res.score != 0xFFFFFFFF).sum()), flagged=int((res.flags != 0).sum()),
           score_sum=int(res.score.astype(np.uint64).sum()))
# every alignment must consume the whole query exactly once, in order
bad = 0
for i in range(min(n, 64)):
    a = res.raw_alignment(i)
    qp = [q for (_, q) in a if q != 0xFFFFFFFF]
    L = int(qoff[i + 1] - qoff[i])
    if qp != sorted(qp) or len(set(qp)) != len(qp) or (qp and qp[-1] != L - 1):
        bad += 1
out["bad_alignments_in_first_64"] = bad
if args.oracle_sample:
    from oracle import pyoracle as O  # test-side checker
    og = O.OracleGraph.from_csr(g.as_dict())
    k = args.oracle_sample
    D = og.dense_batch(qseq[:int(qoff[k])], qoff[:k + 1], O.Costs(4, 6, 2), threads=16)
    out["oracle_sample"] = k
    out["oracle_score_equal"] = int((D["score"] == res.score[:k]).sum())
    out["oracle_alignment_equal"] = int(sum(res.raw_alignment(i) == O.batch_alignment(D, i) for i in range(k)))
print(json.dumps(out), flush=True)
rb.close()

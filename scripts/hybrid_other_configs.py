#!/usr/bin/env python3
"""Hybrid mode (dense pass + replay of the reference's search) on samples of the other BASELINE.json configs (GPU box):
alignments against the restated reference, replay time."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O   # test-side checker
from poasta_amd import aligner, workloads as W

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--n5", type=int, default=8)
ap.add_argument("--n4", type=int, default=2)
args = ap.parse_args()
print("generating", flush=True)
work = []
if args.n5:
    work.append(("configs[4] sample: 20 002 rows, in-degree 4, %d x 5 kbp" % args.n5, W.config5(n_queries=args.n5)))
if args.n4:
    work.append(("configs[3] sample: 56 101 rows, %d x 10 kbp, Global" % args.n4, W.config4(n_queries=args.n4)))
for name, (g, (qseq, qoff)) in work:
    print("running", name, flush=True)
    n = len(qoff) - 1
    rb = aligner.ResidentBatch(g, qseq, qoff)
    cfg = aligner.make_config("hybrid", queue_entries_per_cell=0.25)
    t0 = time.time(); rb.run(aligner.GapAffine(4, 2, 6), None, cfg); st = rb.stats(); dt = time.time() - t0
    res = rb.fetch()
    print("gpu done in %.2f s, ms_exact %.1f; oracle ..." % (dt, st["ms_exact"]), flush=True)
    og = O.OracleGraph.from_csr(g.as_dict())
    A = og.astar_batch(qseq, qoff, O.Costs(4, 6, 2), O.H_MINGAP, True, threads=16)
    same = sum(res.raw_alignment(i) == O.batch_alignment(A, i) for i in range(n) if A["status"][i] == 0)
    sc = rb.search_counters()
    print(json.dumps(dict(workload=name, queries=n, seconds=round(dt, 3), ms_exact=round(st["ms_exact"], 1), replayed=int(res.stats["n_exact"]),
                          overflow=int(((res.flags & 0x40) != 0).sum()), scores_equal=int((res.score == A["score"]).sum()),
                          alignments_identical=int(same), reference_ok=int((A["status"] == 0).sum()),
                          pops_mean=float(sc[:, 0].mean()), steps_mean=float(sc[:, 3].mean()))), flush=True)
    rb.close()

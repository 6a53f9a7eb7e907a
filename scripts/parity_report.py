#!/usr/bin/env python3
"""Parity report on a BASELINE.json config (run on the GPU box): HIP path vs the oracle's A* restatement.
Test infrastructure (uses oracle/): prints and writes gpurun_out/parity_<config>.json."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from poasta_amd import aligner, workloads as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2")
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    if args.config == "2":
        g, (qseq, qoff) = W.config2(n_queries=args.queries)
    elif args.config == "5s":
        poa = W.LayeredPOA(n_layers=1000, width=4, indeg=4)
        from poasta_amd.graph import pack_queries
        g, (qseq, qoff) = poa.graph, pack_queries(poa.queries(args.queries, length=1000))
    elif args.config == "4s":
        poa = W.PangenomePOA(ref_len=3000, n_hap=16)
        from poasta_amd.graph import pack_queries
        g, (qseq, qoff) = poa.graph, pack_queries(poa.queries(args.queries, length=1000))
    else:
        raise SystemExit("unknown config")
    n = len(qoff) - 1
    al = aligner.PoastaAligner(aligner.AffineMinGapCost(aligner.GapAffine(4, 2, 6)))
    t0 = time.time()
    res = al.align_batch(g, qseq=qseq, qoff=qoff)
    t_gpu = time.time() - t0
    og = O.OracleGraph.from_csr(g.as_dict())
    t0 = time.time()
    A = og.astar_batch(qseq, qoff, O.Costs(4, 6, 2), O.H_MINGAP, True, threads=args.threads, want_counters=True)
    t_cpu = time.time() - t0
    ok = A["status"] == 0
    score_eq = int(((res.score == A["score"]) & ok).sum())
    certified = int(((res.flags == 0) & ok).sum())
    score_uncertain = int((((res.flags & (2 | 8)) != 0) & ok).sum())
    same = cert_same = 0
    for i in range(n):
        if not ok[i]:
            continue
        eq = res.raw_alignment(i) == O.batch_alignment(A, i)
        same += eq
        if res.flags[i] == 0:
            cert_same += eq
    out = dict(config=args.config, rows=g.n, queries=n, ref_panics=int((~ok).sum()), score_equal=score_eq,
               score_uncertain_flagged=score_uncertain, certified_unique=certified, certified_identical=cert_same,
               alignment_identical=int(same), gpu_call_s=round(t_gpu, 3), cpu_astar_s=round(t_cpu, 3),
               cpu_threads=args.threads, visited_per_query=float(A["counters"][:, 1].mean()),
               stats=res.stats)
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_config%s.json" % args.config), "w") as f:
        json.dump(out, f, indent=1)
    assert cert_same == certified, "a certified alignment differs from the A* restatement"
    assert score_eq + score_uncertain >= int(ok.sum()), "an unflagged score differs"


if __name__ == "__main__":
    main()

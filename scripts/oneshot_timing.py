#!/usr/bin/env python3
"""Wall time of the one-shot host-buffer entry point (poa_align_batch) on config 2 — the PCIe-inclusive figure."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poasta_amd import aligner, workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g, (qseq, qoff) = W.config2(n_queries=n)
al = aligner.PoastaAligner(aligner.AffineMinGapCost(aligner.GapAffine(4, 2, 6)))
out = []
for rep in range(4):
    t0 = time.time()
    r = al.align_batch(g, qseq=qseq, qoff=qoff)
    dt = time.time() - t0
    out.append(dict(call=rep, wall_s=round(dt, 4), ms_h2d=round(r.stats["ms_h2d"], 2), ms_d2h=round(r.stats["ms_d2h"], 2),
                    ms_forward=round(r.stats["ms_forward"], 2), ms_traceback=round(r.stats["ms_traceback"], 2)))
print(json.dumps(out))

#!/usr/bin/env python3
"""Two resident batches on two streams: does the traceback of one step hide under the forward pass of the next?  (GPU box)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from poasta_amd import aligner, workloads as W

g, (qseq, qoff) = W.config2(n_queries=10000)
costs = aligner.GapAffine(4, 2, 6)
cells = int(g.n * (np.diff(qoff).astype(np.int64) + 1).sum())
out = {}
for depth in (1, 2, 3):
    ws = 40 << 30
    batches = [aligner.ResidentBatch(g, qseq, qoff, workspace_bytes=ws) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for k in range(2 * depth):
        batches[k % depth].run(costs, streams[k % depth].cuda_stream)
    torch.cuda.synchronize()
    steps = 24
    t0 = time.perf_counter()
    for k in range(steps):
        batches[k % depth].run(costs, streams[k % depth].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = [b.stats() for b in batches]
    out[depth] = dict(ms_per_step=round(dt / steps * 1e3, 3), gcells=round(cells * steps / dt / 1e9, 1),
                      fwd_ms=round(sum(s["ms_forward"] for s in st) / steps, 3), tb_ms=round(sum(s["ms_traceback"] for s in st) / steps, 3),
                      chunks=st[0]["n_chunks"])
    res = [b.fetch() for b in batches]
    for r in res[1:]:
        assert np.array_equal(r.score, res[0].score) and np.array_equal(r.pairs, res[0].pairs)
    for b in batches:
        b.close()
print(json.dumps(out))

#!/usr/bin/env python3
"""Per-read latency of a sequential POA build (BASELINE.json configs[0]'s shape: `poasta align`, one read after the other, the
graph updated after each) on the engine — handle refresh (poa_graph_update), alignment call with its stages, graph update —
for the 10-read fixture and a 200-read synthetic build, beside the restated reference's CPU time for the same builds.
GPU box:  python scripts/sequential_poa_latency.py > gpurun_out/sequential_poa_latency.json"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O
from poasta_amd import workloads as W

BIN = os.path.join(ROOT, "poasta_amd", "poasta_align_amd")


def synthetic(n_reads=200, length=300, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    base = W.ACGT[rng.integers(0, 4, size=length)]
    recs = []
    for i in range(n_reads):
        s = W.mutate(rng, base, 0.03, 0.01, 0.01)
        recs.append(("r%d" % i, s.tobytes().decode()))
    return recs


def run(recs, tag):
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "reads.fa")
        with open(fa, "w") as f:
            for name, seq in recs:
                f.write(">%s\n%s\n" % (name, seq))
        tsv, out = os.path.join(td, "t.tsv"), os.path.join(td, "o.fa")
        t0 = time.perf_counter()
        p = subprocess.run([BIN, "align", "--timing", tsv, "-o", out, fa], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr[-400:]}
        rows = [ln.rstrip("\n").split("\t") for ln in open(tsv)][1:]
        cols = ["us_graph_refresh", "us_align_call", "us_h2d", "us_dense_kernels", "us_replay", "us_d2h", "us_graph_update"]
        a = np.array([[float(x) for x in r[3:]] for r in rows])
        warm = a[1:] if len(a) > 2 else a          # (the first call allocates the workspace)
        msa = open(out).read()
    t0 = time.perf_counter()
    g, _scores = O.sequential_poa(recs)
    cpu = time.perf_counter() - t0
    try:
        ofa = g.to_fasta()
        same = (ofa if isinstance(ofa, str) else "".join(">%s\n%s\n" % (n, r) for n, r in ofa)) == msa
    except Exception:
        same = None
    return {"build": tag, "reads": len(recs), "aligned_reads": len(rows), "mean_read_len": round(float(np.mean([len(s) for _, s in recs])), 1),
            "final_graph_nodes": int(rows[-1][2]) if rows else None, "wall_s_whole_run_incl_process_start": round(wall, 3),
            "per_read_us_median": {c: round(float(np.median(warm[:, i])), 1) for i, c in enumerate(cols)},
            "per_read_us_mean": {c: round(float(np.mean(warm[:, i])), 1) for i, c in enumerate(cols)},
            "first_call_us": {c: round(float(a[0, i]), 1) for i, c in enumerate(cols)},
            "graph_refresh_share_of_align_call": round(float(np.median(warm[:, 0]) / max(np.median(warm[:, 1]), 1e-9)), 4),
            "oracle_cpu_whole_build_s": round(cpu, 4), "oracle_cpu_per_read_us": round(cpu / max(len(recs) - 1, 1) * 1e6, 1),
            "msa_equal_to_oracle": same}


if __name__ == "__main__":
    fx = O.read_fasta(os.path.join(ROOT, "tests", "golden", "test2_from_abpoa.fa"))
    out = {"fixture_test2_from_abpoa": run(fx, "tests/golden/test2_from_abpoa.fa (10 reads)"),
           "synthetic_200x300": run(synthetic(), "200 reads x ~300 bp, 3% sub / 1% ins / 1% del of one random sequence, seed 7"),
           "note": "hybrid mode (the graph update consumes the reference's own tie-breaks); us_align_call = poa_align_batch_ex wall time for one read: "
                   "upload + dense pass + replay of the read if flagged + download (the stage columns are HIP-event / host timings inside it)"}
    print(json.dumps(out, indent=1))

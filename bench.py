#!/usr/bin/env python3
"""Headline benchmark: Gcells/s of the gap-affine POA hot path on BASELINE.json configs[1]
(synthetic 1000-node linear-ish POA graph, 10 000 queries x 1 kbp per GPU, penalties 4/6/2, Global).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed over RCCL when N > 1; launched by torch.distributed.run).
A "step" is one pass of the hot path over the rank's resident query batch: forward (score planes)
+ traceback + compaction, results left in HBM.  Queries are independent, so ranks share nothing on
the data path (weak scaling: 10 000 queries per GPU == configs[2] at N = 8); after the timed region
the results are gathered to rank 0 with one RCCL all_gather (the path's only collective).

Rank 0 prints ONE JSON line (see the task contract) with `roofline` (dominant kernel = the forward
pass; algorithmic bytes = 12 B per cell) and, at N = 1, `cpu_baseline` (the
oracle's restated reference CPU path — A* + min-gap heuristic + pruning, lasagna-shaped thread
pool — on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL = 12.0      # three u32 planes written once per cell
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=10000, help="queries per GPU")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: every rank uses cuda:0 and the gather runs over gloo (not a measurement)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse else dev  # where the tensors of the collectives live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from poasta_amd import aligner, workloads

    # ---- workload: configs[1] per GPU; rank r owns queries [r*Q, (r+1)*Q) of the seeded stream ----
    t0 = time.time()
    poa = workloads.LinearishPOA(seed=1)
    graph = poa.graph
    qs = poa.queries(args.queries, length=args.length, seed=2, first=rank * args.queries)
    from poasta_amd.graph import pack_queries
    qseq, qoff = pack_queries(qs)
    t_gen = time.time() - t0
    costs = aligner.GapAffine(4, 2, 6)  # (mismatch, extend, open): the CLI defaults of the reference

    stream = torch.cuda.current_stream().cuda_stream
    batch = aligner.ResidentBatch(graph, qseq, qoff, device=local_rank)  # inputs now resident in HBM
    n_rows = graph.n
    cells_rank = int(sum(n_rows * (int(qoff[i + 1] - qoff[i]) + 1) for i in range(args.queries)))
    bases_rank = int(qoff[-1])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run(costs, stream)
    barrier()
    batch.stats()  # drop warm-up timings
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        batch.run(costs, stream)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    st = batch.stats()  # HIP events recorded on `stream` around every kernel of the timed steps
    elapsed_t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(elapsed_t, op=dist.ReduceOp.MAX)
    elapsed_max = float(elapsed_t.item())

    # ---- result gather (the path's only exchange step): fixed-stride records to every rank ----
    res = batch.fetch(want_pairs=(rank == 0))
    gather_ms = None
    flagged_total = int((res.flags != 0).sum())
    score_sum = int(res.score.astype(np.uint64).sum())
    if dist is not None and not args.no_gather:
        ptrs = batch.device_results()
        rec = torch.stack([torch.from_numpy(res.score.astype(np.int64)), torch.from_numpy(res.flags.astype(np.int64)),
                           torch.from_numpy((res.pair_off[1:] - res.pair_off[:-1]).astype(np.int64))], dim=1).to(cdev)
        out = [torch.empty_like(rec) for _ in range(world)]
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        dist.all_gather(out, rec)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        allrec = torch.cat(out).cpu().numpy()
        flagged_total = int((allrec[:, 1] != 0).sum())
        score_sum = int(allrec[:, 0].sum())
        del ptrs

    if rank == 0:
        total_cells = cells_rank * world
        total_bases = bases_rank * world
        ms_per_step = elapsed_max / args.steps * 1e3
        gcells = total_cells * args.steps / elapsed_max / 1e9
        launches = max(st["n_forward_launches"], 1)
        avg_launch_ms = st["ms_forward"] / launches
        cells_per_launch = cells_rank * st["n_runs"] / launches
        achieved = ALG_BYTES_PER_CELL * cells_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("workload") == "config2" and tj.get("queries") == args.queries:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if os.environ.get("POA_PLANES") == "32":
            kernel_name = "poa_forward_kernel<4, unsigned int>"
        elif os.environ.get("POA_COMPACT") == "0" or os.environ.get("POA_PACKED") == "0":
            kernel_name = "poa_forward_kernel<2, unsigned short>"
        else:
            kernel_name = "poa_forward_px_kernel<true>" if os.environ.get("POA_PX") != "0" else "poa_forward_packed_kernel<2>"
        line = {
            "metric": "Gcells/sec (aligned bases/sec in config), gap-affine POA alignment, 1k-node POA x 10k x 1 kbp queries per GPU",
            "value": round(gcells, 3), "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32" if os.environ.get("POA_PLANES") == "32" else "u16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic 1000-node linear-ish POA graph (900 backbone + 50 SNP "
                                   "bubbles + 25 two-node insertion branches, seed 1), %d queries x %d bp per GPU (2%% sub, 1%% ins, "
                                   "1%% del, seed 2), Global, mismatch 4 / open 6 / extend 2" % (args.queries, args.length),
                       "rows": n_rows, "queries_per_gpu": args.queries, "query_len": args.length,
                       "cells_per_step": total_cells, "aligned_bases_per_sec": round(total_bases * args.steps / elapsed_max, 1),
                       "step": "forward planes + traceback + compaction, inputs and results resident in HBM",
                       "arithmetic": "saturating packed u16 min-plus (exact: the optimal score is bounded by 3.8 k here; results are u32)",
                       "flagged_queries": flagged_total, "score_checksum": score_sum,
                       "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                       "workload_gen_s": round(t_gen, 2), "plane_chunks": st["n_chunks"]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": kernel_name, "avg_launch_ms": round(avg_launch_ms, 3),
                         "launches_timed": launches, "cells_per_launch": int(cells_per_launch),
                         "alg_bytes_per_cell": ALG_BYTES_PER_CELL,
                         "note": "B_alg is fixed at 12 B/cell (the reference's three u32 score planes, SURVEY.md 8d); the "
                                 "engine stores u16 M + 4-bit codes + partial D (~3 B/cell, see traffic), so frac > 1 is by "
                                 "construction: the kernel is VALU-issue bound (DESIGN.md 6), not HBM bound",
                         "traceback_ms_per_step": round(st["ms_traceback"] / max(st["n_runs"], 1), 3)},
        }
        if world == 1 and args.cpu_sample > 0:
            sample = qs[:min(args.cpu_sample, len(qs))]
            line["cpu_baseline"], A = cpu_baseline(graph, sample, n_rows)
            line["bit_exact_check"] = bit_exact_check(graph, sample, res, A, costs)
        print(json.dumps(line), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(graph, qs, n_rows):
    """The oracle's restated reference CPU path on the host cores (reported baseline, not the target)."""
    from oracle import pyoracle
    from poasta_amd.graph import pack_queries
    threads = max(1, min(16, os.cpu_count() or 1))
    og = pyoracle.OracleGraph.from_csr(graph.as_dict())
    qseq, qoff = pack_queries(qs)
    og.astar_batch(qseq[:int(qoff[8])], qoff[:9], threads=threads, want_pairs=False)  # warm (bubble index, pages)
    t0 = time.perf_counter()
    A = og.astar_batch(qseq, qoff, pyoracle.Costs(4, 6, 2), pyoracle.H_MINGAP, True, threads=threads,
                       want_pairs=True, want_counters=True)
    dt = time.perf_counter() - t0
    cells = sum(n_rows * (int(qoff[i + 1] - qoff[i]) + 1) for i in range(len(qs)))
    return ({"value": round(cells / dt / 1e9, 4), "unit": "Gcells/s (matrix-equivalent)", "cores": threads,
            "kind": "port",
            "sample": "%d queries of the same workload, A* + min-gap heuristic + superbubble pruning + backtrace "
                      "(C++ restatement of the reference CPU path), %d threads, %.2f s wall" % (len(qs), threads, dt),
            "aligned_bases_per_sec": round(int(qoff[-1]) / dt, 1),
            "visited_states_per_sec": round(float(A["counters"][:, 1].sum()) / dt, 1),
            "host_cpus": os.cpu_count()}, A)


def bit_exact_check(graph, qs, dense_res, A, costs):
    """Outside the timed region: the same sample against the restated reference — scores of the timed dense pass, and
    alignments of the hybrid mode (dense pass + replay of the reference's search where the dense pass found ties)."""
    from oracle import pyoracle
    from poasta_amd import aligner
    n = len(qs)
    ok = A["status"] == 0
    score_equal = int(sum(1 for i in range(n) if ok[i] and int(dense_res.score[i]) == int(A["score"][i])))
    dense_identical = int(sum(1 for i in range(n) if ok[i] and dense_res.raw_alignment(i) == pyoracle.batch_alignment(A, i)))
    k = min(n, 512)
    al = aligner.PoastaAligner(aligner.AffineMinGapCost(costs), mode="hybrid")
    al.align_batch(graph, qs[:8])  # warm (replay workspace)
    t0 = time.perf_counter()
    hy = al.align_batch(graph, qs[:k])
    dt = time.perf_counter() - t0
    hybrid_identical = int(sum(1 for i in range(k) if ok[i] and int(hy.score[i]) == int(A["score"][i]) and
                               hy.raw_alignment(i) == pyoracle.batch_alignment(A, i)))
    return {"against": "restated reference (A*, min-gap, pruning), %d queries" % n, "dense_scores_equal": score_equal,
            "dense_alignments_identical": dense_identical, "dense_flagged_as_tied": int((dense_res.flags[:n] != 0).sum()),
            "hybrid_queries": k, "hybrid_alignments_identical": hybrid_identical, "hybrid_seconds": round(dt, 3),
            "hybrid_replayed": int(hy.stats["n_exact"])}


if __name__ == "__main__":
    main()

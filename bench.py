#!/usr/bin/env python3
"""Headline benchmark: Gcells/s of the gap-affine POA hot path on BASELINE.json configs[1]
(synthetic 1000-node linear-ish POA graph, 10 000 queries x 1 kbp per GPU, penalties 4/6/2, Global).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no launcher in the environment this script starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py` itself (as a child, before anything touches the
GPU) and exits with its code; under the driver's own torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.
A "step" is one pass of the hot path over the rank's resident query batch: forward (score planes) + traceback +
compaction, results left in HBM.  Queries are independent, so ranks share nothing on the data path (weak scaling:
10 000 queries per GPU == configs[2] at N = 8); after the timed region the results travel once, device to device:
records to every rank (all_gather), alignment pairs to rank 0 (gather) — poasta_amd/dist.py, RCCL over xGMI.

Rank 0 prints ONE JSON line (task contract) with
  roofline      the forward kernel against the ceiling that binds it (VALU issue, measured by
                profiles/microbench/valu_issue.hip), with the HBM figures beside it: `alg_*` = SURVEY.md §8(d)'s fixed
                12 B/cell accounting, `traffic*` = PMC bytes;
  like_for_like the same step with u32 score planes (12 real bytes per cell: where §8(d)'s accounting is physical);
  value_incl_d2h  the step including the device->host copy of scores, flags and pairs (SURVEY.md §8(d)'s wall time);
  steps_in_flight the same step with three resident batches round robin on three HIP streams (traceback under the next forward pass);
  bit_exact     (N = 1) the hybrid mode — dense pass + replay of the reference's search for every query the dense pass
                could not certify — over the whole batch: throughput and how many alignments equal the restated
                reference's;
  cpu_baseline  (N = 1) the restated reference CPU path on the host's cores, all physical cores and one thread.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL = 12.0      # three u32 planes written once per cell
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
N_SIMD = 1024                  # 256 CUs x 4


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_json(*rel):
    try:
        with open(os.path.join(ROOT, *rel)) as f:
            return json.load(f)
    except Exception:
        return None


FORWARD_KERNEL_SOURCES = ("poa_forward_px.hpp", "poa_forward_packed.hpp", "poa_kernels.hpp", "poa_graph.hpp", "poa_graph.cpp")


def forward_kernel_source_hash():
    """(sha256[:16], file list) of the sources the forward / traceback kernels are compiled from: ties the PMC counters
    committed under profiles/ to the kernels of this tree."""
    import hashlib
    h = hashlib.sha256()
    for name in FORWARD_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "poasta_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16], list(FORWARD_KERNEL_SOURCES)


def valu_ceiling():
    """(wave-instructions per second one SIMD issues for the packed-u16 / permute / DPP instructions the forward kernels are
    made of, at 8 resident waves; source file).  Round 3's micro-benchmark (profiles/microbench/valu_issue.hip) starts all
    waves of a launch together and takes the rate over the span first-wave-in .. last-wave-out of the loops on the device-wide
    counter; it equals the rate from the launch time (0.567e9 per SIMD = one packed-u16 instruction per 4.2 cycles).  The
    waves' OWN timers give 1.0e9 — round 2's ceiling — because a SIMD issues from its oldest ready wave first: the waves
    finish one after the other, a wave's own elapsed time averages 9/16 of the span.  Only the round-2 file at hand: its
    launch-derived rate."""
    for name in ("valu_issue_r03.json", "valu_issue_r02.json"):
        mb = _load_json("profiles", "microbench", name)
        if not mb:
            continue
        rates = []
        for r in mb["results"]:
            if r["op"] in ("v_pk_add_u16 clamp", "v_pk_min_u16", "v_perm_b32", "v_mov_b32_dpp row_shr:1"):
                w8 = r["waves_per_simd"]["8"]
                rates.append(w8["simd_instr_per_s_span"] if "simd_instr_per_s_span" in w8 else w8["simd_instr_per_s_launch"])
        if rates:
            return sum(rates) / len(rates), "profiles/microbench/" + name
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=10000, help="queries per GPU")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="queries timed on the CPU baseline at all cores (-1 = the whole batch, 0 = skip baseline and bit_exact)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--model", default="affine", choices=["affine", "2piece"],
                    help="2piece: an extra, not the headline — the two-piece affine dense pass (poa_align_batch_2piece, SURVEY.md 8(f) row 3) "
                         "on the configs[1] shape under the CLI's example costs -g 6,24 -e 2,1; one GPU; prints its own JSON line")
    ap.add_argument("--no-extras", action="store_true", help="skip like_for_like / value_incl_d2h / bit_exact (profiling runs)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU (or none for the spawn path): every rank uses cuda:0 and the gather runs "
                         "over gloo on CPU tensors (not a measurement)")
    args = ap.parse_args()

    if args.model == "2piece":
        return bench_two_piece(args)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # no launcher: become one.  Nothing has touched the GPU yet (torch is not even imported).
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from poasta_amd import aligner, workloads
    from poasta_amd import dist as pdist
    from poasta_amd.graph import pack_queries

    # ---- workload: configs[1] per GPU; rank r owns queries [r*Q, (r+1)*Q) of the seeded stream ----
    t0 = time.time()
    poa = workloads.LinearishPOA(seed=1)
    graph = poa.graph
    qs = poa.queries(args.queries, length=args.length, seed=2, first=rank * args.queries)
    qseq, qoff = pack_queries(qs)
    t_gen = time.time() - t0
    costs = aligner.GapAffine(4, 2, 6)  # (mismatch, extend, open): the CLI defaults of the reference

    stream = torch.cuda.current_stream().cuda_stream
    batch = aligner.ResidentBatch(graph, qseq, qoff, device=local_rank)  # inputs now resident in HBM
    n_rows = graph.n
    cells_rank = int(n_rows * (np.diff(qoff).astype(np.int64) + 1).sum())
    bases_rank = int(qoff[-1])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(b, k, cfg=None, collective=True):
        """k steps on batch b between barriers -> (seconds max over ranks, engine stats of those steps).
        collective=False: this rank alone (the extras below must not be able to hang the job if one rank cannot run them)."""
        if collective:
            barrier()
        else:
            torch.cuda.synchronize()
        b.stats()
        t_start = time.perf_counter()
        for _ in range(k):
            b.run(costs, stream, cfg)
        if collective:
            barrier()
        else:
            torch.cuda.synchronize()
        el = time.perf_counter() - t_start
        st_ = b.stats()  # HIP events recorded on `stream` around every kernel of the timed steps
        if dist is not None and collective:
            t = torch.tensor([el], dtype=torch.float64, device=torch.device("cpu") if args.rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, st_

    for _ in range(args.warmup):
        batch.run(costs, stream)
    elapsed_max, st = timed(batch, args.steps)

    # ---- result gather (the path's only exchange step), from the engine's result buffers in HBM ----
    gather_ms = None
    d_score, d_flags, d_off, d_pairs = pdist.device_result_tensors(batch, dev)
    flagged_total = int((d_flags != 0).sum().item())
    score_sum = int((d_score.to(torch.int64) & 0xFFFFFFFF).sum().item())
    pairs_at_root = int(d_pairs.shape[0])
    if dist is not None and not args.no_gather:
        if args.rehearse:
            d_score, d_flags, d_off, d_pairs = (t.cpu() for t in (d_score, d_flags, d_off, d_pairs))
        barrier()
        g0 = time.perf_counter()
        g_score, g_flags, g_np, g_pairs = pdist.gather_result_tensors(d_score, d_flags, d_off, d_pairs)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        flagged_total = int((g_flags != 0).sum().item())
        score_sum = int(g_score.sum().item())
        if rank == 0:
            assert int(g_np.sum().item()) == int(g_pairs.shape[0])
            pairs_at_root = int(g_pairs.shape[0])
        del g_score, g_flags, g_np, g_pairs
    del d_score, d_flags, d_off, d_pairs

    # ---- device -> host copy of one step's results (SURVEY.md §8(d) counts it; `value` does not) ----
    incl_d2h = None
    if not args.no_extras:
        batch.fetch(want_pairs=True, pinned=True)   # page-locks the host buffers once (not part of a step)
        barrier()
        t1 = time.perf_counter()
        batch.run(costs, stream)
        res = batch.fetch(want_pairs=True, pinned=True)   # synchronises, then copies score / flags / pair_off / pairs to the host
        dt = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if args.rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res = _detach(res)   # the pinned buffers are reused by later fetches
        incl_d2h = {"value": round(cells_rank * world / dt / 1e9, 3), "unit": "Gcells/s", "ms_per_step": round(dt * 1e3, 3),
                    "d2h_ms": round(res.stats["ms_d2h"], 3), "d2h_bytes": int(res.pairs.nbytes + res.score.nbytes + res.flags.nbytes + res.pair_off.nbytes),
                    "note": "one step + poa_batch_fetch into page-locked host buffers, single shot"}
    else:
        res = batch.fetch(want_pairs=(rank == 0))

    # ---- steps in flight: three resident batches on three HIP streams, round robin (rank-local, not part of `value`):
    #      the traceback of a step (latency bound, few issue slots) runs under the forward pass of the next ----
    piped = None
    if not args.no_extras and not args.rehearse and "POA_PLANES" not in os.environ:
        try:
            depth = 3
            # (each batch gets what the compact layout needs, not the u32-sized default, so that three fit beside the first)
            ws = int(3.2 * cells_rank) + (1 << 30)
            pb = [aligner.ResidentBatch(graph, qseq, qoff, device=local_rank, workspace_bytes=ws) for _ in range(depth)]
            ps = [torch.cuda.Stream(device=dev) for _ in range(depth)]
            for k in range(2 * depth):
                pb[k % depth].run(costs, ps[k % depth].cuda_stream)
            torch.cuda.synchronize()
            n_p = max(12, args.steps)
            t1 = time.perf_counter()
            for k in range(n_p):
                pb[k % depth].run(costs, ps[k % depth].cuda_stream)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t1
            chunks_p = pb[0].stats()["n_chunks"]
            same = True
            for b_ in pb:
                r_ = b_.fetch(want_pairs=True)
                same = same and np.array_equal(r_.score, res.score) and np.array_equal(r_.pairs, res.pairs) and np.array_equal(r_.flags, res.flags)
                b_.close()
            piped = {"in_flight": depth, "steps": n_p, "value": round(cells_rank * n_p / dtp / 1e9, 3), "unit": "Gcells/s (this rank)",
                     "ms_per_step": round(dtp / n_p * 1e3, 3), "chunks_per_step": chunks_p, "results_equal_the_timed_steps": bool(same),
                     "note": "the same step, three resident batches round robin on three HIP streams: a step's traceback overlaps the next "
                             "step's forward pass (how a driver with batches to spare would call poa_batch_run)"}
        except Exception as exc:
            piped = {"error": "%s: %s" % (type(exc).__name__, exc)}
        from poasta_amd import _lib as _plib
        _plib.lib().poa_release_cache()   # the workspaces just closed go back to the device before the next extra allocates

    # ---- the same step with u32 planes: 12 real bytes per cell (rank-local, not part of `value`) ----
    like = None
    if not args.no_extras and not args.rehearse and "POA_PLANES" not in os.environ:   # (a rehearsal's ranks share one GPU: 2 x 123 GB of u32 planes do not fit)
        os.environ["POA_PLANES"] = "32"
        try:
            b32 = aligner.ResidentBatch(graph, qseq, qoff, device=local_rank)
            b32.run(costs, stream)
            el32, st32 = timed(b32, 2, collective=False)
            l32 = max(st32["n_forward_launches"], 1)
            ms32 = st32["ms_forward"] / l32
            cpl32 = cells_rank * st32["n_runs"] / l32
            like = {"dtype": "u32", "value": round(cells_rank * 2 / el32 / 1e9, 3), "unit": "Gcells/s (this rank)", "ms_per_step": round(el32 / 2 * 1e3, 3),
                    "forward_launch_ms": round(ms32, 3), "hbm_bytes_written_per_cell": 12,
                    "hbm_achieved_GBps": round(12.0 * cpl32 / (ms32 * 1e-3) / 1e9, 1),
                    "hbm_frac": round(12.0 * cpl32 / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "note": "POA_PLANES=32: M, I, D stored as u32 = the reference's VisitedCellAffine; here the 12 B/cell of SURVEY.md 8(d) are real stores"}
            b32.close()
        except Exception as exc:   # an extra never fails the bench (e.g. no room for a second, 123 GB workspace)
            like = {"error": "%s: %s" % (type(exc).__name__, exc)}
        finally:
            del os.environ["POA_PLANES"]

    if rank == 0:
        total_cells = cells_rank * world
        total_bases = bases_rank * world
        ms_per_step = elapsed_max / args.steps * 1e3
        gcells = total_cells * args.steps / elapsed_max / 1e9
        launches = max(st["n_forward_launches"], 1)
        avg_launch_ms = st["ms_forward"] / launches
        cells_per_launch = cells_rank * st["n_runs"] / launches
        alg_achieved = ALG_BYTES_PER_CELL * cells_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        kc = _load_json("profiles", "kernel_counters.json") or {}
        src_hash = forward_kernel_source_hash()[0]
        stale = kc.get("csrc_sha16") != src_hash
        same = kc.get("workload") == "config2" and kc.get("queries") == args.queries and "POA_PLANES" not in os.environ and not stale
        traffic = kc.get("hbm_bytes_per_launch") if same else None
        valu_insts = kc.get("sq_insts_valu_per_launch") if same else None
        ceil_simd, ceil_src = valu_ceiling()
        if os.environ.get("POA_PLANES") == "32":
            kernel_name = "poa_forward_kernel<4, unsigned int>"
        elif os.environ.get("POA_COMPACT") == "0" or os.environ.get("POA_PACKED") == "0":
            kernel_name = "poa_forward_kernel<2, unsigned short>"
        else:
            kernel_name = "poa_forward_px_kernel<2>" if os.environ.get("POA_PX") != "0" else "poa_forward_packed_kernel<2>"
        roof = {"kernel": kernel_name, "avg_launch_ms": round(avg_launch_ms, 3), "launches_timed": launches,
                "cells_per_launch": int(cells_per_launch),
                "alg_bytes_per_cell": ALG_BYTES_PER_CELL, "alg_achieved_GBps": round(alg_achieved, 1),
                "alg_over_hbm_peak_ratio": round(alg_achieved / HBM_PEAK_GBPS, 4),   # not a fraction: the engine stores ~2.4 B/cell of the 12 charged
                "traffic": traffic,
                "hbm_frac_physical": None if not traffic else round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "counters": {"file": "profiles/kernel_counters.json", "csrc_sha16_of_counters": kc.get("csrc_sha16"), "csrc_sha16_of_this_tree": src_hash,
                             "usable": bool(same),
                             "why_not": None if same else ("the committed PMC counters were taken from other forward-kernel sources than this tree's "
                                                           "(re-run profiles/run_rocprof.sh)" if stale else "other workload / plane layout than the counters'")},
                "traceback_ms_per_step": round(st["ms_traceback"] / max(st["n_runs"], 1), 3)}
        if valu_insts and ceil_simd:
            ach = valu_insts / (avg_launch_ms * 1e-3) / 1e9
            peak = ceil_simd * N_SIMD / 1e9
            salu_insts = kc.get("sq_insts_salu_per_launch")
            roof.update({"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s", "frac": round(ach / peak, 4),
                         "peak_source": ceil_src,
                         "insts_per_cell": {"valu": round(valu_insts * 64 / cells_per_launch, 3),
                                            "salu": None if not salu_insts else round(salu_insts * 64 / cells_per_launch, 3)},
                         "note": "Instruction issue binds this kernel: achieved = SQ_INSTS_VALU per launch (PMC, profiles/kernel_counters.json) / "
                                 "launch time; peak = measured issue rate of the packed-u16 / permute / DPP instructions it consists of, "
                                 "8 waves per SIMD (profiles/microbench).  The SALU instructions and wait states of a row take issue slots "
                                 "too (occupancy sweep, profiles/r02b_px_mf4: 0.625 us per wave and 1024-cell row at saturation), so fewer "
                                 "instructions per cell make the kernel faster and this fraction smaller.  HBM beside it: alg_* is "
                                 "SURVEY.md 8(d)'s fixed 12 B/cell (the engine stores ~2.4 B/cell, so that ratio exceeds 1 by construction), "
                                 "traffic* are PMC bytes"})
        else:
            roof.update({"bound": "hbm", "achieved": round(alg_achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg_achieved / HBM_PEAK_GBPS, 4),
                         "note": "12 B/cell accounting of SURVEY.md 8(d): no PMC instruction count usable for this run (see counters.why_not), "
                                 "so the instruction-issue fraction and the physical HBM fraction are not reported"})
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
                       "step": "dense mode: forward planes + traceback + compaction, inputs and results resident in HBM; scores are "
                               "the reference's, alignments are certified-or-flagged (see bit_exact for the identical mode)",
                       "arithmetic": "saturating packed u16 min-plus (exact: the optimal score is bounded by 3.8 k here; results are u32)",
                       "flagged_queries": flagged_total, "score_checksum": score_sum,
                       "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                       "gather": None if gather_ms is None else "records all_gather + pairs gather to rank 0 from the engine's HBM result buffers (%d pairs at rank 0)" % pairs_at_root,
                       "workload_gen_s": round(t_gen, 2), "plane_chunks": st["n_chunks"]},
            "roofline": roof,
        }
        if like is not None:
            line["like_for_like"] = like
        if incl_d2h is not None:
            line["value_incl_d2h"] = incl_d2h
        if piped is not None:
            line["steps_in_flight"] = piped
        if world == 1 and args.cpu_sample != 0 and not args.no_extras:
            n_cpu = len(qs) if args.cpu_sample < 0 else min(args.cpu_sample, len(qs))
            line["cpu_baseline"], A = cpu_baseline(graph, qs[:n_cpu], n_rows)
            line["bit_exact"] = bit_exact(batch, costs, stream, res, A, n_cpu, cells_rank, flagged_total)
            # what `value` is and is not, at the top level: the dense step returns the reference's scores and a co-optimal
            # alignment; the mode whose alignments ARE the reference's is bit_exact
            be = line["bit_exact"]
            line["alignments_identical_fraction"] = round(be["dense_mode_alignments_identical"] / max(n_cpu, 1), 4)
            line["bit_exact_value"] = be["value"]
            line["bit_exact_alignments_identical_fraction"] = round(be["alignments_identical"] / max(n_cpu, 1), 4)
            line["bit_exact_vs_cpu_baseline"] = round(be["value"] / line["cpu_baseline"]["value"], 2) if line["cpu_baseline"]["value"] else None
            # (the main batch holds its replay workspace for 10 000 queries: released first, or the second batch would run in the
            # memory that is left — several chunks, each as long as its longest search)
            batch.close()
            aligner._lib.lib().poa_release_cache()
            line["bit_exact_unpadded"] = bit_exact_unpadded(poa, graph, costs, stream, local_rank, args.queries, n_rows)
        print(json.dumps(line), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _detach(res):
    """Own copies of a BatchResult's arrays (ResidentBatch.fetch(pinned=True) hands out the batch's reusable buffers)."""
    from poasta_amd import aligner
    return aligner.BatchResult(res.score.copy(), res.pairs.copy(), res.pair_off.copy(), res.flags.copy(), res.stats)


def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None   # cgroup CPU quota of this job in cores (the GPU boxes give a one-GPU job a share of the host)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = round(q / per, 2)
        except (OSError, ValueError):
            pass
    return model, (len(phys) or usable), usable, quota


def cpu_baseline(graph, qs, n_rows):
    """The oracle's restated reference CPU path (A* + min-gap heuristic + superbubble pruning + backtrace, one aligner per
    thread over a shared graph and bubble index: `lasagna`'s shape) on the host: all physical cores this process may use,
    and one thread.  A reported baseline, not the target."""
    from oracle import pyoracle
    from poasta_amd.graph import pack_queries
    model, phys, usable, quota = _cpu_info()
    max_threads = max(1, min(phys, usable))
    og = pyoracle.OracleGraph.from_csr(graph.as_dict())
    qseq, qoff = pack_queries(qs)
    og.astar_batch(qseq[:int(qoff[8])], qoff[:9], threads=min(max_threads, 8), want_pairs=False)  # warm (bubble index, pages)
    # the port's thread pool does not scale linearly (allocator, NUMA): take the thread count that is fastest on a slice
    sweep = {}
    ns = min(len(qs), 2048)
    cand = sorted({max_threads, max(1, max_threads // 2), max(1, max_threads // 4), min(16, max_threads)}, reverse=True)
    for t in cand:
        t0 = time.perf_counter()
        og.astar_batch(qseq[:int(qoff[ns])], qoff[:ns + 1], pyoracle.Costs(4, 6, 2), pyoracle.H_MINGAP, True, threads=t, want_pairs=True)
        sweep[t] = time.perf_counter() - t0
    threads = min(sweep, key=sweep.get)
    t0 = time.perf_counter()
    A = og.astar_batch(qseq, qoff, pyoracle.Costs(4, 6, 2), pyoracle.H_MINGAP, True, threads=threads,
                       want_pairs=True, want_counters=True)
    dt = time.perf_counter() - t0
    cells = int(n_rows * (qoff[1:] - qoff[:-1] + 1).astype("int64").sum())
    # one thread, on a bounded slice (about 10 ms per query)
    n1 = min(len(qs), 384)
    t1 = time.perf_counter()
    og.astar_batch(qseq[:int(qoff[n1])], qoff[:n1 + 1], pyoracle.Costs(4, 6, 2), pyoracle.H_MINGAP, True, threads=1, want_pairs=True)
    dt1 = time.perf_counter() - t1
    cells1 = int(n_rows * (qoff[1:n1 + 1] - qoff[:n1] + 1).astype("int64").sum())
    return ({"value": round(cells / dt / 1e9, 4), "unit": "Gcells/s (matrix-equivalent)", "cores": threads, "kind": "port",
             "sample": "%d queries of the same workload, A* + min-gap heuristic + superbubble pruning + backtrace (C++ restatement of "
                       "the reference CPU path, -O3), %d threads (fastest of the sweep below; %d physical cores usable), %.2f s wall" % (len(qs), threads, max_threads, dt),
             "thread_sweep_gcells": {str(t): round(int(n_rows * (qoff[1:ns + 1] - qoff[:ns] + 1).astype("int64").sum()) / v / 1e9, 4) for t, v in sweep.items()},
             "cpu_model": model, "physical_cores": phys, "usable_cpus": usable, "cgroup_cpu_quota_cores": quota,
             "aligned_bases_per_sec": round(int(qoff[-1]) / dt, 1),
             "visited_states_per_sec": round(float(A["counters"][:, 1].sum()) / dt, 1),
             "single_thread": {"value": round(cells1 / dt1 / 1e9, 5), "unit": "Gcells/s (matrix-equivalent)", "cores": 1,
                               "sample": "%d queries, %.2f s wall" % (n1, dt1), "ms_per_query": round(dt1 / n1 * 1e3, 3)}}, A)


def bit_exact(batch, costs, stream, dense_res, A, n_checked, cells_rank, dense_flagged):
    """Outside the timed region: hybrid mode over the WHOLE resident batch (dense pass, then the replay of the reference's
    search for every query the dense pass flagged), timed, and compared with the restated reference on the queries the
    CPU baseline ran."""
    from oracle import pyoracle
    from poasta_amd import aligner
    cfg = aligner.make_config("hybrid", queue_entries_per_cell=0.25)
    batch.run(costs, stream, cfg)   # warm (replay workspace)
    batch.stats()
    t0 = time.perf_counter()
    batch.run(costs, stream, cfg)
    st = batch.stats()
    dt = time.perf_counter() - t0
    hy = batch.fetch(want_pairs=True)
    ok = A["status"] == 0
    n = n_checked
    dense_identical = int(sum(1 for i in range(n) if ok[i] and dense_res.raw_alignment(i) == pyoracle.batch_alignment(A, i)))
    scores_equal = int(sum(1 for i in range(n) if ok[i] and int(hy.score[i]) == int(A["score"][i])))
    identical = int(sum(1 for i in range(n) if ok[i] and int(hy.score[i]) == int(A["score"][i]) and
                        hy.raw_alignment(i) == pyoracle.batch_alignment(A, i)))
    out = {"mode": "hybrid", "value": round(cells_rank / dt / 1e9, 3), "unit": "Gcells/s", "seconds": round(dt, 4),
           "queries": len(hy.score), "replayed": int(hy.stats["n_exact"]), "flagged_fraction": round(dense_flagged / max(len(hy.score), 1), 4),
           "ms_dense_pass": round(st["ms_forward"] + st["ms_traceback"], 3), "ms_replay": round(st["ms_exact"], 3),
           "checked_against": "restated reference (A*, min-gap, pruning), %d queries" % n,
           "scores_equal": scores_equal, "alignments_identical": identical,
           "dense_mode_alignments_identical": dense_identical,
           "overflow": int(((hy.flags & 0x40) != 0).sum())}
    try:
        sc = batch.search_counters()
        sel = sc[:, 3] > 0
        if sel.any():
            out.update(replay_pops_mean=round(float(sc[sel, 0].mean()), 1), replay_steps_mean=round(float(sc[sel, 3].mean()), 1))
    except Exception:
        pass
    return out


def bench_two_piece(args):
    """Extra: the two-piece affine model's dense pass (forward kernel poa2_forward_kernel + traceback poa2_traceback_kernel) over
    configs[1]'s graph and reads, through the one-shot C entry point (host buffers in, host buffers out; the kernel times are HIP
    events inside it).  roofline: HBM — the five planes M, D1, D2, I1, I2 a traceback of this model reads (20 B/cell as u32, the
    algorithmic unit in the sense of SURVEY.md 8(d); the engine stores them as u16 under the score bound: 10 B/cell physical)."""
    import numpy as np
    from poasta_amd import aligner, workloads
    n = min(args.queries, 8000)
    g, (qseq, qoff) = workloads.config2(n_queries=n, length=args.length)
    al = aligner.PoastaAligner(aligner.Affine2PieceDijkstra(aligner.GapAffine2Piece(4, 2, 6, 1, 24)))   # poasta align -g 6,24 -e 2,1
    al.align_batch(g, qseq=qseq[:int(qoff[8])], qoff=qoff[:9])
    best = None
    for _ in range(max(1, args.warmup) + max(1, min(args.steps, 3))):
        t0 = time.perf_counter()
        res = al.align_batch(g, qseq=qseq, qoff=qoff)
        dt = time.perf_counter() - t0
        if best is None or res.stats["ms_forward"] < best[1]["ms_forward"]:
            best = (dt, res.stats)
    dt, st = best
    cells, ms_f, ms_t = st["cells"], st["ms_forward"], st["ms_traceback"]
    launches = max(st["n_forward_launches"], 1)
    alg = 20.0 * cells / (ms_f * 1e-3) / 1e9
    phys = st["plane_bytes"] / (ms_f * 1e-3) / 1e9
    line = {"metric": "Gcells/sec, two-piece gap-affine POA dense pass (extra), 1k-node POA x %d x 1 kbp queries" % n,
            "value": round(cells / ((ms_f + ms_t) * 1e-3) / 1e9, 3), "unit": "Gcells/s (forward + traceback kernels)", "n_gpus": 1,
            "higher_is_better": True, "dtype": "u16" if st["plane_bytes"] == cells * 10 else "u32", "data": "synthetic", "vs_baseline": None,
            "config": {"workload": "BASELINE.json configs[1] graph and reads (%d queries), GapAffine2Piece(mismatch 4, extend1 2, open1 6, extend2 1, open2 24), Global" % n,
                       "chunks": st["n_chunks"], "one_shot_call_s": round(dt, 3)},
            "forward_gcells_per_s": round(cells / (ms_f * 1e-3) / 1e9, 2), "ms_forward": round(ms_f, 3), "ms_traceback": round(ms_t, 3),
            "roofline": {"kernel": "poa2_forward_kernel", "bound": "hbm", "achieved": round(alg, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg / HBM_PEAK_GBPS, 4), "alg_bytes_per_cell": 20.0, "avg_launch_ms": round(ms_f / launches, 3), "launches_timed": launches,
                         "stored_bytes_per_cell": round(st["plane_bytes"] / cells, 2), "stored_GBps": round(phys, 1), "hbm_frac_physical_stores_only": round(phys / HBM_PEAK_GBPS, 4),
                         "traffic": None,
                         "note": "achieved = 20 B/cell (five u32 planes: the reference's visited cell of this model) x cells / forward time, HIP events inside the "
                                 "call; the engine writes the planes as u16 when the score bound allows (stored_*); PMC traffic: profiles/r03_two_piece/"}}
    # the model's own search replayed (Affine2PieceMinGapCost + pruning: what the CLI runs), on a sample: queries per second and
    # how far its scores lie above the dense optimum
    try:
        k = min(n, 4096)
        ex = aligner.PoastaAligner(aligner.Affine2PieceMinGapCost(aligner.GapAffine2Piece(4, 2, 6, 1, 24)), mode="exact")
        t0 = time.perf_counter()
        rx = ex.align_batch(g, qseq=qseq[:int(qoff[k])], qoff=qoff[:k + 1], want_pairs=False)
        dtx = time.perf_counter() - t0
        dn = al.align_batch(g, qseq=qseq[:int(qoff[k])], qoff=qoff[:k + 1], want_pairs=False)
        line["bit_exact"] = {"mode": "exact (poa_align_batch_2piece_ex: replay of the five-state search, min-gap heuristic, pruning)", "queries": k,
                             "seconds": round(dtx, 3), "ms_replay": round(rx.stats["ms_exact"], 1), "queries_per_s": round(k / dtx, 1),
                             "value": round(int(g.n) * float((np.diff(qoff[:k + 1]).astype(np.int64) + 1).sum()) / dtx / 1e9, 4), "unit": "Gcells/s",
                             "states_visited_mean": round(float(rx.search_counters[:, 1].mean()), 1), "flagged": int((rx.flags != 0).sum()),
                             "scores_above_dense_optimum": int((rx.score > dn.score).sum()), "scores_below_dense_optimum": int((rx.score < dn.score).sum())}
    except Exception as exc:   # (extra: never takes the line down)
        line["bit_exact"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    print(json.dumps(line), flush=True)


def bit_exact_unpadded(poa, graph, costs, stream, device, n_queries, n_rows):
    """The same reads WITHOUT the ~75 random bases that pad them to 1 kbp (BASELINE.json's configs fix the query length; the
    walks through the graph are ~925 bases long): how much of the replayed fraction and of the replay's cost is the padding's —
    the min-gap heuristic prices the padding as one long insertion and the reference's search then sweeps the whole band of
    diagonals it could be opened on."""
    import numpy as np
    from poasta_amd import aligner
    from poasta_amd.graph import pack_queries
    try:
        qs = poa.queries(n_queries, length=0, seed=2)
        qseq, qoff = pack_queries(qs)
        b = aligner.ResidentBatch(graph, qseq, qoff, device=device)
        cells = int(n_rows * (np.diff(qoff).astype(np.int64) + 1).sum())
        b.run(costs, stream)
        d = b.fetch(want_pairs=False)
        cfg = aligner.make_config("hybrid", queue_entries_per_cell=0.25)
        b.run(costs, stream, cfg); b.stats()
        t0 = time.perf_counter()
        b.run(costs, stream, cfg)
        st = b.stats()
        dt = time.perf_counter() - t0
        hy = b.fetch(want_pairs=False)
        out = {"queries": n_queries, "mean_length": round(float(np.diff(qoff).mean()), 1), "flagged_fraction": round(float((d.flags != 0).mean()), 4),
               "value": round(cells / dt / 1e9, 3), "unit": "Gcells/s", "seconds": round(dt, 4), "replayed": int(hy.stats["n_exact"]),
               "ms_replay": round(st["ms_exact"], 3)}
        try:
            sc = b.search_counters()
            sel = sc[:, 3] > 0
            if sel.any():
                out["replay_pops_mean"] = round(float(sc[sel, 0].mean()), 1)
        except Exception:
            pass
        b.close()
        return out
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


if __name__ == "__main__":
    main()

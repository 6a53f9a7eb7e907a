"""Multi-GPU driver plumbing: static query sharding + the path's single collective (result gather).

The reference's only parallelism is `lasagna`'s worker pool over independent reads against one
immutable graph (/root/reference/src/bin/lasagna.rs:246-268).  The same independence is used here:
queries are split into contiguous blocks, one per rank (one process per GPU), the graph is replicated
(KB..MB), and nothing is exchanged during the DP.  After compute, results travel once:
fixed-stride records {score, flags, n_pairs} to every rank with `all_gather`, then the variable-length alignment
pairs to rank 0 with `gather` (sizes are known from the records).  The tensors are the engine's own result buffers in
HBM (`ResidentBatch.device_results`, wrapped without a copy), so with backend "nccl" the exchange is RCCL over xGMI,
device to device; the gloo tests and `bench.py --rehearse` run the same function on CPU tensors.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block of ceil(n/world) queries for `rank` -> (first, count)."""
    per = -(-n_total // world)
    first = min(rank * per, n_total)
    return first, max(0, min(per, n_total - first))


class _DevArray:
    """A device buffer the engine owns, described through __cuda_array_interface__ so that torch aliases it."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_result_tensors(batch, device):
    """score i32[n], flags i32[n], pair_off i64[n+1], pairs i32[pair_off[n], 2] as torch tensors ALIASING the engine's result
    buffers of the last run (valid until the next run / close).  The u32 / u64 values are carried bit for bit in the signed
    types NCCL moves."""
    import torch
    n = batch.n
    ptrs = batch.device_results()
    torch.cuda.synchronize(device)
    if n == 0:
        z32 = torch.zeros(0, dtype=torch.int32, device=device)
        return z32, z32.clone(), torch.zeros(1, dtype=torch.int64, device=device), torch.zeros((0, 2), dtype=torch.int32, device=device)
    score = torch.as_tensor(_DevArray(ptrs["score"], (n,), "<i4"), device=device)
    flags = torch.as_tensor(_DevArray(ptrs["flags"], (n,), "<i4"), device=device)
    pair_off = torch.as_tensor(_DevArray(ptrs["pair_off"], (n + 1,), "<i8"), device=device)
    n_pairs = int(pair_off[n].item())
    if n_pairs:
        pairs = torch.as_tensor(_DevArray(ptrs["pairs"], (n_pairs, 2), "<i4"), device=device)
    else:
        pairs = torch.zeros((0, 2), dtype=torch.int32, device=device)
    return score, flags, pair_off, pairs


def gather_result_tensors(score, flags, pair_off, pairs, group=None, dst=0):
    """Every rank calls this with its shard's results as tensors (all on one device type: HBM for nccl, CPU for gloo).
    Returns on EVERY rank the records of the whole batch in rank order, (score, flags, n_pairs) int64 tensors, and on `dst`
    also the concatenated pairs [sum n_pairs, 2] (None elsewhere).  all_gather (records), then grouped send / recv (pairs)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = score.device
    n_local = int(score.shape[0])
    npairs = (pair_off[1:] - pair_off[:-1]).to(torch.int64)
    # 1) how many queries / pairs each rank holds
    meta = torch.tensor([n_local, int(pairs.shape[0])], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    pcounts = [int(m[1]) for m in metas]
    max_n = max(counts + [1])
    # 2) fixed-stride records to every rank
    rec = torch.zeros((max_n, 3), dtype=torch.int64, device=dev)
    if n_local:
        rec[:n_local, 0] = score.to(torch.int64) & 0xFFFFFFFF
        rec[:n_local, 1] = flags.to(torch.int64) & 0xFFFFFFFF
        rec[:n_local, 2] = npairs
    recs = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec, group=group)
    all_rec = torch.cat([recs[r][:counts[r]] for r in range(world)])
    # 3) alignment pairs to the consumer: one buffer of exactly the batch's pairs at `dst`, every other rank sends its
    #    shard straight into its slice (grouped send / recv — RCCL point to point under "nccl", each peer over its own xGMI
    #    link; the sizes are known from step 1).  No padding to the largest shard, no `world` staging buffers at the root.
    all_pairs = None
    ops = []
    if rank == dst:
        offs = [0]
        for c in pcounts:
            offs.append(offs[-1] + c)
        all_pairs = torch.empty((offs[-1], 2), dtype=torch.int32, device=dev)
        if pcounts[rank]:
            all_pairs[offs[rank]:offs[rank + 1]] = pairs
        for r in range(world):
            if r != dst and pcounts[r]:
                ops.append(dist.P2POp(dist.irecv, all_pairs[offs[r]:offs[r + 1]], r if group is None else dist.get_global_rank(group, r), group))
    elif pcounts[rank]:
        ops.append(dist.P2POp(dist.isend, pairs.contiguous(), dst if group is None else dist.get_global_rank(group, dst), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return all_rec[:, 0], all_rec[:, 1], all_rec[:, 2], all_pairs


def gather_results(score, flags, pair_off, pairs, device=None, group=None):
    """numpy in, numpy out (the whole batch on every rank's records; pairs on every rank for symmetry with the tests):
    (score u32, flags u32, pair_off u64, pairs u32[., 2])."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
    s = t(np.asarray(score, np.uint32), np.int32)
    f = t(np.asarray(flags, np.uint32), np.int32)
    o = t(np.asarray(pair_off, np.uint64), np.int64)
    p = t(np.asarray(pairs, np.uint32).reshape(-1, 2), np.int32)
    g_score, g_flags, g_np, g_pairs = gather_result_tensors(s, f, o, p, group=group, dst=0)
    # hand the pairs to every rank (tests read them on rank 0; the symmetric call keeps the helper simple)
    world = dist.get_world_size(group)
    total = torch.tensor([int(g_np.sum())], dtype=torch.int64, device=dev)
    buf = g_pairs if dist.get_rank(group) == 0 else torch.zeros((int(total.item()), 2), dtype=torch.int32, device=dev)
    if world > 1:
        dist.broadcast(buf, src=0, group=group)
    g_off = np.zeros(len(g_np) + 1, np.uint64)
    g_off[1:] = np.cumsum(g_np.cpu().numpy())
    return (g_score.cpu().numpy().astype(np.uint32), g_flags.cpu().numpy().astype(np.uint32), g_off,
            buf.cpu().numpy().view(np.uint32).reshape(-1, 2))

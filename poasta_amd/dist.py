"""Multi-GPU driver plumbing: static query sharding + the path's single collective (result gather).

The reference's only parallelism is `lasagna`'s worker pool over independent reads against one
immutable graph (/root/reference/src/bin/lasagna.rs:246-268).  The same independence is used here:
queries are split into contiguous blocks, one per rank (one process per GPU), the graph is replicated
(KB..MB), and nothing is exchanged during the DP.  After compute, results travel once:
fixed-stride records {score, flags, n_pairs} with `all_gather`, then the variable-length alignment
pairs with a padded `all_gather` (sizes are known from the records).  With backend "nccl" this is
RCCL over xGMI; tests run the same code on "gloo".
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block of ceil(n/world) queries for `rank` -> (first, count)."""
    per = -(-n_total // world)
    first = min(rank * per, n_total)
    return first, max(0, min(per, n_total - first))


def gather_results(score, flags, pair_off, pairs, device=None, group=None):
    """All ranks call this with their shard's results (numpy arrays as returned by BatchResult).
    Returns (score, flags, pair_off, pairs) of the whole batch in rank order on EVERY rank
    (all_gather keeps the call symmetric; rank 0 is the consumer in bench.py)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    n_local = len(score)
    npairs = (np.asarray(pair_off[1:], np.int64) - np.asarray(pair_off[:-1], np.int64))
    # 1) how many queries / pairs each rank holds
    meta = torch.tensor([n_local, int(npairs.sum())], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    pcounts = [int(m[1]) for m in metas]
    max_n, max_p = max(counts + [1]), max(pcounts + [1])
    # 2) fixed-stride records
    rec = torch.zeros((max_n, 3), dtype=torch.int64, device=dev)
    if n_local:
        rec[:n_local, 0] = torch.from_numpy(np.asarray(score, np.int64)).to(dev)
        rec[:n_local, 1] = torch.from_numpy(np.asarray(flags, np.int64)).to(dev)
        rec[:n_local, 2] = torch.from_numpy(npairs).to(dev)
    recs = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec, group=group)
    # 3) alignment pairs (padded to the largest shard)
    pbuf = torch.full((max_p, 2), -1, dtype=torch.int64, device=dev)
    if len(pairs):
        pbuf[:len(pairs)] = torch.from_numpy(np.asarray(pairs, np.int64)).to(dev)
    pbufs = [torch.zeros_like(pbuf) for _ in range(world)]
    dist.all_gather(pbufs, pbuf, group=group)
    all_rec = torch.cat([recs[r][:counts[r]] for r in range(world)]).cpu().numpy()
    all_pairs = torch.cat([pbufs[r][:pcounts[r]] for r in range(world)]).cpu().numpy()
    g_score = all_rec[:, 0].astype(np.uint32)
    g_flags = all_rec[:, 1].astype(np.uint32)
    g_off = np.zeros(len(all_rec) + 1, np.uint64)
    g_off[1:] = np.cumsum(all_rec[:, 2])
    return g_score, g_flags, g_off, all_pairs.astype(np.uint32)

"""Multi-GPU sharding of sample / edge batches.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
States are independent, so the data path has NO collective: rank r evaluates the contiguous index range
shard_range(n, r, world).  The only exchanges are
  * all-gather of the packed validity bit mask (n/8 bytes in total), or of the per-edge verdicts, and
  * all-reduce of the few scalars of mvk_stats (sum of counts, min/max with arg via a gathered table).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items, aligned to 64 so packed-mask words never straddle ranks."""
    words = (n + 63) // 64
    per = (words + world - 1) // world
    lo = min(n, rank * per * 64)
    hi = min(n, (rank + 1) * per * 64)
    return lo, hi


def shard_words(n, world):
    return ((n + 63) // 64 + world - 1) // world


def gather_masks(local_mask, n, group=None):
    """local_mask: int64 tensor [shard_words] with this rank's packed flags.  Returns the full mask
    [ceil(n/64)] on every rank (one all-gather of n/8 bytes)."""
    world = dist.get_world_size(group)
    per = shard_words(n, world)
    buf = torch.zeros(per, dtype=torch.int64, device=local_mask.device)
    buf[: local_mask.numel()] = local_mask
    full = torch.empty(world * per, dtype=torch.int64, device=local_mask.device)
    dist.all_gather_into_tensor(full, buf, group=group)
    return full[: (n + 63) // 64]


def shard_edges(n_edges, rank, world):
    """Contiguous shard [lo, hi) of a batch of edges (config c4): edges shard by edge, never by step, so an
    edge's first_invalid needs no cross-rank reduction."""
    per = (n_edges + world - 1) // world
    return min(n_edges, rank * per), min(n_edges, (rank + 1) * per)


def gather_edge_verdicts(local_ok, local_first, n_edges, group=None):
    """local_ok uint8 [hi-lo], local_first int32 [hi-lo] of this rank's shard_edges range.  Returns (ok, first)
    for all n_edges on every rank: one all-gather of 5 bytes per edge."""
    world = dist.get_world_size(group)
    per = (n_edges + world - 1) // world
    dev = local_ok.device
    buf = torch.zeros(per, 2, dtype=torch.int32, device=dev)
    k = local_ok.numel()
    buf[:k, 0] = local_ok.to(torch.int32)
    buf[:k, 1] = local_first
    full = torch.empty(world * per, 2, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(full, buf, group=group)
    return full[:n_edges, 0].to(torch.uint8), full[:n_edges, 1].contiguous()


_MIN_KEYS = ("min_dist", "min_margin", "min_thrust", "min_rp_position_margin")
_MAX_KEYS = ("max_thrust",)


def reduce_stats(local, offset, device, group=None):
    """Combine per-rank mvk_stats dicts (indices local to the shard starting at `offset`).
    Counts are summed; extrema keep their global sample index (ties -> smallest index)."""
    world = dist.get_world_size(group)
    counts = torch.tensor([local["n"], local["valid_count"], local["band_count"]], dtype=torch.int64, device=device)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    keys = _MIN_KEYS + _MAX_KEYS
    row = torch.tensor([[local[k], float(local[k + "_index"] + offset if local[k + "_index"] >= 0 else -1)] for k in keys],
                       dtype=torch.float64, device=device)
    table = torch.empty((world,) + tuple(row.shape), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(table, row.unsqueeze(0), group=group)
    table = table.cpu().numpy()
    out = {"n": int(counts[0]), "valid_count": int(counts[1]), "band_count": int(counts[2])}
    for j, k in enumerate(keys):
        vals, idx = table[:, j, 0], table[:, j, 1]
        vals = np.where(idx < 0, np.inf if k in _MIN_KEYS else -np.inf, vals)
        best = vals.min() if k in _MIN_KEYS else vals.max()
        cand = idx[vals == best]
        out[k] = float(best)
        out[k + "_index"] = int(cand.min()) if len(cand) else -1
    return out

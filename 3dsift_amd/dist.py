"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The extraction path shards by independent volumes (BASELINE.json configs[4]): no collective on the data
path.  The only exchange step is BEFORE matching: every rank contributes its N_r x 768 descriptors and
N_r x 3 coordinates (ragged), all ranks receive all of them (all-gather over xGMI), and the ordered
volume pairs (i, j), i != j, are dealt round-robin to ranks for `enhancedMatch`.
Works on CPU tensors with the gloo backend (tests/test_dist_cpu.py, world_size 2)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT. Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"), **kw)
    return rank, world


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_ragged(t):
    """t: [n_r, d] on this rank (n_r differs per rank) -> list of world tensors [n_i, d].
    Two collectives: counts (world x int64), then rows padded to max(n_i)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:c] for b, c in zip(bufs, counts)]


def ordered_pairs(world):
    """All ordered (ref, tar) volume pairs, i != j: 56 for 8 volumes."""
    return [(i, j) for i in range(world) for j in range(world) if i != j]


def my_pairs(rank, world):
    """Round-robin deal of the ordered pairs: 7 per rank for 8 volumes."""
    return [p for k, p in enumerate(ordered_pairs(world)) if k % world == rank]


def match_pairs(descs, xyzs, match_fn, rank, world):
    """run match_fn on the ordered pairs dealt to `rank` of `world`, from the gathered per-volume lists"""
    return {(i, j): match_fn(descs[i], xyzs[i], descs[j], xyzs[j]) for (i, j) in my_pairs(rank, world)}


def allpairs_match(desc, xyz, match_fn):
    """BASELINE configs[4] matching step: all-gather every rank's (ragged) descriptors and coordinates, then run
    `match_fn(desc_i, xyz_i, desc_j, xyz_j)` (enhancedMatch) on this rank's share of the ordered volume pairs.
    Returns {(i, j): result of match_fn} for the pairs dealt to this rank (all pairs without a process group)."""
    descs = allgather_ragged(desc)
    xyzs = allgather_ragged(xyz)
    world = len(descs)
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return match_pairs(descs, xyzs, match_fn, rank, world)

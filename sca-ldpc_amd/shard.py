"""Sharding of independent trials over ranks: the only multi-GPU
structure this path has.  One process per GPU; rank r owns a contiguous block of
global trial indices; trial inputs are seeded by GLOBAL index, so results do not depend
on the number of ranks; no collective inside the decode loop, one gather of the
per-trial results at the end (RCCL `all_gather` on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def trial_range(total, rank, world):
    """Contiguous, balanced split: first (total % world) ranks get one extra trial."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_results(local, total, rank, world, device=None):
    """All-gather per-trial result vectors (uint8 / int32 / ...), ragged shards allowed.
    `local`: 1-D numpy array of this rank's results.  Returns the length-`total` array
    in global trial order on every rank.  world == 1 -> no communication."""
    local = np.ascontiguousarray(local)
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist

    sizes = [trial_range(total, r, world)[1] - trial_range(total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank]
    mx = max(sizes)
    pad = np.zeros(mx, dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy()[: sizes[r]] for r, o in enumerate(outs)])


def success_checksum(success):
    """One number that pins WHICH trials failed, not just how many: the sum of the global indices of the first 1000
    failed trials.  Trial inputs depend on (seed, global index) only, so a sweep's checksum must not change with the
    number of ranks (bench.py prints it; `--workload hqc128_mc --trials 1000000 --gpus N` is BASELINE config 5)."""
    return int(np.flatnonzero(np.asarray(success) == 0)[:1000].sum())

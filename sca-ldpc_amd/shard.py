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


class Collectives:
    """The process groups of a multi-GPU run, set up so that the transport cannot cost the run its result.

    The path has ONE exchange: a gather of per-trial flags after the timed region (a few MB for a million
    trials).  `gloo` is brought up first and always -- it carries the barriers around the timed region (host-side: the
    device is synchronised before and after) and is the fallback for the gathers.  RCCL (`nccl`) is then TRIED as a second
    group: one all_reduce of a counter with a deadline, on every rank; the ranks agree over gloo (MIN) whether it worked
    everywhere.  If it did, the gathers travel over RCCL / xGMI from device memory; if it did not -- init raised, the
    probe raised, timed out or returned a wrong sum, on ANY rank -- every rank uses gloo, in the same process (a
    process that has touched the GPU is never re-executed), and `backend` / `note` say so on the bench line.
    SCALDPC_FORCE_NCCL_FAILURE=1 forces the fallback (the CPU / gloo test of this logic)."""

    def __init__(self, rank, world, device=None, want="nccl", probe_timeout_s=90.0):
        self.rank, self.world, self.device = int(rank), int(world), device
        self.backend, self.note, self.group = "none", "", None
        self._abandoned = False
        if self.world <= 1:
            return
        import datetime
        import os
        import time

        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.backend = "gloo"
        if want != "nccl":
            self.note = f"gloo requested ({want})"
            return
        ok, why, grp = 1, "", None
        try:
            if device is None or not torch.cuda.is_available():
                raise RuntimeError("no GPU visible to this rank")
            # (the communicator is built inside the first collective, a blocking call: the group's own timeout bounds that
            # part, the deadline below the collective itself)
            grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(120.0, 2 * probe_timeout_s)))
            if os.environ.get("SCALDPC_FORCE_NCCL_FAILURE") == "1":
                raise RuntimeError("forced by SCALDPC_FORCE_NCCL_FAILURE=1")
            t = torch.ones(1, device=device, dtype=torch.int32)
            work = dist.all_reduce(t, group=grp, async_op=True)
            deadline = time.monotonic() + probe_timeout_s
            while not work.is_completed():
                if time.monotonic() > deadline:
                    self._abandoned = True  # (a collective that never completes cannot be cancelled: never touch the group again)
                    raise TimeoutError(f"first RCCL all_reduce not complete after {probe_timeout_s:.0f} s")
                time.sleep(0.02)
            torch.cuda.synchronize(device)
            if int(t.item()) != self.world:
                raise RuntimeError(f"first RCCL all_reduce returned {int(t.item())}, expected {self.world}")
        except Exception as ex:  # noqa: BLE001 -- whatever the transport throws, the run goes on over gloo
            ok, why = 0, f"{type(ex).__name__}: {ex}"
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # (gloo) every rank, or none
        if int(flag.item()) == 1:
            self.backend, self.group = "nccl", grp
        else:
            self.note = "nccl failed, gathers over gloo: " + (why or "another rank's RCCL probe failed")
            self.note += f" [HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset')}]"

    @property
    def abandoned_collective(self):
        """An RCCL collective was left in flight (probe timeout): leave through os._exit, teardown may never return."""
        return self._abandoned

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist

            dist.barrier()  # gloo: host-side

    def gather(self, local, total):
        """gather_results over the group that works (device tensors over RCCL, host tensors over gloo)."""
        return gather_results(local, total, self.rank, self.world, device=self.device if self.backend == "nccl" else None,
                              group=self.group if self.backend == "nccl" else None)

    def gather_scalars(self, x):
        """One float per rank, in rank order, on every rank (per-rank timings for rank 0's line)."""
        if self.world <= 1:
            return [float(x)]
        return [float(v) for v in gather_results(np.array([x], dtype=np.float64), self.world, self.rank, self.world,
                                                 device=self.device if self.backend == "nccl" else None,
                                                 group=self.group if self.backend == "nccl" else None)]

    def close(self):
        if self.world > 1 and not self._abandoned:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()


def gather_results(local, total, rank, world, device=None, group=None):
    """All-gather per-trial result vectors (uint8 / int32 / ...), ragged shards allowed.
    `local`: 1-D numpy array of this rank's results.  Returns the length-`total` array
    in global trial order on every rank.  world == 1 -> no communication.  `group`: the process group to use
    (default: the default group)."""
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
    dist.all_gather(outs, t, group=group)
    return np.concatenate([o.cpu().numpy()[: sizes[r]] for r, o in enumerate(outs)])


def success_checksum(success):
    """One number that pins WHICH trials failed, not just how many: the sum of the global indices of the first 1000
    failed trials.  Trial inputs depend on (seed, global index) only, so a sweep's checksum must not change with the
    number of ranks (bench.py prints it; `--workload hqc128_mc --trials 1000000 --gpus N` is BASELINE config 5)."""
    return int(np.flatnonzero(np.asarray(success) == 0)[:1000].sum())

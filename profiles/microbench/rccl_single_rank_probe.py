"""Does RCCL itself come up in this environment?  One rank, backend nccl (= RCCL on ROCm): process group, an all_reduce, the
all_gather of shard.gather_results' shape, a barrier.  (Two ranks cannot share the box's one GPU under RCCL, so this is as far
as a one-GPU box goes; the N > 1 data path is rehearsed over gloo.)"""
import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(8, dtype=torch.float64, device="cuda")
dist.all_reduce(x)
u = torch.arange(5, dtype=torch.uint8, device="cuda")
outs = [torch.empty_like(u)]
dist.all_gather(outs, u)
dist.barrier()
torch.cuda.synchronize()
print("rccl single-rank probe ok: backend", dist.get_backend(), "all_reduce", x.tolist()[:3], "all_gather", outs[0].tolist(),
      "HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "%.2f s" % (time.time() - t0))
dist.destroy_process_group()

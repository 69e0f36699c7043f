"""400 x {build, decode (host buffers, small-call path), append rows twice, batched early-exit decode, Monte-Carlo run,
close} + q-ary into_llr: device free memory AND the library allocator's own accounting (scaldpc_debug_live_blocks:
device / pinned-host blocks live handles still own) before and after.  Output kept under profiles/r03/leak_check.log."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import hqc_instance, S
bp = importlib.import_module("sca-ldpc_amd.bp"); lib = importlib.import_module("sca-ldpc_amd._lib"); qary = importlib.import_module("sca-ldpc_amd.qary")
H, Hin, probs, msg, y = hqc_instance(1499, 9, 600, 7, 0.03, 70, seed=5)
N = 1499
def graph(r):
    rp = Hin.row_ptr[: r + 1]
    cols = np.concatenate([Hin.col_idx[: rp[-1]].reshape(r, -1), N + np.arange(r, dtype=np.int32)[:, None]], axis=1)
    return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * cols.shape[1], cols.reshape(-1))
def cycle():
    d = bp.bp_decoder(graph(300), max_iter=20, bp_method="product_sum", channel_probs=np.concatenate([probs[:N], probs[N:N+300]]))
    d.decode_batch(np.concatenate([msg[:1, :N], msg[:1, N:N+300]], axis=1))
    for r0, r1 in ((300, 400), (400, 600)):
        g = graph(r1); rp = g.row_ptr[r0:r1+1].astype(np.int64)
        d.append_rows((rp - rp[0]).astype(np.int32), g.col_idx[rp[0]:rp[-1]], N + r1, probs[N+r0:N+r1])
        d.decode_batch(np.concatenate([msg[:1, :N], msg[:1, N:N+r1]], axis=1))
    d.decode_batch(msg, early_exit=True)
    d.mc_hqc_run(128, omega=7, eps=0.03, seed=1)
    d.close()
    qary.into_llr(np.random.RandomState(1).dirichlet(np.ones(3), size=1000).astype(np.float32))
b0 = lib.live_blocks()
for _ in range(20): cycle()
lib.trim(); torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]; b1 = lib.live_blocks()
for _ in range(400): cycle()
b2 = lib.live_blocks()
lib.trim(); torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]; b3 = lib.live_blocks()
import resource
print("device free before %.1f MB after %.1f MB delta %.2f MB; host maxrss %.0f MB" % (f0/1e6, f1/1e6, (f0-f1)/1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1e3))
print("allocator accounting  at start        :", b0)
print("                      after 20 cycles :", b1)
print("                      after 420 cycles:", b2, "(parked blocks are the cache, returned by trim)")
print("                      after trim      :", b3)
live = ("device_blocks", "device_bytes", "pinned_blocks", "pinned_bytes")
assert all(b3[k] == b0[k] for k in live) and b3["idle_blocks"] == 0, "blocks left behind"
print("no device or pinned-host block is left behind")

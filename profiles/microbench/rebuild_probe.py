import importlib, json, os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp"); lib = importlib.import_module("sca-ldpc_amd._lib")
rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
N, omega, eps = 17669, 66, 0.05
Rmax = 6000
_, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"], R=Rmax)
rng = np.random.RandomState(1)
y = np.zeros((1, N), dtype=np.uint8); y[0, rng.choice(N, omega, replace=False)] = 1
checks = Hin.syndrome(y) ^ (rng.rand(1, Rmax) < eps).astype(np.uint8)
W1 = Hin.col_idx.size // Rmax + 1
cols = np.concatenate([Hin.col_idx.reshape(Rmax, -1), N + np.arange(Rmax, dtype=np.int32)[:, None]], axis=1)
graph = lambda r: S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * W1, cols[:r].reshape(-1))
probs = lambda r: np.concatenate([np.full(N, omega / N), np.full(r, eps)])
msg = lambda r: np.concatenate([np.zeros((1, N), dtype=np.uint8), checks[:, :r]], axis=1)
for rep in range(4):
    for step in (50, 100):
        sizes = list(range(4000, Rmax + 1, step))
        G = {r: graph(r) for r in sizes}; P = {r: probs(r) for r in sizes}; M = {r: msg(r) for r in sizes}
        tc, td, tx = [], [], []
        for r in sizes[1:]:
            t0 = time.perf_counter(); d = bp.bp_decoder(G[r], max_iter=100, bp_method="product_sum", channel_probs=P[r]); t1 = time.perf_counter()
            d.decode_batch(M[r], early_exit=True); t2 = time.perf_counter(); d.close(); t3 = time.perf_counter()
            tc.append(t1 - t0); td.append(t2 - t1); tx.append(t3 - t2)
        f = lambda a: "median %.3f max %.3f first5 %s" % (np.median(a) * 1e3, np.max(a) * 1e3, np.round(np.array(a[:5]) * 1e3, 2))
        print("rep", rep, "step", step, "| create", f(tc), "| decode", f(td), "| close", f(tx), "| blocks", lib.live_blocks())

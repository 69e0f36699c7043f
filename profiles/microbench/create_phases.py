import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp")
rows = json.load(open("tests/golden/hqc_first_rows.json"))
H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"], R=4000)
N, omega = 17669, 66
probs = np.concatenate([np.full(N, omega / N), np.full(4000, 0.05)])
x = np.concatenate([np.zeros(N, np.uint8), np.random.RandomState(0).randint(0, 2, 4000).astype(np.uint8)])
for i in range(4):
    t0 = time.perf_counter(); d = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs); t1 = time.perf_counter()
    d.decode_batch(x[None], early_exit=True); t2 = time.perf_counter(); d.decode_batch(x[None], early_exit=True); t3 = time.perf_counter(); d.close()
    print("create %.3f  first decode %.3f  second decode %.3f ms  iters %s" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, d.iter), flush=True)

import importlib, json, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp")
rows = json.load(open("tests/golden/hqc_first_rows.json"))
H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])
N, omega = S.codes.HQC_PARAMS["hqc128"]; eps=0.05
probs = np.concatenate([np.full(N, omega / N), np.full(H.m, eps)])
dec = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
r = dec.mc_hqc_run(32768, omega=omega, eps=eps, seed=1)
print("compacted", dec.last_compacted())
h = np.bincount(r["iters"], minlength=101)
print({i:int(c) for i,c in enumerate(h) if c})
print("success", r["success"].mean())

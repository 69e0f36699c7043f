#!/usr/bin/env python3
"""300 CONVERGING single decode() calls (the attack loop's call once enough checks are in: 3-4 iterations,
tanh rule, early exit) on the HQC-128 bench graph, host buffers in and out as `decode()` takes them.
Prints the wall time per call; run under `rocprofv3 --hip-trace --kernel-trace --stats` for the API / kernel split."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp")
trials = importlib.import_module("sca-ldpc_amd.trials")
rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])
N, omega = S.codes.HQC_PARAMS["hqc128"]
probs = trials.hqc_priors(N, H.m, omega, 0.05)
msg, ys = trials.hqc_trials(Hin, omega, 0.05, 8, base_seed=2, first_index=0)
dec = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
for i in range(8):
    dec.decode(msg[i])
n = 300
t0 = time.perf_counter()
its = []
for i in range(n):
    dec.decode(msg[i % 8])
    its.append(dec.iter)
dt = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for i in range(n):
    dec.decode_batch(msg[i % 8][None, :], early_exit=True)
dt2 = (time.perf_counter() - t0) / n
print(json.dumps({"ms_per_decode_with_llr": dt * 1e3, "ms_per_decode_batch1_no_llr": dt2 * 1e3, "iterations": [min(its), max(its)],
                  "row_parallel": dec.last_stats()["row_parallel"]}))
dec.close()

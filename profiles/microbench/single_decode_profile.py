#!/usr/bin/env python3
"""20 non-converging single decode() calls (100 iterations each, tanh rule, early exit) on the
HQC-128 bench graph: run under `rocprofv3 --kernel-trace --stats` to see the row-parallel kernels."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp")
rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])
N, omega = S.codes.HQC_PARAMS["hqc128"]
probs = np.concatenate([np.full(N, omega / N), np.full(H.m, 0.05)])
x = np.concatenate([np.zeros(N, np.uint8), np.random.RandomState(0).randint(0, 2, H.m).astype(np.uint8)])
dec = bp.bp_decoder(H, max_iter=100, bp_method=sys.argv[1] if len(sys.argv) > 1 else "product_sum", channel_probs=probs)
for _ in range(20):
    dec.decode(x)
print(dec.iter, dec.converge, dec.last_stats())
dec.close()

#!/usr/bin/env python3
"""Latency of small decode calls on the HQC-128 bench graph (E = 204000, n = 21669): the
64-codeword-tile kernels (SCALDPC_PATH=stream) against the row-parallel kernels
(SCALDPC_PATH=edge, wave = row, lane = edge).  Fixed iteration counts isolate the cost per
iteration; the early-exit rows are what the attack loop's single `decode()` (hqc.py:708) sees.

    python profiles/microbench/small_batch_latency.py            (on the GPU box)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib

S = importlib.import_module("sca-ldpc_amd")
bp = importlib.import_module("sca-ldpc_amd.bp")


def main():
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
    H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])
    N, omega = S.codes.HQC_PARAMS["hqc128"]
    eps = 0.05
    probs = np.concatenate([np.full(N, omega / N), np.full(H.m, eps)])
    out = []
    # what one hqc.decode() of the attack loop pays: a NEW decoder per call (hqc.py:694) + one decode
    os.environ.pop("SCALDPC_PATH", None)
    warm = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
    x1 = warm.mc_hqc_run(1, omega=omega, eps=eps, seed=3, want_inputs=True)["msg"][0]
    warm.decode(x1)
    for rep in range(3):
        t0 = time.perf_counter()
        d = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
        t1 = time.perf_counter()
        d.decode(x1)
        t2 = time.perf_counter()
        d.decode(x1)
        t3 = time.perf_counter()
        d.close()
        t4 = time.perf_counter()
        print(json.dumps({"new_decoder_ms": round((t1 - t0) * 1e3, 3), "first_decode_ms": round((t2 - t1) * 1e3, 3),
                          "second_decode_ms": round((t3 - t2) * 1e3, 3), "close_ms": round((t4 - t3) * 1e3, 3)}), flush=True)
    # the attack loop's usual case while checks are still scarce: a decode that never converges
    # runs all max_iter = 100 iterations WITH the per-iteration convergence test
    rng = np.random.RandomState(0)
    xbad = np.concatenate([np.zeros(N, np.uint8), rng.randint(0, 2, H.m).astype(np.uint8)])
    # (knobs are per handle since round 2: the environment is only read when a decoder is created)
    for label, kn in (("tiles", dict(path="stream", el_fuse=1)), ("row-parallel, 4 launches/iteration", dict(path="auto", el_fuse=0)),
                      ("row-parallel, 2 launches/iteration", dict(path="auto", el_fuse=1))):
        warm.configure(**kn)
        warm.decode(xbad)
        t0 = time.perf_counter()
        for _ in range(5):
            warm.decode(xbad)
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"non_converging_decode_100_iterations": label, "ms": round(dt * 1e3, 3), "iter": int(warm.iter),
                          "converge": int(warm.converge)}), flush=True)
    warm.close()
    # ... and the whole attack-loop step through the build's driver: 4000 accumulated checks,
    # sparse graph assembly in Python + new decoder + decode + statistics (hqc.py:661-759)
    D = importlib.import_module("sca-ldpc_amd.driver")
    acc = D.HqcCheckAccumulator(N, rows["N17669_W50_s0"], omega)
    rs = np.random.RandomState(5)
    y = np.sort(rs.choice(N, omega, replace=False))
    yv = np.zeros(N, np.uint8); yv[y] = 1
    for b in rs.permutation(N)[:4000]:
        sup = (int(b) - acc.k) % N
        acc.add_check(int(b), int(yv[sup].sum() & 1), 0.95)
    acc.decode(list(y))
    for rep in range(3):
        t0 = time.perf_counter()
        g = acc.graph()
        t1 = time.perf_counter()
        ok = acc.decode(list(y))
        t2 = time.perf_counter()
        print(json.dumps({"accumulator_graph_ms": round((t1 - t0) * 1e3, 3), "accumulator_decode_total_ms": round((t2 - t1) * 1e3, 3),
                          "success": bool(ok)}), flush=True)
    for method in ("min_sum", "product_sum"):
        dec = bp.bp_decoder(H, max_iter=100, bp_method=method, channel_probs=probs)
        trials = dec.mc_hqc_run(64, omega=omega, eps=eps, seed=3, want_inputs=True)
        msg = trials["msg"]
        for nb in (1, 2, 4, 8, 16, 32, 64):
            x = np.ascontiguousarray(msg[:nb])
            row = {"method": method, "codewords": nb}
            for path in ("stream", "edge"):
                dec.configure(path=path)
                for label, kw, iters in (("fixed100", dict(early_exit=False), 100), ("early", dict(early_exit=True), None)):
                    dec.decode_batch(x, **kw)  # warm-up (allocations)
                    reps = 5
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        r = dec.decode_batch(x, **kw)
                    dt = (time.perf_counter() - t0) / reps
                    row[f"{path}_{label}_ms"] = round(dt * 1e3, 3)
                    if iters:
                        row[f"{path}_us_per_iter"] = round(dt * 1e6 / iters, 2)
                    else:
                        row["mean_iters"] = float(np.mean(r["iters"]))
            out.append(row)
            print(json.dumps(row), flush=True)
        dec.close()


if __name__ == "__main__":
    main()

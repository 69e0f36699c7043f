#!/usr/bin/env python3
"""One attack-loop step at ~4000 accumulated checks on the HQC-128 graph (N = 17669, W = 50): the
reference rebuilds the decoder on every decode (hqc.py:680,694); round 1 did the same through the C
ABI (create + decode + destroy, 0.8 ms per step through the driver).  Round 2 keeps ONE decoder and
appends the new rows (scaldpc_bp_append_rows).  Times per step, same inputs, same outputs:
    rebuild : bp_decoder(graph(R)) + decode_batch(1 codeword, early exit) + close()
    append  : append_rows(last `step` rows) + decode_batch(1 codeword, early exit)
and the decode alone on a warm decoder, for reference."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd")
bp = importlib.import_module("sca-ldpc_amd.bp")


def main():
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
    N, omega, eps = 17669, 66, 0.05
    Rmax = 6000
    _, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"], R=Rmax)
    rng = np.random.RandomState(1)
    y = np.zeros((1, N), dtype=np.uint8)
    y[0, rng.choice(N, omega, replace=False)] = 1
    checks = Hin.syndrome(y) ^ (rng.rand(1, Rmax) < eps).astype(np.uint8)
    W1 = Hin.col_idx.size // Rmax + 1
    cols = np.concatenate([Hin.col_idx.reshape(Rmax, -1), N + np.arange(Rmax, dtype=np.int32)[:, None]], axis=1)

    def graph(r):
        return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * W1, cols[:r].reshape(-1))

    def probs(r):
        return np.concatenate([np.full(N, omega / N), np.full(r, eps)])

    def msg(r):
        return np.concatenate([np.zeros((1, N), dtype=np.uint8), checks[:, :r]], axis=1)

    for step in (50, 100):
        R0 = 4000
        sizes = list(range(R0, Rmax + 1, step))  # 20 / 40 steps: medians below
        # rebuild per decode
        graphs = {r: graph(r) for r in sizes}
        pr = {r: probs(r) for r in sizes}
        ms = {r: msg(r) for r in sizes}
        outs_a = []
        bp.bp_decoder(graphs[R0], max_iter=100, bp_method="product_sum", channel_probs=pr[R0]).close()  # warm the block cache
        tr = []
        for r in sizes[1:]:
            t0 = time.perf_counter()
            d = bp.bp_decoder(graphs[r], max_iter=100, bp_method="product_sum", channel_probs=pr[r])
            outs_a.append(d.decode_batch(ms[r], early_exit=True))
            d.close()
            tr.append(time.perf_counter() - t0)
        t_rebuild = float(np.median(tr))
        # append
        live = bp.bp_decoder(graphs[R0], max_iter=100, bp_method="product_sum", channel_probs=pr[R0])
        live.decode_batch(ms[R0], early_exit=True)
        tails = {r: (np.arange(step + 1, dtype=np.int32) * W1, cols[r - step : r].reshape(-1).copy(), np.full(step, eps)) for r in sizes[1:]}
        live.append_rows(*tails[sizes[1]][:2], N + sizes[1], tails[sizes[1]][2])  # first append: CSR moves to growable buffers
        live.decode_batch(ms[sizes[1]], early_exit=True)
        outs_b = [None]
        ts, tapp = [], []
        for r in sizes[2:]:
            ta = time.perf_counter()
            live.append_rows(*tails[r][:2], N + r, tails[r][2])
            tb = time.perf_counter()
            outs_b.append(live.decode_batch(ms[r], early_exit=True))
            ts.append(time.perf_counter() - ta)
            tapp.append(tb - ta)
        t_append, t_app = float(np.median(ts)), float(np.median(tapp))
        t0 = time.perf_counter()
        for _ in range(20):
            live.decode_batch(ms[sizes[-1]], early_exit=True)
        t_dec = (time.perf_counter() - t0) / 20
        same = all(np.array_equal(a["bits"], b["bits"]) and np.array_equal(a["iters"], b["iters"]) for a, b in zip(outs_a[1:], outs_b[1:]))
        it = [int(a["iters"][0]) for a in outs_a]
        print(json.dumps({"rows_per_step": step, "checks": [sizes[0], sizes[-1]], "steps_timed": len(ts), "ms_per_step_rebuild": t_rebuild * 1e3,
                          "ms_per_step_append": t_append * 1e3, "ms_append_call_alone": t_app * 1e3,
                          "ms_decode_alone_warm": t_dec * 1e3, "iterations_per_decode": [min(it), max(it)], "outputs_identical": same}))
        live.close()


if __name__ == "__main__":
    main()

"""K6 on the device vs its oracle: generated trials bit-exact, success flags equal to decoding
the exported inputs through decode_batch, independence from batch split / first_trial."""
import importlib

import numpy as np
import pytest

from helpers import S, hqc_instance

pytestmark = pytest.mark.gpu
bp = importlib.import_module("sca-ldpc_amd.bp")
trials = importlib.import_module("sca-ldpc_amd.trials")


@pytest.mark.parametrize("method", ["min_sum", "product_sum"])
def test_fer_run(oracle, golden, method):
    rng = S.codes.make_random_state(0)
    g = S.codes.make_regular_ldpc_graph(300, 150, 3, 6, rng)
    probs = np.array([golden["distr_files"]["binary_distr"][i % 4][0] for i in range(g.n)]) * 0.3
    dec = bp.bp_decoder(g, max_iter=30, bp_method=method, channel_probs=probs)
    r = dec.mc_fer_run(300, seed=11, first_trial=5, want_errors=True)
    err = oracle.mc_bernoulli(11, 5, 300, g.n, probs)
    assert np.array_equal(r["errors"], err)
    d = dec.decode_batch(g.syndrome(err), early_exit=True)
    assert np.array_equal(r["success"], (d["bits"] == err).all(axis=1).astype(np.uint8))
    assert np.array_equal(r["iters"], d["iters"])
    assert 0 < r["success"].sum() < 300
    # global-index determinism: two half-batches == one batch
    a = dec.mc_fer_run(100, seed=11, first_trial=5)
    b = dec.mc_fer_run(200, seed=11, first_trial=105)
    assert np.array_equal(np.concatenate([a["success"], b["success"]]), r["success"])


@pytest.mark.parametrize("eps", [0.0, 0.04])
def test_hqc_run(oracle, eps):
    H, Hin, probs, _, _ = hqc_instance(499, 7, 260, 5, eps, 1, seed=11, flip=False)
    with np.errstate(divide="ignore"):
        dec = bp.bp_decoder(H, max_iter=40, bp_method="product_sum", channel_probs=probs)
    runs = 200
    r = dec.mc_hqc_run(runs, omega=5, eps=eps, seed=42, first_trial=1000, want_inputs=True)
    y = oracle.mc_hqc_secret(42, 1000, runs, 499, 5)
    assert np.array_equal(r["y"], y)
    yv = np.zeros((runs, 499), dtype=np.uint8)
    np.put_along_axis(yv, y.astype(np.int64), 1, axis=1)
    checks = Hin.syndrome(yv) ^ oracle.mc_bernoulli(42, 1000, runs, Hin.m, None, eps)
    msg = np.concatenate([np.zeros((runs, 499), dtype=np.uint8), checks], axis=1)
    assert np.array_equal(r["msg"], msg)
    d = dec.decode_batch(msg, early_exit=True)
    assert np.array_equal(r["success"], trials.success(d["bits"], y, 499).astype(np.uint8))
    assert np.array_equal(r["iters"], d["iters"])
    assert r["success"].mean() > 0.3


def test_config5_sweep_sheds_stragglers_twice(oracle):
    """BASELINE config 5 (HQC-128 graph, eps = 0.05, tanh rule, early exit, max_iter 100): the
    stragglers of the first pass (codewords needing 5+ iterations) are re-decoded in dense tiles,
    and that pass sheds the few that never converge once more, so they run their 100 iterations in
    a fraction of the tiles.  Held to the ORACLE on the sweep's own trials (the sweep used to be
    compared with itself only): the inputs of all 8192 trials are exported (`want_inputs`), decoded by the f32 oracle
    with early exit and max_iter 100, and iteration counts and success flags must agree -- including the trials that
    went through compaction level 1 (5 .. 99 iterations) and level 2 (the never-converging ones: 100 iterations, not
    converged).  A posterior that is a tie of `L <= 0 -> 1` may move ONE iteration count by one between the device's
    transcendentals and glibc's (helpers.compare), so at most 8192 / 2000 trials may differ in their count; the
    success flags of all others must be equal.  Then the sweep against itself with the scheduling knobs moved."""
    import json, os

    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hqc_first_rows.json")))
    H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])
    N, omega = S.codes.HQC_PARAMS["hqc128"]
    probs = np.concatenate([np.full(N, omega / N), np.full(H.m, 0.05)])
    dec = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
    runs = 8192
    a = dec.mc_hqc_run(runs, omega=omega, eps=0.05, seed=7, want_inputs=True)
    st = dec.last_stats()
    assert st["compacted"] > 0 and st["levels"] >= 2, st
    # -- the oracle on the same trials ---------------------------------------------------------------------------
    late = (a["iters"] >= 5) & (a["iters"] < 100)  # went through compaction level 1 (handed over at iteration 4)
    never = (a["iters"] == 100) & (a["success"] == 0)
    assert late.sum() >= 20 and never.sum() >= 5, (int(late.sum()), int(never.sum()))
    threads = max(1, min(os.cpu_count() or 1, oracle.max_threads()))
    ref = oracle.bp_decode_batch(H, probs, a["msg"], 1, 100, "tanh_complement", dtype="f32", threads=threads, early_exit=True)
    same = ref["iters"] == a["iters"]
    assert (~same).sum() <= runs // 2000, f"{int((~same).sum())} iteration counts differ from the oracle's"
    assert same[late].mean() > 0.9 and same[never].all()  # (the straggler levels are what this test is about)
    ref_success = trials.success(ref["bits"], a["y"], N).astype(np.uint8)
    assert np.array_equal(ref_success[same], a["success"][same]), "success flags differ from the oracle's"
    assert ref["converged"][same & (a["iters"] < 100)].all() and not ref["converged"][never].any()
    # -- and against itself ---------------------------------------------------------------------------------------
    dec.configure(fuse_test=0)  # a stand-alone convergence test after every variable pass
    c = dec.mc_hqc_run(runs, omega=omega, eps=0.05, seed=7)
    assert np.array_equal(a["success"], c["success"]) and np.array_equal(a["iters"], c["iters"])
    dec.configure(fuse_test=1, first_fused=0)  # iteration 1 with its check pass
    e = dec.mc_hqc_run(runs, omega=omega, eps=0.05, seed=7)
    assert np.array_equal(a["success"], e["success"]) and np.array_equal(a["iters"], e["iters"])
    dec.configure(first_fused=1, compact_after=0)
    b = dec.mc_hqc_run(runs, omega=omega, eps=0.05, seed=7)
    assert dec.last_stats()["levels"] == 0
    dec.close()
    assert np.array_equal(a["success"], b["success"]) and np.array_equal(a["iters"], b["iters"])
    assert (a["iters"] == 100).sum() > 0 and 0.7 < a["success"].mean() < 0.9


def test_hqc_run_needs_identity_block():
    g = S.codes.rep_code_graph(9)
    dec = bp.bp_decoder(g, error_rate=0.1)
    with pytest.raises(ValueError, match="Hin"):
        dec.mc_hqc_run(4, omega=2, eps=0.1, seed=1)


def test_mc_device_outputs():
    """The Monte-Carlo entry points with device-side outputs (SCALDPC_F_DEVICE_IO)."""
    import ctypes as C

    import torch

    lib = importlib.import_module("sca-ldpc_amd._lib")
    H, Hin, probs, _, _ = hqc_instance(499, 7, 260, 5, 0.04, 1, seed=11, flip=False)
    dec = bp.bp_decoder(H, max_iter=40, bp_method="min_sum", channel_probs=probs)
    host = dec.mc_hqc_run(200, omega=5, eps=0.04, seed=9)
    d_s = torch.zeros(200, dtype=torch.uint8, device="cuda")
    d_i = torch.zeros(200, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # stream = NULL below means the handle's own stream: order it after torch's fills
    lib.check(dec._lib.scaldpc_mc_hqc_run(dec._h, 5, 0.04, 0, 200, 9, 40, lib.BP_MIN_SUM, 1.0,
                                          lib.F_EARLY_EXIT | lib.F_DEVICE_IO, None, C.c_void_p(d_s.data_ptr()),
                                          C.c_void_p(d_i.data_ptr()), None, None))
    assert np.array_equal(d_s.cpu().numpy(), host["success"]) and np.array_equal(d_i.cpu().numpy(), host["iters"])
    dec.close()

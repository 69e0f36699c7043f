"""the attack loop's growing parity-check matrix on a LIVE decoder.

The reference adds one row per oracle answer (`H = np.vstack([H, row])`, simulate/hqc.py:885-908) and
rebuilds `bp_decoder([H | I])` from the dense matrix on every decode (hqc.py:680,694,972-980).  Here rows
are appended to the decoder that already exists (`scaldpc_bp_append_rows`): device CSR / priors with spare
capacity, the row-parallel kernels' tables updated in place.  The claim under test: after ANY sequence
of appends, every output (decisions, posteriors, iteration counts, flags) equals that of a decoder
freshly built on the grown graph -- bit for bit, on every kernel family."""
import importlib
import json
import os

import numpy as np
import pytest

from helpers import ORACLE_METHOD, S, check_reference_form, compare, hqc_instance, reference_floor

pytestmark = pytest.mark.gpu
bp = importlib.import_module("sca-ldpc_amd.bp")
driver = importlib.import_module("sca-ldpc_amd.driver")


def _same(a, b, what):
    for k in ("bits", "llr", "iters", "converged"):
        assert np.array_equal(a[k], b[k], equal_nan=True) if k == "llr" else np.array_equal(a[k], b[k]), (what, k)


def _rows_csr(H, r0, r1):
    rp = H.row_ptr[r0 : r1 + 1].astype(np.int64)
    return (rp - rp[0]).astype(np.int32), H.col_idx[rp[0] : rp[-1]]


def test_hqc128_graph_grown_50_rows_at_a_time_equals_fresh_decoders():
    """The attack loop at BASELINE size: N = 17669, W = 50, checks arriving 50 at a time up to 4000.  At
    every step the live decoder's single decode() (row-parallel kernels, tanh rule, early exit,
    max_iter 100 as hqc.py:696) equals a freshly built decoder's; every 16th step a 70-codeword batch
    (tile kernels: the stale CSC / buckets / tile tables are rebuilt on demand) does too."""
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hqc_first_rows.json")))
    Hfull, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"])  # 4000 x (17669 + 4000)
    N, omega, eps, R = 17669, 66, 0.05, 4000
    rng = np.random.RandomState(77)
    y = np.zeros((70, N), dtype=np.uint8)
    for b in range(70):
        y[b, rng.choice(N, omega, replace=False)] = 1
    checks = Hin.syndrome(y) ^ (rng.rand(70, R) < eps).astype(np.uint8)
    cert = np.where(rng.rand(R) < 0.1, 1.0, 1.0 - eps)  # some certainty-1.0 checks: p = 0 priors (hqc.py:689)

    def graph(r):  # [Hin[:r] | I_r]
        rp = Hin.row_ptr[: r + 1]
        cols = np.concatenate([Hin.col_idx[: rp[-1]].reshape(r, -1), N + np.arange(r, dtype=np.int32)[:, None]], axis=1)
        return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * cols.shape[1], cols.reshape(-1))

    def probs(r):
        return np.concatenate([np.full(N, omega / N), 1.0 - cert[:r]])

    def msg(r, nb):
        return np.concatenate([np.zeros((nb, N), dtype=np.uint8), checks[:nb, :r]], axis=1)

    step = 50
    with np.errstate(divide="ignore"):
        live = bp.bp_decoder(graph(step), max_iter=100, bp_method="product_sum", channel_probs=probs(step))
        for i, r in enumerate(range(step, R + 1, step)):
            if r > step:
                g = graph(r)
                rp, ci = _rows_csr(g, r - step, r)
                live.append_rows(rp, ci, N + r, 1.0 - cert[r - step : r])
            fresh = bp.bp_decoder(graph(r), max_iter=100, bp_method="product_sum", channel_probs=probs(r))
            a = live.decode_batch(msg(r, 1), early_exit=True, want_llr=True)
            b = fresh.decode_batch(msg(r, 1), early_exit=True, want_llr=True)
            assert live.last_row_parallel() == (1 if r >= 100 else 0)  # (the first 50 rows still fit the LDS-resident decoder)
            _same(a, b, f"single decode at {r} checks")
            if i % 16 == 15 or r == R:
                a = live.decode_batch(msg(r, 70), early_exit=True, want_llr=True)
                b = fresh.decode_batch(msg(r, 70), early_exit=True, want_llr=True)
                assert live.last_row_parallel() == 0
                _same(a, b, f"70-codeword batch at {r} checks")
            fresh.close()
        # the end state is the BASELINE config-2 graph: its Monte-Carlo entry point works on the grown handle
        assert (live.m, live.n) == (R, N + R)
        r1 = live.mc_hqc_run(256, omega=omega, eps=eps, seed=3)
        fresh = bp.bp_decoder(graph(R), max_iter=100, bp_method="product_sum", channel_probs=probs(R))
        r2 = fresh.mc_hqc_run(256, omega=omega, eps=eps, seed=3)
        assert np.array_equal(r1["success"], r2["success"]) and np.array_equal(r1["iters"], r2["iters"])
        fresh.close()
        live.close()


@pytest.mark.parametrize("method", ["min_sum", "product_sum"])
@pytest.mark.parametrize("path", ["auto", "stream", "edge"])
def test_ragged_appends_on_every_kernel_family(oracle, method, path, monkeypatch):
    """Small HQC-shaped graph grown in ragged steps (1, 2, 7, 40, ... rows), enough of them that columns
    outgrow their segments and move; decoded after every append through the LDS-resident decoder (while
    the graph still fits), the tile kernels and the row-parallel kernels; batch sizes on both sides of
    the row-parallel limit.  Equal to a fresh decoder at every step, and to the oracle at the end."""
    if path == "auto":
        monkeypatch.delenv("SCALDPC_PATH", raising=False)
    else:
        monkeypatch.setenv("SCALDPC_PATH", path)
    H, Hin, probs, msg, y = hqc_instance(997, 9, 600, 6, 0.03, 90, seed=41)
    N = 997

    def graph(r):
        rp = Hin.row_ptr[: r + 1]
        cols = np.concatenate([Hin.col_idx[: rp[-1]].reshape(r, -1), N + np.arange(r, dtype=np.int32)[:, None]], axis=1)
        return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * cols.shape[1], cols.reshape(-1))

    sizes = [5, 6, 8, 15, 55, 56, 130, 131, 260, 400, 600]
    live = bp.bp_decoder(graph(sizes[0]), max_iter=25, bp_method=method, channel_probs=np.concatenate([probs[:N], probs[N : N + sizes[0]]]))
    for prev, r in zip([None] + sizes[:-1], sizes):
        if prev is not None:
            rp, ci = _rows_csr(graph(r), prev, r)
            live.append_rows(rp, ci, N + r, probs[N + prev : N + r])
        pr = np.concatenate([probs[:N], probs[N : N + r]])
        fresh = bp.bp_decoder(graph(r), max_iter=25, bp_method=method, channel_probs=pr)
        for nb in (1, 3, 70):
            x = np.concatenate([msg[:nb, :N], msg[:nb, N : N + r]], axis=1)
            for early in (True, False):
                a = live.decode_batch(x, early_exit=early, want_llr=True)
                b = fresh.decode_batch(x, early_exit=early, want_llr=True)
                _same(a, b, (r, nb, early))
        fresh.close()
    ref = oracle.bp_decode_batch(H, probs, msg, 1, 25, ORACLE_METHOD[method], dtype="f32", threads=8)
    final = live.decode_batch(msg, early_exit=True, want_llr=True)
    compare(final, ref, method)
    if method == "product_sum":
        check_reference_form(oracle, final, H, probs, msg, 1, 25, True, min_fraction=reference_floor(ref))
    live.close()


def test_general_growth_new_columns_old_columns_isolated_columns(oracle):
    """Not only [Hin | I]: appended rows may touch any old column, bring several new columns, bring none,
    or leave a new column untouched (an isolated variable still has a posterior = its prior); an append
    of zero rows only adds columns.  Fresh-decoder equality after every step."""
    rng = np.random.RandomState(9)
    m0, n0 = 60, 400
    dense = (rng.rand(m0, n0) < 0.02).astype(np.int8)
    dense[np.arange(m0), rng.randint(0, n0, m0)] = 1
    live = None
    probs = rng.uniform(0.02, 0.2, size=n0)
    Hd = dense
    steps = [(30, 0), (0, 5), (10, 7), (25, 1), (1, 0), (40, 40)]  # (rows added, columns added)
    live = bp.bp_decoder(S.TannerGraph.from_dense(Hd), max_iter=20, bp_method="product_sum", channel_probs=probs)
    live.configure(path="stream")
    for add_r, add_c in steps:
        n_new = Hd.shape[1] + add_c
        new_rows = (rng.rand(add_r, n_new) < 0.02).astype(np.int8)
        if add_r:
            new_rows[np.arange(add_r), rng.randint(0, n_new, add_r)] = 1
        if add_c > 2:
            new_rows[:, n_new - 1] = 0  # the last new column stays isolated
        g_new = S.TannerGraph.from_dense(new_rows) if add_r else None
        tail = rng.uniform(0.02, 0.2, size=add_c)
        live.append_rows(g_new.row_ptr if add_r else np.zeros(1, np.int32), g_new.col_idx if add_r else np.zeros(0, np.int32),
                         n_new, tail)
        Hd = np.concatenate([np.concatenate([Hd, np.zeros((Hd.shape[0], add_c), np.int8)], axis=1), new_rows], axis=0)
        probs = np.concatenate([probs, tail])
        G = S.TannerGraph.from_dense(Hd)
        err = (rng.rand(80, G.n) < 0.05).astype(np.uint8)
        synd = G.syndrome(err)
        for pth, nb in (("stream", 80), ("edge", 5), ("edge", 64)):
            live.configure(path=pth)
            fresh = bp.bp_decoder(G, max_iter=20, bp_method="product_sum", channel_probs=probs)
            fresh.configure(path=pth)
            a = live.decode_batch(synd[:nb], early_exit=True, want_llr=True)
            b = fresh.decode_batch(synd[:nb], early_exit=True, want_llr=True)
            _same(a, b, (Hd.shape, pth, nb))
            fresh.close()
    ref = oracle.bp_decode_batch(G, probs, synd, 0, 20, "tanh_complement", dtype="f32", threads=8)
    live.configure(path="stream")
    final = live.decode_batch(synd, early_exit=True, want_llr=True)
    compare(final, ref, "product_sum")
    check_reference_form(oracle, final, G, probs, synd, 0, 20, True, min_fraction=reference_floor(ref))
    live.close()


def test_append_errors_and_unset_priors():
    g = S.codes.rep_code_graph(9)
    dec = bp.bp_decoder(g, error_rate=0.1, max_iter=9)
    with pytest.raises(ValueError):  # column out of range
        dec.append_rows(np.array([0, 1], np.int32), np.array([12], np.int32), 10, np.array([0.1]))
    with pytest.raises(ValueError):  # not ascending
        dec.append_rows(np.array([0, 2], np.int32), np.array([3, 1], np.int32), 9, np.zeros(0))
    with pytest.raises(ValueError):  # tail length
        dec.append_rows(np.array([0, 1], np.int32), np.array([2], np.int32), 11, np.array([0.1]))
    lib = importlib.import_module("sca-ldpc_amd._lib")
    rp, ci = np.array([0, 2], np.int32), np.array([0, 9], np.int32)
    lib.check(dec._lib.scaldpc_bp_append_rows(dec._h, 1, lib.ptr(rp), lib.ptr(ci), 10))
    dec.m, dec.n = 9, 10
    with pytest.raises(ValueError, match="not set"):  # the new column has no prior yet
        dec.decode_batch(np.zeros((1, 9), np.uint8), input_vector_type="syndrome")
    dec.close()


def test_incremental_accumulator_keeps_one_decoder_alive(golden):
    """driver.HqcCheckAccumulator (the attack loop's add_checks / decode, hqc.py:953-984) on the HIP
    decoder: one decoder object for the whole run, rows appended between decodes; the stats rows equal
    those of the rebuild-per-decode pattern (`hqc_decode`)."""
    t = golden["hqc_decode_tests"]["full"]
    N, ysp, r1 = t["N"], t["y_sparse"], t["first_row"]
    from test_oracle_pins import sparse_times_sparse

    yr = set(sparse_times_sparse(ysp, r1, N))
    rng = np.random.RandomState(5)
    bits = rng.permutation(N)[:1500]
    acc = driver.HqcCheckAccumulator(N, r1, len(ysp), bp_decoder=bp.bp_decoder, max_iter=100, decode_every=250)
    seen = []
    ok = False
    for i in range(0, 1500, 50):
        chunk = [(int(b), 1.0) for b in bits[i : i + 50] if b in yr] + [(int(b), 0.97) for b in bits[i : i + 50] if b not in yr]
        for b, c in chunk:
            acc.add_check(b, 1 if b in yr else 0, c)
        if len(acc) // 250 > len(seen):
            first = acc._bpd
            ok = acc.decode(ysp)
            assert first is None or acc._bpd is first  # the same decoder object, grown
            seen.append(len(acc))
            checks = acc.checks
            R = len(acc)
            Hin = S.TannerGraph.from_csr(R, N, np.arange(R + 1, dtype=np.int64) * len(r1), acc._cols[:R, :-1].reshape(-1))
            ref_ok, ref_stats = driver.hqc_decode(N, Hin, checks, ysp, bp_decoder=bp.bp_decoder, max_iter=100)
            got = acc.decoder_stats[-1]
            assert ok == ref_ok and all(got[k] == v for k, v in ref_stats.items() if k in got), (len(acc), got, ref_stats)
    assert len(seen) >= 3
    acc.close()


try:
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=int(os.environ.get("SCALDPC_PROPERTY_EXAMPLES", "25")), deadline=None, derandomize=True,
              suppress_health_check=list(HealthCheck))
    @given(N=st.integers(300, 2200), W=st.integers(4, 12), r0=st.integers(1, 300), steps=st.lists(st.integers(1, 120), min_size=1, max_size=6),
           omega=st.integers(2, 9), eps=st.sampled_from([0.0, 0.03]), method=st.sampled_from(["min_sum", "product_sum"]),
           path=st.sampled_from(["auto", "stream", "edge"]), nb=st.sampled_from([1, 2, 5, 9, 70]), early=st.booleans(),
           decode_between=st.booleans(), seed=st.integers(0, 9999))
    def test_random_append_sequences_equal_fresh_decoders(N, W, r0, steps, omega, eps, method, path, nb, early, decode_between, seed):
        """Random HQC-shaped graphs grown by random step sizes, decoded (or not) between the appends, on a random
        kernel family and batch size: after the last append -- and after every one when `decode_between` -- the
        live decoder equals a fresh one bit for bit.  Not decoding between appends exercises several appends
        against stale full tables and tables that do not exist yet."""
        R = r0 + sum(steps)
        H, Hin, probs, msg, y = hqc_instance(N, W, min(R, N), omega, eps, nb, seed=seed)
        R = Hin.m
        sizes = [min(r0, R)]
        for s_ in steps:
            if sizes[-1] < R:
                sizes.append(min(R, sizes[-1] + s_))

        def graph(r):
            rp = Hin.row_ptr[: r + 1]
            cols = np.concatenate([Hin.col_idx[: rp[-1]].reshape(r, -1), N + np.arange(r, dtype=np.int32)[:, None]], axis=1)
            return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * cols.shape[1], cols.reshape(-1))

        def pr(r):
            return np.concatenate([probs[:N], probs[N : N + r]])

        def x(r):
            return np.concatenate([msg[:, :N], msg[:, N : N + r]], axis=1)

        with np.errstate(divide="ignore"):
            live = bp.bp_decoder(graph(sizes[0]), max_iter=20, bp_method=method, channel_probs=pr(sizes[0]))
            live.configure(path=path)
            if decode_between:
                live.decode_batch(x(sizes[0]), early_exit=early)
            for prev, r in zip(sizes[:-1], sizes[1:]):
                rp, ci = _rows_csr(graph(r), prev, r)
                live.append_rows(rp, ci, N + r, probs[N + prev : N + r])
                if decode_between or r == sizes[-1]:
                    fresh = bp.bp_decoder(graph(r), max_iter=20, bp_method=method, channel_probs=pr(r))
                    fresh.configure(path=path)
                    a = live.decode_batch(x(r), early_exit=early, want_llr=True)
                    b = fresh.decode_batch(x(r), early_exit=early, want_llr=True)
                    fresh.close()
                    _same(a, b, (N, W, sizes, r, method, path, nb, early))
            live.close()
except ImportError:  # hypothesis is optional
    pass


def test_append_to_an_empty_graph_and_while_asynchronous_work_is_in_flight(oracle):
    """Two corners: (i) a decoder built on a graph WITHOUT edges (one empty check) grows into a real one;
    (ii) rows are appended right after SCALDPC_F_ASYNC decodes were enqueued on the caller's stream, with no
    synchronisation in between -- the append must wait for them (their tables and buffers are being
    replaced), and both the in-flight results and the later ones must be right."""
    import torch

    lib = importlib.import_module("sca-ldpc_amd._lib")
    H, Hin, probs, msg, y = hqc_instance(997, 9, 400, 6, 0.03, 130, seed=51)
    N = 997

    def graph(r):
        rp = Hin.row_ptr[: r + 1]
        cols = np.concatenate([Hin.col_idx[: rp[-1]].reshape(r, -1), N + np.arange(r, dtype=np.int32)[:, None]], axis=1)
        return S.TannerGraph.from_csr(r, N + r, np.arange(r + 1, dtype=np.int64) * cols.shape[1], cols.reshape(-1))

    # (i) empty start: one check with no variables over the N secret columns
    empty = S.TannerGraph.from_csr(1, N, np.zeros(2, dtype=np.int64), np.zeros(0, dtype=np.int32))
    live = bp.bp_decoder(empty, max_iter=20, bp_method="product_sum", channel_probs=probs[:N])
    z = live.decode_batch(np.zeros((2, 1), np.uint8), input_vector_type="syndrome")
    assert not z["bits"].any()
    g = graph(200)
    live.append_rows(g.row_ptr, g.col_idx, N + 200, probs[N : N + 200])
    full = S.TannerGraph.from_csr(201, N + 200, np.concatenate([[0], g.row_ptr]), g.col_idx)  # the empty check stays row 0
    fresh = bp.bp_decoder(full, max_iter=20, bp_method="product_sum", channel_probs=np.concatenate([probs[:N], probs[N : N + 200]]))
    synd = np.concatenate([np.zeros((5, 1), np.uint8), msg[:5, N : N + 200]], axis=1)
    for nb in (1, 5):
        _same(live.decode_batch(synd[:nb], want_llr=True, input_vector_type="syndrome"),
              fresh.decode_batch(synd[:nb], want_llr=True, input_vector_type="syndrome"), ("from empty", nb))
    fresh.close()
    live.close()

    # (ii) append with asynchronous decodes still in flight
    pr = lambda r: np.concatenate([probs[:N], probs[N : N + r]])
    x = lambda r: np.ascontiguousarray(np.concatenate([msg[:, :N], msg[:, N : N + r]], axis=1))
    live = bp.bp_decoder(graph(300), max_iter=30, bp_method="min_sum", channel_probs=pr(300))
    live.configure(path="stream")
    ref300 = oracle.bp_decode_batch(graph(300), pr(300), x(300), 1, 30, "min_sum", dtype="f32", threads=8, early_exit=False)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        d_in = torch.from_numpy(x(300)).cuda()
        outs = [torch.zeros((130, N + 300), dtype=torch.uint8, device="cuda") for _ in range(3)]
        for o in outs:
            live.decode_batch_device(d_in.data_ptr(), lib.IN_RECEIVED, 130, o.data_ptr(), early_exit=False, stream=st.cuda_stream,
                                     asynchronous=True)
    rp, ci = _rows_csr(graph(400), 300, 400)
    live.append_rows(rp, ci, N + 400, probs[N + 300 : N + 400])  # no synchronisation since the enqueues
    after = live.decode_batch(x(400), early_exit=False)
    st.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref300["bits"])
    ref400 = oracle.bp_decode_batch(H, probs, msg, 1, 30, "min_sum", dtype="f32", threads=8, early_exit=False)
    assert np.array_equal(after["bits"], ref400["bits"])
    live.close()

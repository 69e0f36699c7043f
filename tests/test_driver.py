"""Host-side driver logic without a GPU (decoders injected from tests/fakes.py): RNG-stream
identity with the reference's per-position loops, the reference's doctest answers through
the batched drivers, hqc.decode statistics."""
import importlib

import numpy as np
import pytest

from fakes import OracleBp, oracle_qary_class
from helpers import S

drv = importlib.import_module("sca-ldpc_amd.driver")


def test_errors_provider_stream_identity(golden):
    """get_errors(runs, n) == runs*n sequential get_error calls, for every distribution kind."""
    for dist in (None, golden["distr_files"]["binary_distr"], golden["distr_files"]["qary_distr"]):
        a = drv.ErrorsProvider(0.05, None, np.random.RandomState(5))
        b = drv.ErrorsProvider(0.05, None, np.random.RandomState(5))
        a.error_distribution = b.error_distribution = dist
        loop = np.array([[a.get_error(i) for i in range(37)] for _ in range(11)])
        assert np.array_equal(loop, b.get_errors(11, 37))
        assert a.rng.rand() == b.rng.rand()  # same number of draws consumed


def test_errors_provider_rates(golden):
    """decode.py:47-86: empirical rates within 0.009 over 10000 draws."""
    rng = S.codes.make_random_state(0)
    ep = drv.ErrorsProvider(0.05, None, rng)
    assert abs(ep.get_errors(10000, 1).mean() - 0.05) < 0.009
    ep = drv.ErrorsProvider.from_distribution(golden["distr_files"]["binary_distr"], rng)
    e = ep.get_errors(10000, 4)
    assert (np.abs(e.mean(axis=0) - np.array([0.1, 0.3, 0.05, 0.14])) < 0.009).all()
    assert ep.get_binary_channel_probs(6) == [0.1, 0.3, 0.05, 0.14, 0.1, 0.3]
    ep = drv.ErrorsProvider.from_distribution(golden["distr_files"]["qary_distr"], rng)
    e = ep.get_errors(10000, 2)
    for i, expect in enumerate([{-1: 0.2, 0: 0.5, 1: 0.3}, {-1: 0.1, 0: 0.6, 1: 0.3}]):
        for val, p in expect.items():
            assert abs((e[:, i] == val).mean() - p) < 0.009
    with pytest.raises(ValueError):
        ep.get_binary_channel_probs(4)


def test_official_example_doctest():
    """decode.py:139-149 through the batched driver: 100/100."""
    assert drv.official_example(0, 100, error_rate=0.05, bp_decoder=OracleBp) == 100


def test_config1_regular_ldpc_code(golden, tmp_path):
    """BASELINE config 1: main.py regular_ldpc_code, noise from binary_distr.txt, seed 0.
    The batched driver must equal the reference's one-at-a-time loop (restated here)."""
    f = tmp_path / "binary_distr.txt"
    f.write_text("\n".join(str(r[0]) for r in golden["distr_files"]["binary_distr"]))
    runs = 8
    got = drv.regular_ldpc_code(0, runs, error_rate=None, error_file=str(f), bp_decoder=OracleBp)
    # reference order: H first, then per run n draws (main.py:189-208, decode.py:162-177)
    rng = S.codes.make_random_state(0)
    ep = drv.ErrorsProvider(None, str(f), rng)
    g = S.codes.make_regular_ldpc_graph(300, 150, 3, 6, rng)
    dec = OracleBp(g, max_iter=g.n, bp_method="product_sum", channel_probs=ep.get_binary_channel_probs(g.n))
    ok = 0
    for _ in range(runs):
        error = np.array([ep.get_error(i) for i in range(g.n)], dtype=np.uint8)
        out = dec.decode_batch(g.syndrome(error)[None], input_vector_type="syndrome")["bits"][0]
        ok += int((out == error).all())
    assert got == ok


def test_qary_fer_doctest():
    """decode.py:192-209 through the batched driver: seed 1, 1 run -> 1."""
    rng = S.codes.make_random_state(1)
    H = S.codes.make_regular_ldpc_identity_graph(300, 150, 3, 6, rng)
    assert drv.simulate_frame_error_rate_rust(H, 1, 0.005, 1, rng, 1, decoder_class=oracle_qary_class) == 1


@pytest.mark.parametrize("which,all_checks", [("toy", True), ("full", False)])
def test_hqc_decode_doctests_and_stats(golden, which, all_checks):
    from test_oracle_pins import sparse_times_sparse

    t = golden["hqc_decode_tests"][which]
    N, y, r1 = t["N"], t["y_sparse"], t["first_row"]
    yr = set(sparse_times_sparse(y, r1, N))
    bits = [b for b in range(N) if all_checks or b in yr]
    checks = [(b in yr, 1.0) for b in bits]
    ok, stats = drv.hqc_decode(N, S.codes.hqc_check_graph(r1, N, bits), checks, y, bp_decoder=OracleBp)
    assert ok is golden["hqc_decode_tests"]["expected"][which]
    assert stats == {"checks": len(bits), "unsatisfied": sum(c for c, _ in checks), "good_flips": len(y),
                     "bad_flips": 0, "found_bad_satisfied_checks": 0, "found_bad_unsatisfied_checks": 0,
                     "success": True}


def test_hqc_stats_counters():
    # N=6, y={1,4}; decoded y' = {1,2}; checks c=[1,0,1], decoded check part [0,1,1]
    ok, st = drv.hqc_stats(6, np.array([0, 1, 1, 0, 0, 0, 0, 1, 1]), np.array([1, 0, 1], dtype=np.uint8), [1, 4])
    assert not ok and st["good_flips"] == 1 and st["bad_flips"] == 1 and st["unsatisfied"] == 2
    assert st["found_bad_satisfied_checks"] == 1 and st["found_bad_unsatisfied_checks"] == 1

"""The batched drivers on the real HIP decoders: the reference's doctest answers, and
equality with the same driver run on the oracle-backed decoder."""
import importlib

import numpy as np
import pytest

from fakes import OracleBp, oracle_qary_class
from helpers import S

pytestmark = pytest.mark.gpu
drv = importlib.import_module("sca-ldpc_amd.driver")


def test_official_example_doctest():
    assert drv.official_example(0, 100, error_rate=0.05) == 100  # decode.py:139-149


def test_config1_and_siblings_match_oracle_driver(golden, tmp_path):
    """BASELINE config 1 (+ the sibling commands of main.py:189-276) on the GPU decoder vs the
    same driver on the float64 oracle decoder: identical success counts."""
    f = tmp_path / "binary_distr.txt"
    f.write_text("\n".join(str(r[0]) for r in golden["distr_files"]["binary_distr"]))
    for cmd, kw in ((drv.regular_ldpc_code, dict(error_file=str(f))), (drv.regular_ldpc_code, dict(error_rate=0.03)),
                    (drv.regular_ldpc_code_identity, dict(error_rate=0.05)), (drv.qc_ldpc_code, dict(error_rate=0.02))):
        a = cmd(0, 40, **kw)
        b = cmd(0, 40, bp_decoder=OracleBp, **kw)
        assert a == b, (cmd.__name__, kw, a, b)


def test_qary_fer_doctest():
    rng = S.codes.make_random_state(1)
    H = S.codes.make_regular_ldpc_identity_graph(300, 150, 3, 6, rng)
    assert drv.simulate_frame_error_rate_rust(H, 1, 0.005, 1, rng, 1) == 1  # decode.py:192-209


def test_qary_fer_many_runs_matches_oracle_driver():
    r1, r2 = S.codes.make_random_state(4), S.codes.make_random_state(4)
    H = S.codes.make_regular_ldpc_identity_graph(300, 150, 3, 6, r1)
    S.codes.make_regular_ldpc_identity_graph(300, 150, 3, 6, r2)
    a = drv.simulate_frame_error_rate_rust(H, 1, 0.008, 60, r1, 1)
    b = drv.simulate_frame_error_rate_rust(H, 1, 0.008, 60, r2, 1, decoder_class=oracle_qary_class)
    assert a == b and 0 < a < 60


@pytest.mark.parametrize("which,all_checks", [("toy", True), ("full", False)])
def test_hqc_decode(golden, which, all_checks):
    from test_oracle_pins import sparse_times_sparse

    t = golden["hqc_decode_tests"][which]
    N, y, r1 = t["N"], t["y_sparse"], t["first_row"]
    yr = set(sparse_times_sparse(y, r1, N))
    bits = [b for b in range(N) if all_checks or b in yr]
    checks = [(b in yr, 1.0) for b in bits]
    Hin = S.codes.hqc_check_graph(r1, N, bits)
    ok, stats = drv.hqc_decode(N, Hin, checks, y)
    ok2, stats2 = drv.hqc_decode(N, Hin, checks, y, bp_decoder=OracleBp)
    assert ok is golden["hqc_decode_tests"]["expected"][which] and stats == stats2


def test_incremental_accumulator_full_example(golden):
    """hqc.py:1277-1311 driven through the incremental accumulator on the real decoder:
    rows appended one by one, decode every 50 checks, success once enough are in."""
    from test_oracle_pins import sparse_times_sparse

    t = golden["hqc_decode_tests"]["full"]
    N, y, r1 = t["N"], t["y_sparse"], t["first_row"]
    yr = sparse_times_sparse(y, r1, N)
    acc = drv.HqcCheckAccumulator(N, r1, len(y), decode_every=50)
    found = acc.add_checks([(b, 1.0) for b in yr], 1, y)
    if not found:  # the doctest decodes once with all of them
        found = acc.decode(y)
    assert found
    assert [r["checks"] for r in acc.decoder_stats][:3] == [50, 100, 150]
    assert acc.decoder_stats[-1]["good_flips"] == len(y) and acc.decoder_stats[-1]["bad_flips"] == 0


def test_command_bodies_reproduce_the_reference_drivers_answers_on_the_gpu(golden, tmp_path):
    """The same eight main.py command-body runs (tests/test_driver.py) on the HIP decoder: the success counts
    of the REFERENCE's driver (0 / 40 / 0 / 8 / 0 / 1 / 40 / 40 of 40 frames; fp32 tanh rule here, float64
    ratio domain there)."""
    from test_driver import _command_body_cases

    for key, want, run in _command_body_cases(golden, tmp_path):
        assert run() == want, key

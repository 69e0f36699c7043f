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


def test_incremental_accumulator_equals_batch_hqc_decode(golden, tmp_path):
    """8f-3: checks appended one at a time (sparse) decode exactly like hqc_decode on the
    stacked matrix, stats rows and CSV format as the reference writes them."""
    from test_oracle_pins import sparse_times_sparse

    t = golden["hqc_decode_tests"]["toy"]
    N, y, r1 = t["N"], t["y_sparse"], t["first_row"]
    yr = set(sparse_times_sparse(y, r1, N))
    acc = drv.HqcCheckAccumulator(N, r1, len(y), bp_decoder=OracleBp, decode_every=5)
    order = list(np.random.RandomState(1).permutation(N))
    done = acc.add_checks([(b, 1.0) for b in order if b in yr], 1, y) or acc.add_checks(
        [(b, 1.0) for b in order if b not in yr], 0, y
    )
    assert done and acc.decoder_stats[-1]["success"]
    # the graph built incrementally is the stacked [Hin | I]
    bits = [b for b in order if b in yr] + [b for b in order if b not in yr]
    R = len(acc)
    Hin = S.codes.hqc_check_graph(r1, N, bits[:R])
    assert np.array_equal(acc.graph().to_dense(), Hin.with_identity().to_dense())
    ok, stats = drv.hqc_decode(N, Hin, acc.checks, y, bp_decoder=OracleBp)
    last = acc.decoder_stats[-1]
    assert ok and all(stats[k] == last[k] for k in stats)
    # CSV: header once, rows appended (main.py:150-156)
    f = tmp_path / "stats.csv"
    drv.write_decoder_stats_csv(str(f), acc.decoder_stats, "lbl", "Hqc128", 3, (1.0, 1.0))
    drv.write_decoder_stats_csv(str(f), acc.decoder_stats[:1], "lbl", "Hqc128", 3, (1.0, 1.0))
    lines = f.read_text().strip().splitlines()
    assert lines[0].split(",") == drv.STATS_COLUMNS and len(lines) == 1 + len(acc.decoder_stats) + 1
    import pandas as pd

    df = pd.read_csv(f)  # what visualize.load_data does first (visualize.py:104)
    assert {"good_flips", "bad_flips", "found_bad_unsatisfied_checks", "found_bad_satisfied_checks", "oracle_calls",
            "unsatisfied"} <= set(df.columns)


def test_kyber_channel_probabilities_layout():
    s = np.random.RandomState(0).dirichlet(np.ones(5), size=(3, 256))
    ss = np.random.RandomState(1).dirichlet(np.ones(25), size=512)
    co, cs = drv.kyber_channel_probabilities(s, ss, 6, 2)
    assert co.shape == (768, 5) and cs.shape == (512, 25) and co.dtype == np.float32
    assert np.allclose(co[256 + 7], s[1][7]) and np.allclose(cs[3], ss[3][::-1])


def _command_body_cases(golden, tmp_path):
    """(name, callable returning the success count) for the eight main.py command-body runs whose answers the
    REFERENCE's own driver gave in the build container (tests/golden/make_call_protocol.py)."""
    import json, os

    proto = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "call_protocol.json")))
    body = proto["answers"]["main_py_command_bodies"]
    f = tmp_path / "binary_distr.txt"
    f.write_text("\n".join(str(r[0]) for r in golden["distr_files"]["binary_distr"]))
    cmds = {"regular_ldpc_code": drv.regular_ldpc_code, "regular_ldpc_code_identity": drv.regular_ldpc_code_identity,
            "qc_ldpc_code": drv.qc_ldpc_code, "official_example": drv.official_example}
    for key, want in body["successes"].items():
        cname, noise = key.split(" ")
        kw = dict(error_rate=0.0, error_file=str(f)) if noise.startswith("error_file") else dict(error_rate=0.03)
        yield key, want, (lambda c=cmds[cname], k=kw, **extra: c(body["seed"], body["runs"], **k, **extra))


def test_command_bodies_reproduce_the_reference_drivers_answers(golden, tmp_path):
    """main.py:189-276 (BASELINE config 1 = regular_ldpc_code with binary_distr.txt): the build's driver on the
    float64 oracle decoder returns the success counts the reference's own generators + ErrorsProvider +
    simulate_frame_error_rate returned on the same decoder -- same code construction, same noise stream,
    same loop."""
    for key, want, run in _command_body_cases(golden, tmp_path):
        assert run(bp_decoder=OracleBp) == want, key

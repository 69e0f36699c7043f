"""K6 oracle: the Philox4x32-10 restatement against the Random123 known-answer vectors,
and the statistical contract of the reference's noise model (decode.py:47-86)."""
import numpy as np


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, out in kat:
        assert oracle.philox4x32_10(ctr, key) == out


def test_bernoulli_rates_and_edges(oracle, golden):
    probs = np.array([r[0] for r in golden["distr_files"]["binary_distr"]] + [0.0, 1.0])
    e = oracle.mc_bernoulli(seed=7, first=0, batch=20000, length=6, probs=probs)
    assert (np.abs(e[:, :4].mean(axis=0) - probs[:4]) < 0.009).all()  # the reference's own tolerance
    assert not e[:, 4].any() and e[:, 5].all()
    # a trial's draw depends only on (seed, global index)
    assert np.array_equal(e[100:164], oracle.mc_bernoulli(7, 100, 64, 6, probs))
    assert not np.array_equal(e[:64], oracle.mc_bernoulli(8, 0, 64, 6, probs))


def test_hqc_secret_distinct_uniform(oracle):
    y = oracle.mc_hqc_secret(seed=3, first=0, batch=2000, N=499, omega=20)
    assert ((y >= 0) & (y < 499)).all()
    assert all(len(set(r)) == 20 for r in y)
    counts = np.bincount(y.ravel(), minlength=499)
    assert abs(counts.mean() - 2000 * 20 / 499) < 1e-9 and counts.std() < 3 * np.sqrt(2000 * 20 / 499)

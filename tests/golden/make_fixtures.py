#!/usr/bin/env python3
"""Mint golden fixtures from the *importable* parts of the reference.

Runs ONLY in the build container (needs /root/reference); its outputs
(`tests/golden/*.json`) are committed and are the only thing that travels.
The reference's decoders cannot run here (`ldpc==0.1.3` is an absent PyPI
dependency, `simulate_rs` is Rust with no toolchain), so what is minted is:

  * the parity-check matrices / first rows the reference's generators emit
    for the seeds its own doctests and the BASELINE configs use
    (simulate/make_code.py, simulate/distance_spectrum.py), stored sparse;
  * the inputs of the two liboqs-free HQC decode tests (hqc.py:1229-1311);
  * the data files the reference's tests hold (parity_check_150_450.txt,
    binary_distr.txt, qary_distr.txt), as data.

`coloredlogs` (a log prettifier imported by simulate/utils.py) is not installed;
an empty module object is placed in sys.modules for the import — it takes no
part in any arithmetic.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/simulate-with-python"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.modules.setdefault("coloredlogs", types.ModuleType("coloredlogs"))
    sys.path.insert(0, REF)
    from simulate import make_code, distance_spectrum, utils  # noqa

    return make_code, distance_spectrum, utils


def coo(H):
    H = np.asarray(H)
    r, c = np.nonzero(H)
    return {
        "shape": [int(H.shape[0]), int(H.shape[1])],
        "rows": [int(x) for x in r],
        "cols": [int(x) for x in c],
        "vals": [int(x) for x in H[r, c]],
    }


def dump(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {name} ({os.path.getsize(path)} B)")


def main():
    make_code, ds, utils = _import_reference()
    rs = utils.make_random_state

    # --- generator known answers (doctest seeds) + config matrices ---------
    gen = {}
    gen["fixed_weight_vec_10_3_s0"] = [int(x) for x in make_code.fixed_weight_vec(10, 3, rs(0))]
    gen["qc_6_2_2_s0"] = coo(make_code.make_qc_parity_check_matrix(6, 2, 2, rs(0)))
    gen["regular_6_4_2_3_s0"] = coo(make_code.make_regular_ldpc_parity_check_matrix(6, 4, 2, 3, rs(0)))
    gen["regular_identity_6_4_2_3_s0"] = coo(
        make_code.make_regular_ldpc_parity_check_matrix_identity(6, 4, 2, 3, rs(0))
    )
    gen["random_ldpc_10_3_s0"] = coo(make_code.make_random_ldpc_parity_check_matrix(10, 3, rs(0)))
    gen["random_ldpc_identity_10_3_s0"] = coo(
        make_code.make_random_ldpc_parity_check_matrix_with_identity(10, 3, rs(0))
    )
    rng = rs(0)
    a1 = ds.gen_array_ds_multiplicity(10, 3, 1, rng)
    a2 = ds.gen_array_ds_multiplicity(10, 4, 2, rng)
    gen["ds_10_3_1_then_10_4_2_s0"] = {
        "a1": [int(x) for x in a1],
        "ds1": [int(x) for x in ds.calc_ds(a1)],
        "a2": [int(x) for x in a2],
        "ds2": [int(x) for x in ds.calc_ds(a2)],
    }
    # BASELINE config 1 (main.py:189-208, seed 0) and its siblings
    gen["regular_300_150_3_6_s0"] = coo(make_code.make_regular_ldpc_parity_check_matrix(300, 150, 3, 6, rs(0)))
    gen["regular_identity_300_150_3_6_s0"] = coo(
        make_code.make_regular_ldpc_parity_check_matrix_identity(300, 150, 3, 6, rs(0))
    )
    # q-ary doctest H (decode.py:192-209, seed 1) = BASELINE config 4
    gen["regular_identity_300_150_3_6_s1"] = coo(
        make_code.make_regular_ldpc_parity_check_matrix_identity(300, 150, 3, 6, rs(1))
    )
    # the same doctest continues to draw noise from the same rng; record the
    # first draws after H construction so the driver restatement can be pinned
    rng = rs(1)
    make_code.make_regular_ldpc_parity_check_matrix_identity(300, 150, 3, 6, rng)
    gen["rand_after_regular_identity_300_150_3_6_s1"] = [float(x) for x in rng.rand(1350)]
    gen["qc_500_3_2_s0"] = coo(make_code.make_qc_parity_check_matrix(500, 3, 2, rs(0)))
    # Kyber-shaped q-ary QC matrix (kyber.py:67-74 -> make_code.py:72-94)
    gen["qary_qc_256_6_3_s0_cb2"] = coo(make_code.make_qary_qc_parity_check_matrix(256, 6, 3, rs(0), 2))
    gen["qary_qc_256_6_3_s0_cb1"] = coo(make_code.make_qary_qc_parity_check_matrix(256, 6, 3, rs(0), 1))
    dump("generators.json", gen)

    # --- HQC first rows (distance-spectrum multiplicity <= 1), seed 0 -------
    rows = {}
    for (N, W) in [(17669, 20), (17669, 50), (35851, 50), (57637, 50), (57637, 60)]:
        a = ds.gen_array_ds_multiplicity(N, W, 1, rs(0))
        rows[f"N{N}_W{W}_s0"] = [int(i) for i in np.nonzero(a)[0]]
        print(N, W, rows[f"N{N}_W{W}_s0"][:4])
    dump("hqc_first_rows.json", rows)

    # --- the two liboqs-free HQC decode tests (hqc.py:1229-1311), seed 0 ----
    # toy: N=20, W=3, y=[4,5,7,9], all 20 checks, certainty 1.0
    rng = rs(0)
    first = ds.gen_array_ds_multiplicity(20, 3, 1, rng)
    toy = {"N": 20, "W": 3, "y_sparse": [4, 5, 7, 9], "first_row": [int(i) for i in np.nonzero(first)[0]]}
    # full: N=17669, W=3, OMEGA=66; y drawn first, then the first row, same rng
    rng = rs(0)
    N, OMEGA = 17669, 66
    y = rng.choice(N, OMEGA, replace=False)
    first = ds.gen_array_ds_multiplicity(N, 3, 1, rng)
    full = {
        "N": N,
        "W": 3,
        "OMEGA": OMEGA,
        "y_sparse": [int(i) for i in y],
        "first_row": [int(i) for i in np.nonzero(first)[0]],
    }
    dump("hqc_decode_tests.json", {"toy": toy, "full": full, "expected": {"toy": True, "full": True}})

    # --- data files held by the reference's tests ---------------------------
    H = np.loadtxt(os.path.join(REF, "simulate_rs/benches/parity_check_150_450.txt"), dtype=np.int64)
    dump("parity_check_150_450.json", coo(H))

    def read_distr(p):
        import re

        out = []
        with open(p) as f:
            for line in f:
                line = line.strip()
                if line:
                    out.append([float(x) for x in re.split("[, ]+", line)])
        return out

    dump(
        "distr_files.json",
        {
            "binary_distr": read_distr(os.path.join(REF, "binary_distr.txt")),
            "qary_distr": read_distr(os.path.join(REF, "qary_distr.txt")),
        },
    )


if __name__ == "__main__":
    main()

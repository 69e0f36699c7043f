#!/usr/bin/env python3
"""Run the reference's OWN drivers, unchanged, on the drop-in modules -- and record how they call them.

BUILD CONTAINER ONLY (needs /root/reference, which never travels; no GPU here).  What this proves and
produces:

  * `sca-ldpc_amd/dropin` goes FIRST on sys.path, then the reference's `simulate-with-python`; the
    reference's `simulate/decode.py` and `simulate/hqc.py` are imported as they stand.  Their
    `from ldpc import bp_decoder`, `from ldpc.codes import rep_code`, `from ldpc.code_util import
    get_code_parameters`, `from simulate_rs import Hqc128, Hqc192, Hqc256`, `getattr(simulate_rs,
    "DecoderN450R150V3C7B1")` all resolve to this repository's drop-in modules.
  * There is no GPU in this container, so the two decoder classes the drop-ins export are replaced, for
    this script only, by RECORDING doubles backed by the CPU oracle (float64 ratio-domain product-sum =
    what ldpc 0.1.3 runs; the f32 q-ary restatement of decoder.rs).  Every call is first bound against
    the signature of the REAL HIP class (`inspect.signature(...).bind`), so a call form the real class
    would reject fails here.
  * The four liboqs-free entry points the reference's doctests pin are run:
        simulate_frame_error_rate          (decode.py:139-149)  -> 100
        simulate_frame_error_rate_rust     (decode.py:192-209)  -> 1
        test_hqc_decode_toy_example(0)     (hqc.py:1229-1274)   -> True
        test_hqc_decode_full_example(0)    (hqc.py:1277-1311)   -> True
  * Every constructor / decode / min_sum call is written to tests/golden/call_protocol.json: argument
    names, Python types, dtypes, shapes, and (sparse) values with the value returned.  The `-m gpu`
    test tests/test_call_protocol_gpu.py replays that protocol against the real HIP classes.

`coloredlogs` (a log prettifier imported by simulate/utils.py) is not installed; an empty module object is
placed in sys.modules for the import, as tests/golden/make_fixtures.py does -- it takes no part in anything.
"""
import inspect
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/simulate-with-python"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CALLS = []
RECORD = [True]  # the four pinned entry points are recorded call by call; the command-body sweeps below only by their answers


def describe(x):
    """name-free description of one argument: Python type, dtype, shape."""
    if isinstance(x, np.ndarray):
        return {"type": "ndarray", "dtype": str(x.dtype), "shape": list(x.shape)}
    if isinstance(x, (list, tuple)):
        inner = type(x[0]).__name__ if len(x) else None
        return {"type": type(x).__name__, "len": len(x), "item_type": inner}
    return {"type": type(x).__name__, "value": x if isinstance(x, (int, float, str, bool, type(None))) else repr(x)}


def coo(H):
    H = np.asarray(H)
    r, c = np.nonzero(H)
    return {"shape": [int(H.shape[0]), int(H.shape[1])], "dtype": str(H.dtype), "rows": [int(v) for v in r],
            "cols": [int(v) for v in c], "vals": [int(v) for v in H[r, c]]}


def runs_of(values):
    """[(value, count)] -- priors come in long runs of one value"""
    out = []
    for v in values:
        v = float(v)
        if out and out[-1][0] == v:
            out[-1][1] += 1
        else:
            out.append([v, 1])
    return out


def main():
    import importlib

    sys.modules.setdefault("coloredlogs", types.ModuleType("coloredlogs"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "sca-ldpc_amd", "dropin"))  # FIRST: `import ldpc` / `import simulate_rs` land here
    import ldpc  # noqa: E402  the drop-in
    import simulate_rs  # noqa: E402  the drop-in

    assert ldpc.__file__.startswith(os.path.join(ROOT, "sca-ldpc_amd", "dropin")), ldpc.__file__
    assert simulate_rs.__file__.startswith(os.path.join(ROOT, "sca-ldpc_amd", "dropin")), simulate_rs.__file__
    from oracle import pyoracle

    S = importlib.import_module("sca-ldpc_amd")
    real_bp = ldpc.bp_decoder  # sca-ldpc_amd.bp.BpDecoder
    real_sig = inspect.signature(real_bp.__init__)
    qary = importlib.import_module("sca-ldpc_amd.qary")

    class RecordingBp:
        def __init__(self, *args, **kwargs):
            bound = real_sig.bind(None, *args, **kwargs)  # the REAL class must accept this call form
            bound.apply_defaults()
            a = dict(bound.arguments)
            a.pop("self")
            H = np.asarray(a["parity_check_matrix"])
            self.rec = {
                "class": "ldpc.bp_decoder",
                "positional": len(args),
                "keywords": sorted(kwargs),
                "args": {k: describe(v) for k, v in bound.arguments.items() if k != "self" and (k in kwargs or k == "parity_check_matrix")},
                "H": coo(H),
                "max_iter": int(a["max_iter"]),
                "bp_method": a["bp_method"],
                "error_rate": a["error_rate"],
                "channel_probs_runs": runs_of(a["channel_probs"]) if a["channel_probs"][0] is not None else None,
                "decode": [],
            }
            if RECORD[0]:
                CALLS.append(self.rec)
            self.g = S.TannerGraph.from_dense(H)
            cp = a["channel_probs"]
            self.probs = np.asarray(cp, dtype=np.float64) if cp[0] is not None else np.full(self.g.n, float(a["error_rate"]))
            self.max_iter = int(a["max_iter"]) or self.g.n

        def decode(self, v):
            v = np.asarray(v) if not isinstance(v, np.ndarray) else v
            kind = 0 if v.shape[0] == self.g.m else 1
            with np.errstate(divide="ignore", invalid="ignore"):
                r = pyoracle.bp_decode_batch(self.g, self.probs, (v[None, :] & 1).astype(np.uint8), kind, self.max_iter,
                                             "product_sum", dtype="f64", threads=1)
            out = r["bits"][0].astype(int)
            if RECORD[0]:
                self.rec["decode"].append({"arg": describe(v), "ones_in": [int(i) for i in np.flatnonzero(v)],
                                       "ret": describe(out), "ones_out": [int(i) for i in np.flatnonzero(out)],
                                       "converged": int(r["converged"][0]), "iters": int(r["iters"][0])})
            return out

    def recording_qary(name):
        real_cls = qary.decoder_class(name)  # the real class exists for this name, or AttributeError as the reference expects
        ctor_sig = inspect.signature(real_cls.__init__)
        ms_sig = inspect.signature(real_cls.min_sum)

        class RecordingQary:
            def __init__(self, *args, **kwargs):
                b = ctor_sig.bind(None, *args, **kwargs)
                H = b.arguments["py_parity_check"]
                assert isinstance(H, np.ndarray) and H.dtype == np.int8  # PyReadonlyArray2<i8>, pydecoder.rs:24
                self.g = S.TannerGraph.from_dense(H)
                self.it = int(b.arguments["iterations"])
                self.rec = {"class": f"simulate_rs.{name}", "positional": len(args), "keywords": sorted(kwargs),
                            "args": {k: describe(v) for k, v in b.arguments.items() if k != "self"}, "H": coo(H),
                            "iterations": self.it, "min_sum": []}
                CALLS.append(self.rec)

            def min_sum(self, *args, **kwargs):
                b = ms_sig.bind(None, *args, **kwargs)
                p = b.arguments["py_channel_output"]
                out = [int(x) for x in pyoracle.qary_min_sum_batch(self.g, real_cls.Q, p[None].astype(np.float32), self.it)[0]]
                rows, idx = np.unique(p, axis=0, return_inverse=True)
                self.rec["min_sum"].append({"arg": describe(p), "distinct_rows": [[float(x) for x in r] for r in rows],
                                            "row_of_variable": [int(i) for i in np.asarray(idx).reshape(-1)], "ret": describe(out),
                                            "nonzero_out": {int(i): int(out[i]) for i in range(len(out)) if out[i]}})
                return out

        RecordingQary.__name__ = name
        return RecordingQary

    ldpc.bp_decoder = RecordingBp
    simulate_rs.__getattr__ = lambda name: recording_qary(name) if name.startswith("Decoder") else (_ for _ in ()).throw(AttributeError(name))

    # ---- the reference's own modules, as they stand -------------------------------------------------
    from simulate import decode as ref_decode  # noqa: E402
    from simulate import hqc as ref_hqc  # noqa: E402
    from simulate import make_code, utils  # noqa: E402

    assert ref_decode.__file__.startswith(REF) and ref_hqc.__file__.startswith(REF)
    assert ref_decode.bp_decoder is RecordingBp and ref_hqc.bp_decoder is RecordingBp
    answers = {}

    def mark(driver, n0):
        for c in CALLS[n0:]:
            c["driver"] = driver

    # decode.py:139-149
    n0 = len(CALLS)
    from ldpc.codes import rep_code

    rng = utils.make_random_state(0)
    ep = ref_decode.ErrorsProvider(0.05, None, rng)
    answers["simulate_frame_error_rate"] = int(ref_decode.simulate_frame_error_rate(rep_code(13), ep, 100, rng))
    mark("simulate_frame_error_rate (decode.py:139-149)", n0)
    # decode.py:192-209
    n0 = len(CALLS)
    rng = utils.make_random_state(1)
    H = make_code.make_regular_ldpc_parity_check_matrix_identity(300, 150, 3, 6, rng)
    answers["simulate_frame_error_rate_rust"] = int(ref_decode.simulate_frame_error_rate_rust(H, 1, 0.005, 1, rng, 1))
    mark("simulate_frame_error_rate_rust (decode.py:192-209)", n0)
    # hqc.py:1229-1274
    n0 = len(CALLS)
    answers["test_hqc_decode_toy_example"] = bool(ref_hqc.test_hqc_decode_toy_example(0))
    mark("test_hqc_decode_toy_example (hqc.py:1229-1274)", n0)
    # hqc.py:1277-1311 (dense 17669 x 17669 circulant inside: ~2.5 GB for a few seconds)
    n0 = len(CALLS)
    answers["test_hqc_decode_full_example"] = bool(ref_hqc.test_hqc_decode_full_example(0))
    mark("test_hqc_decode_full_example (hqc.py:1277-1311)", n0)

    # ---- the bodies of main.py's four FER commands (main.py:189-276; BASELINE config 1 is the first with
    # --error-file binary_distr.txt), executed with the reference's own generators, ErrorsProvider and
    # simulate_frame_error_rate exactly as those bodies call them.  main.py itself is not imported (it pulls in
    # plotting and liboqs modules that are not installed); only the success counts are kept: the build's own
    # driver must reproduce them on the HIP decoder (tests/test_driver_gpu.py).
    RECORD[0] = False
    bodies = {}
    runs = 40
    cmds = {
        "regular_ldpc_code": lambda rng: make_code.make_regular_ldpc_parity_check_matrix(300, 150, 3, 6, rng),
        "regular_ldpc_code_identity": lambda rng: make_code.make_regular_ldpc_parity_check_matrix_identity(300, 150, 3, 6, rng),
        "qc_ldpc_code": lambda rng: make_code.make_qc_parity_check_matrix(block_len=500, column_weight=3, num_blocks=2, rng=rng),
        "official_example": lambda rng: rep_code(13),
    }
    noise = {"error_file=binary_distr.txt": dict(error_rate=0.0, error_file=os.path.join(REF, "binary_distr.txt")),
             "error_rate=0.03": dict(error_rate=0.03, error_file=None)}
    for cname, mk in cmds.items():
        for nname, nz in noise.items():
            rng = utils.make_random_state(0)  # --seed 0
            ep = ref_decode.ErrorsProvider(nz["error_rate"], nz["error_file"], rng)  # as the command bodies do, before H
            H = mk(rng)
            bodies[f"{cname} {nname}"] = int(ref_decode.simulate_frame_error_rate(H, ep, runs, rng))
    RECORD[0] = True
    answers["main_py_command_bodies"] = {"seed": 0, "runs": runs, "successes": bodies}
    print("command bodies:", bodies)

    expected = {"simulate_frame_error_rate": 100, "simulate_frame_error_rate_rust": 1, "test_hqc_decode_toy_example": True,
                "test_hqc_decode_full_example": True}
    assert {k: v for k, v in answers.items() if k != "main_py_command_bodies"} == expected, answers
    out = {"note": "minted by tests/golden/make_call_protocol.py from the reference's own drivers running on the drop-in "
                   "modules with oracle-backed recording doubles; data only (call forms, inputs, returned values)",
           "answers": answers, "calls": CALLS}
    path = os.path.join(HERE, "call_protocol.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"answers {answers}; {len(CALLS)} decoder objects; wrote call_protocol.json ({os.path.getsize(path)} B)")


if __name__ == "__main__":
    main()

"""Replay of tests/golden/call_protocol.json against the real HIP classes.

The protocol was recorded in the build container by tests/golden/make_call_protocol.py: the reference's OWN
drivers (simulate/decode.py, simulate/hqc.py -- unchanged, imported from /root/reference) ran on this
repository's drop-in modules (`sca-ldpc_amd/dropin` first on sys.path) with oracle-backed recording doubles
in place of the two decoder classes, and reproduced the four answers its doctests pin (100, 1, True,
True).  The reference never travels; what travels is the record of HOW it calls the decoders -- argument
names, positional / keyword form, Python types, dtypes, shapes, values -- and what came back.  Here the same
calls are made, in the same form, on the real classes reached the way the reference reaches them
(`from ldpc import bp_decoder`, `getattr(simulate_rs, name)`), and must return the same types and the same
decisions (hard decisions of float64 product-sum vs fp32 tanh rule: these trials all converge)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROTOCOL = json.load(open(os.path.join(ROOT, "tests", "golden", "call_protocol.json")))


def dense(c):
    H = np.zeros(c["shape"], dtype=np.dtype(c["dtype"]))
    H[c["rows"], c["cols"]] = c["vals"]
    return H


def test_protocol_answers_and_call_forms_bind_to_the_real_signatures():
    """CPU side: the recorded answers are the reference's doctest values, and every recorded call form
    binds to the real classes' signatures (no GPU needed for that)."""
    import inspect

    pinned = {k: v for k, v in PROTOCOL["answers"].items() if k != "main_py_command_bodies"}
    assert pinned == {"simulate_frame_error_rate": 100, "simulate_frame_error_rate_rust": 1,
                      "test_hqc_decode_toy_example": True, "test_hqc_decode_full_example": True}
    assert len(PROTOCOL["answers"]["main_py_command_bodies"]["successes"]) == 8  # (reproduced in tests/test_driver*.py)
    bp = importlib.import_module("sca-ldpc_amd.bp")
    qary = importlib.import_module("sca-ldpc_amd.qary")
    for c in PROTOCOL["calls"]:
        if c["class"] == "ldpc.bp_decoder":
            sig = inspect.signature(bp.bp_decoder.__init__)
            assert c["positional"] == 1 and set(c["keywords"]) <= set(sig.parameters)
            assert inspect.signature(bp.bp_decoder.decode).parameters.keys() >= {"input_vector"}
        else:
            cls = qary.decoder_class(c["class"].split(".")[1])
            assert c["positional"] == 2 and not c["keywords"]
            assert (cls.N, cls.R) == (c["H"]["shape"][1], c["H"]["shape"][0])


@pytest.mark.gpu
def test_replay_on_the_hip_classes(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "sca-ldpc_amd", "dropin"))
    for m in ("ldpc", "ldpc.codes", "ldpc.code_util", "simulate_rs"):
        sys.modules.pop(m, None)
    import ldpc  # the drop-in, as `from ldpc import bp_decoder` finds it
    import simulate_rs

    assert ldpc.__file__.startswith(os.path.join(ROOT, "sca-ldpc_amd", "dropin"))
    successes = {}
    for c in PROTOCOL["calls"]:
        H = dense(c["H"])
        if c["class"] == "ldpc.bp_decoder":
            kw = {}
            for k in c["keywords"]:
                a = c["args"][k]
                if k == "channel_probs":
                    if c["channel_probs_runs"] is None:
                        kw[k] = [None]
                    else:
                        vals = np.concatenate([np.full(n, v, dtype=np.float64) for v, n in c["channel_probs_runs"]])
                        kw[k] = vals if a["type"] == "ndarray" else list(vals)
                else:
                    kw[k] = a["value"]
            assert H.dtype == np.dtype(c["args"]["parity_check_matrix"]["dtype"])
            with np.errstate(divide="ignore"):
                dec = ldpc.bp_decoder(H, **kw)  # dense int64 matrix, positional, as decode.py:155 / hqc.py:694 pass it
            ok = 0
            for d in c["decode"]:
                v = np.zeros(d["arg"]["shape"], dtype=np.dtype(d["arg"]["dtype"]))
                v[d["ones_in"]] = 1
                out = dec.decode(v)
                assert isinstance(out, np.ndarray) and list(out.shape) == d["ret"]["shape"] and out.dtype.kind == "i"
                assert [int(i) for i in np.flatnonzero(out)] == d["ones_out"], (c["driver"], d["ones_in"])
                # (iteration counts are not part of the protocol -- the reference never reads them -- and the float64
                # ratio-domain double and the fp32 LLR kernels may settle a tie `L = 0` one iteration apart)
                assert dec.converge == d["converged"]
                ok += 1
            successes[c["driver"]] = ok
            dec.close()
        else:
            name = c["class"].split(".")[1]
            cls = getattr(simulate_rs, name)  # decode.py:227-229
            dec = cls(H, c["iterations"])  # (H.astype(np.int8), iterations), decode.py:230
            for d in c["min_sum"]:
                rows = np.asarray(d["distinct_rows"], dtype=np.float32)
                p = rows[np.asarray(d["row_of_variable"])]
                assert list(p.shape) == d["arg"]["shape"] and str(p.dtype) == d["arg"]["dtype"]
                out = dec.min_sum(p.copy())  # `channel_output.copy()`, decode.py:262
                assert isinstance(out, list) and len(out) == d["ret"]["len"] and all(isinstance(x, int) for x in out)
                assert {i: x for i, x in enumerate(out) if x} == {int(k): v for k, v in d["nonzero_out"].items()}
                assert out == [0] * len(out)  # decode.py:273: the doctest's single frame is corrected
            dec.close()
    assert successes["simulate_frame_error_rate (decode.py:139-149)"] == 100

"""Drop-in for the reference's PyO3 module `simulate_rs` (simulate_rs/src/lib.rs:27-83)
as the drivers use it: `getattr(simulate_rs, "DecoderN450R150V3C7B1")`
(simulate/decode.py:227-229), `DecoderN1280R512SW6` (simulate/kyber.py:396-402).

Any `DecoderN{N}R{R}V{DV}C{DC}B{B}` / `DecoderN{N}R{R}SW{SW}` name with check degree DC <= 8 (the
reference registers 4 and 7, lib.rs:32-75) and symbols within int8 resolves: sizes are run-time values
here, not compile-time const generics.  A name beyond that is an AttributeError at look-up, exactly as a
size the reference has not registered is.

`Hqc128/192/256` (liboqs KEM wrappers, simulate_rs/src/hqc.rs) are outside the decode path.
They exist here only so that `from simulate_rs import Hqc128, Hqc192, Hqc256`
(simulate/hqc.py:23) imports, and they answer the two static queries that need no
cryptography -- `params(what)` (simulate_rs/src/hqc.rs:34-47; the public HQC round-3
constants) and `name()` -- which is all the liboqs-free decode tests of the reference
touch (hqc.py:172-174, 1289).  Every KEM operation raises NotImplementedError.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import sub  # noqa: E402

_qary = sub("qary")


def _hqc_class(bits, N, N1, N2, delta, omega):
    consts = {"N": N, "N1": N1, "N2": N2, "N1N2": N1 * N2, "SECURITY": bits, "DELTA": delta, "OMEGA": omega}

    def _kem(*_a, **_k):
        raise NotImplementedError(
            f"simulate_rs.Hqc{bits} wraps a patched liboqs (simulate_rs/src/hqc.rs); the KEM is outside the decode "
            "path this module replaces"
        )

    def params(what):
        try:
            return consts[str(what).upper()]
        except KeyError:
            raise ValueError("No such param!")  # hqc.rs:45

    ns = {"params": staticmethod(params), "name": staticmethod(lambda: f"Hqc{bits}")}
    for m in ("keypair", "new_plaintext", "secrets_from_key", "num_rejections", "encaps", "decaps",
              "encaps_with_plaintext_and_r1", "eprime", "decode_intermediates", "decode_oracle"):
        ns[m] = staticmethod(_kem)
    return type(f"Hqc{bits}", (), ns)


Hqc128 = _hqc_class(128, 17669, 46, 384, 15, 66)
Hqc192 = _hqc_class(192, 35851, 56, 640, 16, 100)
Hqc256 = _hqc_class(256, 57637, 90, 640, 29, 131)


def __getattr__(name):
    if name.startswith("Decoder"):
        return _qary.decoder_class(name)
    raise AttributeError(name)

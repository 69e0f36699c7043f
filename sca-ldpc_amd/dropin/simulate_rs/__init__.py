"""Drop-in for the reference's PyO3 module `simulate_rs` (simulate_rs/src/lib.rs:27-83)
as the drivers use it: `getattr(simulate_rs, "DecoderN450R150V3C7B1")`
(simulate/decode.py:227-229), `DecoderN1280R512SW6` (simulate/kyber.py:396-402).

Any `DecoderN{N}R{R}V{DV}C{DC}B{B}` / `DecoderN{N}R{R}SW{SW}` name resolves: sizes are
run-time values here, not compile-time const generics.  `Hqc128/192/256` (liboqs KEM
wrappers, simulate_rs/src/hqc.rs) are outside the decode path and are not provided.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import sub  # noqa: E402

_qary = sub("qary")


def __getattr__(name):
    if name in ("Hqc128", "Hqc192", "Hqc256"):
        raise AttributeError(
            f"simulate_rs.{name} wraps a patched liboqs (simulate_rs/src/hqc.rs); it is outside the decode path "
            "this module replaces"
        )
    if name.startswith("Decoder"):
        return _qary.decoder_class(name)
    raise AttributeError(name)

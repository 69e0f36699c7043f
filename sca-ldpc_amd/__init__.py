"""sca-ldpc_amd -- MI355X-native LDPC belief-propagation decoders behind the
decoder API of atneit/SCA-LDPC's Monte-Carlo drivers.

The directory name carries a hyphen, so import it with
    scaldpc = importlib.import_module("sca-ldpc_amd")
(or put `sca-ldpc_amd/dropin` on sys.path and `import ldpc` / `import simulate_rs`
exactly as the reference's drivers do).

Only host-side helpers are imported eagerly; the HIP library is loaded on first
use of a decoder (`_lib.load()`), and that load FAILS LOUDLY when the extension
is missing -- there is no CPU fallback in the product path.
"""
from . import codes, graph  # noqa: F401
from .graph import TannerGraph  # noqa: F401

__all__ = ["TannerGraph", "codes", "graph"]

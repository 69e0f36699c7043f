"""Drop-in for the third-party `ldpc` module as the reference imports it
(`from ldpc import bp_decoder`, simulate/decode.py:2, simulate/hqc.py:24;
`from ldpc.codes import rep_code`, main.py:42; `from ldpc.code_util import
get_code_parameters`, simulate/hqc.py:1267).

Put this directory's parent (`sca-ldpc_amd/dropin`) on sys.path ahead of
site-packages and the reference's drivers run on the MI355X decoder unchanged.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import sub  # noqa: E402

bp_decoder = sub("bp").bp_decoder
from . import code_util, codes  # noqa: E402,F401

__all__ = ["bp_decoder", "codes", "code_util"]

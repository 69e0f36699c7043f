"""Test doubles: decoder classes with the product's Python surface, backed by the CPU
oracle, so that the HOST logic (drivers, sharding) can run in `-m "not gpu"` tests.
Never used by the product."""
import numpy as np

from helpers import S
from oracle import pyoracle


class OracleBp:
    def __init__(self, H, error_rate=None, max_iter=0, bp_method=0, ms_scaling_factor=1.0, channel_probs=[None],
                 input_vector_type=-1, dtype="f64"):
        self.g = S.TannerGraph.coerce(H)
        self.n, self.m = self.g.n, self.g.m
        self.max_iter = max_iter or self.n
        self.method = {"product_sum": "product_sum", "min_sum": "min_sum"}[bp_method]
        self.alpha = ms_scaling_factor
        cp = list(channel_probs)
        self.probs = np.asarray(cp, dtype=np.float64) if cp and cp[0] is not None else np.full(self.n, float(error_rate))
        self.dtype = dtype

    def decode_batch(self, inputs, max_iter=None, early_exit=True, want_llr=False, input_vector_type=None):
        x = np.asarray(inputs, dtype=np.uint8)
        kind = {"syndrome": 0, "received_vector": 1, None: 0 if x.shape[1] == self.m else 1}[input_vector_type]
        return pyoracle.bp_decode_batch(self.g, self.probs, x, kind, max_iter or self.max_iter, self.method,
                                        alpha=self.alpha, dtype=self.dtype, threads=4, early_exit=early_exit)


def oracle_qary_class(name):
    import re

    N, R, V, C, B = map(int, re.match(r"DecoderN(\d+)R(\d+)V(\d+)C(\d+)B(\d+)", name).groups())

    class _D:
        def __init__(self, H, iterations):
            self.g = S.TannerGraph.from_dense(np.asarray(H))
            self.it = iterations

        def min_sum_batch(self, pmf):
            return pyoracle.qary_min_sum_batch(self.g, 2 * B + 1, pmf, self.it, threads=4)

    return _D

"""Batched counterparts of the reference's Monte-Carlo drivers -- the callers either
side of the decode path.  Same inputs, same RNG consumption, same
success criteria and statistics as the reference, but every trial of a run goes to the
decoder in ONE batched call.

  ErrorsProvider                  simulate/decode.py:9-127
  simulate_frame_error_rate       simulate/decode.py:130-177  (binary FER loop)
  simulate_frame_error_rate_rust  simulate/decode.py:180-286  (q-ary FER loop)
  hqc_decode                      simulate/hqc.py:661-759      (input assembly + 7 stats fields)
  regular_ldpc_code & co          main.py:189-276              (the four FER commands, as functions)

The decoder classes are parameters (default: the HIP decoders) so that the host logic
can be exercised without a GPU by injecting any object with the same surface.
"""
from __future__ import annotations

import itertools
import re

import numpy as np

from . import codes
from .graph import TannerGraph


class ErrorsProvider:
    """simulate/decode.py:9-127.  `get_error(pos)` consumes exactly one `rng.rand()`;
    `get_errors(runs, n)` draws `runs*n` uniforms in the same order (row-major) and is
    therefore stream-identical to the reference's nested loops (decode.py:165-167)."""

    def __init__(self, error_rate, error_file, rng):
        self.error_rate = error_rate
        self.error_distribution = None
        self.rng = rng
        if error_file is not None:
            dist = []
            with open(error_file, "rt") as f:
                for line in f:
                    line = line.strip()
                    if line:
                        dist.append([float(x) for x in re.split("[, ]+", line)])
            self.error_distribution = dist

    @classmethod
    def from_distribution(cls, distribution, rng, error_rate=None):
        self = cls(error_rate, None, rng)
        self.error_distribution = [list(map(float, row)) for row in distribution]
        return self

    def get_error(self, pos):
        if self.error_distribution is None:
            return 1 if self.rng.rand() < self.error_rate else 0
        pr = self.error_distribution[pos % len(self.error_distribution)]
        if len(pr) == 1:
            return 1 if self.rng.rand() < pr[0] else 0
        rand = self.rng.rand()
        res = -(len(pr) // 2)
        threshold = 0
        for p in pr:
            threshold += p
            if threshold > rand:
                return res
            res += 1
        return None  # the reference falls off the loop the same way when rand >= sum(pr)

    def get_errors(self, runs, n):
        """[runs, n] errors, same uniforms in the same order as runs*n get_error calls."""
        u = self.rng.rand(runs, n)
        if self.error_distribution is None:
            return (u < self.error_rate).astype(np.int64)
        L = len(self.error_distribution)
        widths = {len(r) for r in self.error_distribution}
        if widths == {1}:
            thr = np.array([self.error_distribution[i % L][0] for i in range(n)])
            return (u < thr[None, :]).astype(np.int64)
        out = np.empty((runs, n), dtype=np.int64)
        for i in range(n):
            pr = self.error_distribution[i % L]
            if len(pr) == 1:
                out[:, i] = u[:, i] < pr[0]
            else:
                cum = np.cumsum(np.asarray(pr))  # threshold += p, `threshold > rand`
                out[:, i] = (cum[None, :] > u[:, i : i + 1]).argmax(axis=1) - len(pr) // 2
        return out

    def get_error_rate(self):
        return self.error_rate if self.error_distribution is None else None

    def get_binary_channel_probs(self, n=None):
        if self.error_distribution is None:
            return [None]
        if len(self.error_distribution[0]) != 1:
            raise ValueError("Distribution from the file isn't binary")
        if n is None:
            return [x[0] for x in self.error_distribution]
        it = itertools.cycle(self.error_distribution)
        return [next(it)[0] for _ in range(n)]


def _default_bp():
    from .bp import bp_decoder

    return bp_decoder


def simulate_frame_error_rate(H, errors_provider, runs, rng, bp_decoder=None):
    """simulate/decode.py:130-177: number of frames decoded exactly.  `rng` is the one
    held by `errors_provider` (the reference passes it twice as well)."""
    g = TannerGraph.coerce(H)
    n = g.n
    bpd = (bp_decoder or _default_bp())(
        g,
        error_rate=errors_provider.get_error_rate(),
        max_iter=n,
        bp_method="product_sum",
        channel_probs=errors_provider.get_binary_channel_probs(n),
    )
    error = errors_provider.get_errors(runs, n)
    syndrome = g.syndrome(error.astype(np.uint8))
    decoding = bpd.decode_batch(syndrome, early_exit=True, input_vector_type="syndrome")["bits"]
    return int((decoding == error).all(axis=1).sum())


def simulate_frame_error_rate_rust(H, B, error_rate, runs, rng, threads=1, decoder_class=None):
    """simulate/decode.py:180-286: all-zero codeword with noisy symbols; frames without a
    bad symbol are skipped (but their draws are consumed), success = all-zero decoding.
    `threads` is accepted for signature parity; the batch replaces the thread pool."""
    Hd = H.to_dense(np.int8) if isinstance(H, TannerGraph) else np.asarray(H)
    r, n = Hd.shape
    v = int(np.count_nonzero(Hd, axis=0).max())
    c = int(np.count_nonzero(Hd, axis=1).max())
    B = 1  # decode.py:222 overrides the argument
    BB = 2 * B + 1
    iterations = 5
    name = f"DecoderN{n}R{r}V{v}C{c}B{B}"
    if decoder_class is None:
        from .qary import decoder_class as _dc

        decoder_class = _dc
    decoder = decoder_class(name)(Hd.astype(np.int8), iterations)
    p = 1 / BB
    good = np.full(BB, p)
    bad = np.full(BB, p)
    good[[B, -1]] = [1.75 * p, 0.25 * p]
    bad[[-1, B]] = [1.75 * p, 0.25 * p]
    frames = []
    while len(frames) < runs:
        mask = rng.rand(n) < error_rate
        if not mask.any():
            continue
        frames.append(np.where(mask[:, None], bad, good).astype(np.float32))
    decoding = decoder.min_sum_batch(np.stack(frames))
    return int((decoding == 0).all(axis=1).sum())


def hqc_decode(N, Hin, checks, y_sparse, bp_decoder=None, max_iter=100):
    """simulate/hqc.py:661-759.  Hin: TannerGraph (R x N, one row per oracle answer) or a
    dense array; checks: [(value, certainty)]; returns (success, stats) with the seven
    fields `tracking.add_decoder_stats` records (hqc.py:750-758)."""
    g = TannerGraph.coerce(Hin)
    R = g.m
    H = g.with_identity()
    prob_for_one = len(y_sparse) / N
    channel_probs = np.concatenate(
        [np.full(N, prob_for_one, dtype=np.float64), np.array([1 - p for (_, p) in checks], dtype=np.float64)]
    )
    with np.errstate(divide="ignore"):
        bpd = (bp_decoder or _default_bp())(H, max_iter=max_iter, bp_method="product_sum", channel_probs=channel_probs)
    cvals = np.array([int(c) for (c, _) in checks], dtype=np.uint8)
    msg = np.concatenate([np.zeros(N, dtype=np.uint8), cvals])
    decoded = bpd.decode_batch(msg[None, :], early_exit=True, input_vector_type="received_vector")["bits"][0]
    return hqc_stats(N, decoded, cvals, y_sparse)


def hqc_stats(N, decoded, cvals, y_sparse):
    """The counters of hqc.py:709-758 from one decoded vector."""
    truth = np.zeros(N, dtype=bool)
    truth[np.asarray(list(y_sparse), dtype=np.int64)] = True
    dy = decoded[:N].astype(bool)
    dc = decoded[N:].astype(bool)
    cv = cvals.astype(bool)
    stats = {
        "checks": int(cvals.size),
        "unsatisfied": int(cv.sum()),
        "good_flips": int((dy & truth).sum()),
        "bad_flips": int((dy & ~truth).sum()),
        "found_bad_satisfied_checks": int((~cv & dc).sum()),
        "found_bad_unsatisfied_checks": int((cv & ~dc).sum()),
    }
    success = bool((dy == truth).all())
    stats["success"] = success
    return success, stats


# -- the four FER commands of main.py (as functions; the CLI itself is out of scope) ------
def regular_ldpc_code(seed, runs, error_rate=None, error_file=None, bp_decoder=None):
    """main.py:189-208 (BASELINE config 1 with error_file = binary_distr.txt)."""
    rng = codes.make_random_state(seed)
    ep = ErrorsProvider(error_rate, error_file, rng)
    H = codes.make_regular_ldpc_graph(300, 150, 3, 6, rng)
    return simulate_frame_error_rate(H, ep, runs, rng, bp_decoder)


def regular_ldpc_code_identity(seed, runs, error_rate=None, error_file=None, bp_decoder=None):
    """main.py:210-231."""
    rng = codes.make_random_state(seed)
    ep = ErrorsProvider(error_rate, error_file, rng)
    H = codes.make_regular_ldpc_identity_graph(300, 150, 3, 6, rng)
    return simulate_frame_error_rate(H, ep, runs, rng, bp_decoder)


def qc_ldpc_code(seed, runs, error_rate=None, error_file=None, bp_decoder=None):
    """main.py:233-250."""
    rng = codes.make_random_state(seed)
    ep = ErrorsProvider(error_rate, error_file, rng)
    H = codes.make_qc_parity_check_graph(500, 3, 2, rng)
    return simulate_frame_error_rate(H, ep, runs, rng, bp_decoder)


def official_example(seed, runs, error_rate=None, error_file=None, bp_decoder=None):
    """main.py:265-276."""
    rng = codes.make_random_state(seed)
    ep = ErrorsProvider(error_rate, error_file, rng)
    return simulate_frame_error_rate(codes.rep_code_graph(13), ep, runs, rng, bp_decoder)

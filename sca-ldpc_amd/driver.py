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


# -- the attack loop's side of the decode path -------------------------
class HqcCheckAccumulator:
    """The check-accumulation half of the reference's attack loop, sparse and incremental.

    The reference keeps `H` dense and grows it with `np.vstack([H, Hgen[bit_n]])` per oracle
    answer (simulate/hqc.py:885-908), then rebuilds `[H | I]` and a fresh decoder from the dense
    matrix on every decode (hqc.py:680,694; every DECODE_EVERY answers, hqc.py:972-980) -- O(R(N+R))
    per decode.  Here a check is W+1 appended CSR entries (row `bit_n` of circulant(first_row),
    then its identity column), and a decode hands the CSR arrays to the library as they stand:
    O(E) per decode, nothing dense, the same result as `hqc_decode`.
    """

    def __init__(self, N, first_row_support, weight_of_y, bp_decoder=None, max_iter=100, decode_every=0):
        self.N = int(N)
        self.k = np.sort(np.asarray(first_row_support, dtype=np.int64))
        self.omega = int(weight_of_y)
        self.bp_decoder = bp_decoder
        self.max_iter = max_iter
        self.decode_every = int(decode_every)
        self._cols = np.empty((64, self.k.size + 1), dtype=np.int32)  # row i: sorted support of Hin row i, then N + i
        self._vals = np.empty(64, dtype=np.uint8)  # measured check values
        self._cert = np.empty(64, dtype=np.float64)  # their certainties
        self._R = 0
        self.decoder_stats = []
        self.num_oracle_calls = 0
        self._previous_decoding = 0
        self._bpd = None  # the decoder kept alive between decodes (rows are appended to it)
        self._bpd_R = 0  # checks it already holds
        self._bpd_omega = None

    def __len__(self):
        return self._R

    @property
    def checks(self):
        """[(value, certainty)], the list simulate/hqc.py keeps (hqc.py:907)."""
        return [(int(v), float(c)) for v, c in zip(self._vals[: self._R], self._cert[: self._R])]

    def add_check(self, bit_n, check, certainty):
        """hqc.py:885-908: append row `bit_n` of Hgen with its measured value."""
        if self._R == self._cols.shape[0]:  # amortised doubling
            self._cols = np.concatenate([self._cols, np.empty_like(self._cols)])
            self._vals = np.concatenate([self._vals, np.empty_like(self._vals)])
            self._cert = np.concatenate([self._cert, np.empty_like(self._cert)])
        row = self._cols[self._R]
        row[:-1] = np.sort((int(bit_n) - self.k) % self.N)
        row[-1] = self.N + self._R
        self._vals[self._R] = int(bool(check))
        self._cert[self._R] = float(certainty)
        self._R += 1

    def graph(self):
        """H = [Hin | I_R] as a TannerGraph, built from the appended rows (never dense)."""
        R, W = self._R, self.k.size
        # each row: its W sorted circulant positions (< N), then its identity column N + i -- CSR as it stands
        return TannerGraph.from_csr(R, self.N + R, np.arange(R + 1, dtype=np.int64) * (W + 1), self._cols[:R].reshape(-1))

    def _decoder(self, omega):
        """The decoder for the checks accumulated so far.  The reference builds a new one from the dense
        matrix on every decode (hqc.py:680,694); here ONE decoder lives through the attack and the checks
        added since the last decode are APPENDED to it (`append_rows`: CSR tail, their identity columns and
        priors) -- same results as a fresh decoder, without re-validating, re-sorting and re-uploading the
        rows it already holds.  Decoder classes without `append_rows` (test doubles) are rebuilt."""
        R, W = self._R, self.k.size
        cls = self.bp_decoder or _default_bp()
        if self._bpd is not None and (not hasattr(self._bpd, "append_rows") or self._bpd_omega != omega or R < self._bpd_R):
            self.close()
        if self._bpd is None:
            probs = np.concatenate([np.full(self.N, omega / self.N), 1 - self._cert[:R]])
            with np.errstate(divide="ignore"):
                self._bpd = cls(self.graph(), max_iter=self.max_iter, bp_method="product_sum", channel_probs=probs)
            self._bpd_omega = omega
        elif R > self._bpd_R:
            k = R - self._bpd_R
            try:
                with np.errstate(divide="ignore"):
                    self._bpd.append_rows(np.arange(k + 1, dtype=np.int32) * (W + 1), self._cols[self._bpd_R : R].reshape(-1),
                                          self.N + R, 1 - self._cert[self._bpd_R : R])
            except Exception:
                # a failed append (a certainty outside [0, 1], no memory) must not wedge the accumulator: drop the live
                # decoder, so that the next decode builds a fresh one -- the reference's behaviour on every decode
                self.close()
                raise
        self._bpd_R = R
        return self._bpd

    def close(self):
        """Release the live decoder (device memory, stream).  The accumulator stays usable: the next decode builds a
        new decoder from all the checks.  Call it -- or use the accumulator as a context manager -- when the attack is
        over; rows are APPEND-ONLY (`add_check`), which is what lets the live decoder be extended instead of rebuilt."""
        if self._bpd is not None and hasattr(self._bpd, "close"):
            self._bpd.close()
        self._bpd = None
        self._bpd_R = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, y_sparse):
        """hqc.py:661-759 on the accumulated checks; appends the stats row (hqc.py:750-758)."""
        R = self._R
        bpd = self._decoder(len(y_sparse))
        cvals = self._vals[:R].copy()
        msg = np.concatenate([np.zeros(self.N, dtype=np.uint8), cvals])
        decoded = bpd.decode_batch(msg[None, :], early_exit=True, input_vector_type="received_vector")["bits"][0]
        success, stats = hqc_stats(self.N, decoded, cvals, y_sparse)
        row = {"checks": R, "oracle_calls": self.num_oracle_calls}
        row.update({k: v for k, v in stats.items() if k != "checks"})
        self.decoder_stats.append(row)
        return success

    def add_checks(self, bits, check_value, y_sparse):
        """hqc.py:953-984: add (bit_n, certainty) pairs, decoding every `decode_every` checks.
        Returns True as soon as a decode succeeds, else False."""
        for bit_n, certainty in bits:
            self.add_check(bit_n, check_value, certainty)
            R = self._R
            if self.decode_every and R % self.decode_every == 0 and self._previous_decoding != R:
                self._previous_decoding = R
                if self.decode(y_sparse):
                    self.close()  # the attack is over (hqc.py:976-980 returns here): give the GPU decoder back
                    return True
        return False


STATS_COLUMNS = ["label", "alg", "weight", "epsilon0", "epsilon1", "checks", "oracle_calls", "unsatisfied", "good_flips",
                 "bad_flips", "found_bad_satisfied_checks", "found_bad_unsatisfied_checks", "success"]  # fmt: skip


def write_decoder_stats_csv(path, decoder_stats, label, alg, weight, epsilon):
    """The CSV `main.py hqc_simulate --csv-output` produces (main.py:146-156 from
    hqc.py:245-264): static columns label/alg/weight/epsilon0/epsilon1, then the stats
    fields; header only when the file is new, rows appended otherwise -- the format
    `visualize.load_data` reads (visualize.py:102-119)."""
    import csv
    import os

    new = not os.path.exists(path)
    with open(path, "a" if not new else "w", newline="") as f:
        w = csv.writer(f)
        if new:
            w.writerow(STATS_COLUMNS)
        for row in decoder_stats:
            w.writerow([label, alg, weight, epsilon[0], epsilon[1]] + [row[c] for c in STATS_COLUMNS[5:]])


def kyber_channel_probabilities(s_distr, ssum_distr, sum_weight, check_blocks, eta=2, block_len=256, num_blocks=3):
    """Input layout of the Kyber decoders (simulate/kyber.py:362-376): secret-coefficient
    pmfs [num_blocks*block_len, 2*eta+1] and row-sum pmfs [block_len*check_blocks,
    2*sum_weight*eta+1], the latter REVERSED so that every row of H sums to 0."""
    assert len(s_distr) == num_blocks and len(s_distr[0]) == block_len
    ssum_len = block_len * check_blocks
    assert len(ssum_distr) == ssum_len
    B = sum_weight * eta
    channel_output = np.zeros((block_len * num_blocks, 2 * eta + 1), dtype=np.float32)
    channel_output_sum = np.zeros((ssum_len, 2 * B + 1), dtype=np.float32)
    for j in range(num_blocks):
        for i in range(block_len):
            channel_output[i + j * block_len] = s_distr[j][i]
    for i in range(ssum_len):
        channel_output_sum[i] = np.asarray(ssum_distr[i])[::-1]
    return channel_output, channel_output_sum

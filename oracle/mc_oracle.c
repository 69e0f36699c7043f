/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement of the device-side Monte-Carlo helpers (K6): what the reference does per
 * position in Python (simulate/decode.py:36-40,165-168: `rng.rand() < p`; simulate/hqc.py
 * 684-705: the hqc.decode() inputs), with the build's counter-based generator in place of
 * the reference's sequential MT19937 -- trials must be reproducible from (seed, global
 * trial index) alone so that sharding over GPUs cannot change them.
 *
 * Generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11), restated from the paper; pinned by the known-answer vectors of the
 * Random123 distribution (tests/test_mc.py).  key = (seed_lo, seed_hi),
 * counter = (block, stream, trial_lo, trial_hi).
 *   stream 0 word x : Bernoulli(p)  <=>  word < floor(p * 2^32)   (p >= 1 always, p <= 0 never)
 *   stream 1 word j : j-th candidate position, pos = (word * N) >> 32; duplicates skipped
 */
#include <stdint.h>
#include <stdlib.h>

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint32_t word(uint64_t seed, uint64_t trial, uint32_t stream, uint32_t idx)
{
    uint32_t ctr[4] = {idx >> 2, stream, (uint32_t)trial, (uint32_t)(trial >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, out[4];
    oracle_philox4x32_10(ctr, key, out);
    return out[idx & 3];
}

static uint64_t threshold(double p) { return p >= 1.0 ? (1ull << 32) : p <= 0.0 ? 0ull : (uint64_t)(p * 4294967296.0); }

/* out[b][i] = [stream-0 word i of trial first+b < thr(probs[i])]; probs == NULL -> p0 everywhere */
void oracle_mc_bernoulli(uint64_t seed, int64_t first, int batch, int len, const double *probs, double p0,
                         uint8_t *out)
{
    for (int b = 0; b < batch; b++)
        for (int i = 0; i < len; i++)
            out[(size_t)b * len + i] =
                (uint64_t)word(seed, (uint64_t)(first + b), 0, (uint32_t)i) < threshold(probs ? probs[i] : p0);
}

/* y[b][0..omega): distinct positions in [0, N), in draw order */
void oracle_mc_hqc_secret(uint64_t seed, int64_t first, int batch, int N, int omega, int32_t *y)
{
    for (int b = 0; b < batch; b++) {
        uint32_t j = 0;
        for (int i = 0; i < omega; i++) {
            for (;;) {
                uint32_t w = word(seed, (uint64_t)(first + b), 1, j++);
                int32_t pos = (int32_t)(((uint64_t)w * (uint32_t)N) >> 32);
                int dup = 0;
                for (int q = 0; q < i; q++) dup |= y[(size_t)b * omega + q] == pos;
                if (!dup) {
                    y[(size_t)b * omega + i] = pos;
                    break;
                }
            }
        }
    }
}

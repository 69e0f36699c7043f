// libscaldpc -- binary LDPC belief propagation on MI355X (gfx950), flooding schedule.
//
// Replaces the decode loop behind `ldpc.bp_decoder(H, ...).decode(v)` as the
// reference calls it (simulate-with-python/simulate/decode.py:155-161,171;
// simulate/hqc.py:694-699,708).  The arithmetic follows that package's sweep
// order: per check an exclusive forward/backward sweep
// over the row in ascending column, per variable an exclusive prefix/suffix
// SUM over the column in ascending row, hard decision `L <= 0 -> 1`, then
// H e == s.  Nothing is ever "total minus self", so +-inf priors (certainty-1.0
// checks, hqc.py:689) never meet an inf - inf.
//
// HBM layout (all fp32 messages; TW = 256 codewords form one tile):
//     v2c, c2v : float [tile][edge][256]     edge id = CSR position
//   A check node's edges are one contiguous range, so a wave working on
//   (row, tile) streams deg x 1 KiB of consecutive memory with 16 B per lane;
//   a variable node gathers its 1 KiB edge rows through the CSC permutation.
//   Bit planes (syndrome, received word, hard decision, done mask):
//     u64 [tile][node][4]    bit c of the 256-bit vector = codeword c of the tile
//   posterior : float [tile][var][256]
//
// No MFMA anywhere: this is a sparse gather/scatter stream bound by HBM.
#include "scaldpc_common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

typedef unsigned long long u64;

namespace scaldpc {
std::string &last_error()
{
    static thread_local std::string s;
    return s;
}
int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
}  // namespace scaldpc

using namespace scaldpc;

namespace {

constexpr int TW = 256;  // codewords per tile

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// bits of the low 32 -> even bit positions of a 64-bit word
__device__ __forceinline__ u64 spread32(u64 x)
{
    x &= 0xffffffffull;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// ---------------------------------------------------------------------------
// input / output reshaping
// ---------------------------------------------------------------------------
// uint8 [batch][len] (one row per codeword, as decode() receives them) ->
// bit planes [tile][x][4].  grid (len, T), block 256: thread c = codeword c of the tile.
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t *__restrict__ in, int len, int batch,
                                                   u64 *__restrict__ out)
{
    const int x = blockIdx.x, t = blockIdx.y, c = threadIdx.x;
    const long b = (long)t * TW + c;
    int bit = 0;
    if (b < batch) bit = in[(size_t)b * len + x] & 1;
    const u64 w = __ballot(bit);
    if ((c & 63) == 0) out[((size_t)t * len + x) * 4 + (c >> 6)] = w;
}

// hard decision planes (XOR received planes) -> uint8 [batch][n].
// grid (ceil(n/256), batch), block 256: consecutive threads = consecutive variables.
__global__ __launch_bounds__(256) void k_unpack_bits(const u64 *__restrict__ hard, const u64 *__restrict__ recv,
                                                     int n, uint8_t *__restrict__ out)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (v >= n) return;
    const int t = b >> 8, c = b & 255;
    const size_t i = ((size_t)t * n + v) * 4 + (c >> 6);
    u64 w = hard[i];
    if (recv) w ^= recv[i];
    out[(size_t)b * n + v] = (uint8_t)((w >> (c & 63)) & 1);
}

// posterior [tile][var][256] -> float [batch][n] via a 32x32 LDS transpose.
// grid (ceil(n/32), 8, T), block (32, 8).
__global__ void k_unpack_llr(const float *__restrict__ post, int n, int batch, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int v0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int v = v0 + j;
        if (v < n) tile[j][threadIdx.x] = post[((size_t)t * n + v) * TW + c0 + threadIdx.x];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const long b = (long)t * TW + c0 + j;
        const int v = v0 + threadIdx.x;
        if (b < batch && v < n) out[(size_t)b * n + v] = tile[threadIdx.x][j];
    }
}

// done planes + iteration counters -> int32 iters[batch], uint8 conv[batch]
__global__ __launch_bounds__(256) void k_unpack_state(const u64 *__restrict__ conv_bits, const int *__restrict__ iters,
                                                      int batch, int *__restrict__ out_iters,
                                                      uint8_t *__restrict__ out_conv)
{
    const int t = blockIdx.x, c = threadIdx.x;
    const long b = (long)t * TW + c;
    if (b >= batch) return;
    if (out_iters) out_iters[b] = iters[b];
    if (out_conv) out_conv[b] = (uint8_t)((conv_bits[(size_t)t * 4 + (c >> 6)] >> (c & 63)) & 1);
}

// ---------------------------------------------------------------------------
// parity of bit planes along the rows of H.  One thread per (row, tile).
//   CHECK = false: synd[t][r] = XOR_v bits[t][v]          (received-vector mode: s = H v)
//   CHECK = true : unsat[t] |= synd[t][r] ^ XOR_v bits    (convergence test H e == s)
// The planes of one tile are n x 32 B (693 KB at HQC-128): L2 resident.
// grid (ceil(m/256), T), block 256.
// ---------------------------------------------------------------------------
template <bool CHECK>
__global__ __launch_bounds__(256) void k_parity(const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                                const u64 *__restrict__ bits, int m, int n, u64 *__restrict__ synd,
                                                u64 *__restrict__ unsat)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (r < m) {
        const int e1 = row_ptr[r + 1];
        for (int e = row_ptr[r]; e < e1; e++) {
            const ulonglong2 *p = (const ulonglong2 *)(bits + ((size_t)t * n + col_idx[e]) * 4);
            const ulonglong2 lo = p[0], hi = p[1];
            a0 ^= lo.x; a1 ^= lo.y; a2 ^= hi.x; a3 ^= hi.y;
        }
        u64 *s = synd + ((size_t)t * m + r) * 4;
        if (CHECK) {
            a0 ^= s[0]; a1 ^= s[1]; a2 ^= s[2]; a3 ^= s[3];
        } else {
            s[0] = a0; s[1] = a1; s[2] = a2; s[3] = a3;
        }
    }
    if (CHECK) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a0 |= __shfl_xor(a0, off);
            a1 |= __shfl_xor(a1, off);
            a2 |= __shfl_xor(a2, off);
            a3 |= __shfl_xor(a3, off);
        }
        if ((threadIdx.x & 63) == 0) {
            u64 *u = unsat + (size_t)t * 4;
            if (a0) atomicOr(u + 0, a0);
            if (a1) atomicOr(u + 1, a1);
            if (a2) atomicOr(u + 2, a2);
            if (a3) atomicOr(u + 3, a3);
        }
    }
}

// per-tile state reset.  grid T, block 256.
__global__ __launch_bounds__(256) void k_init_state(int batch, int max_iter, u64 *__restrict__ done,
                                                    u64 *__restrict__ conv, u64 *__restrict__ unsat,
                                                    int *__restrict__ iters)
{
    const int t = blockIdx.x, c = threadIdx.x;
    const long b = (long)t * TW + c;
    iters[b] = max_iter;
    const u64 pad = __ballot(b >= batch);  // padding codewords are born "done"
    if ((c & 63) == 0) {
        done[(size_t)t * 4 + (c >> 6)] = pad;
        conv[(size_t)t * 4 + (c >> 6)] = 0;
        unsat[(size_t)t * 4 + (c >> 6)] = 0;
    }
}

// Latch convergence after the parity test of iteration `it`.  grid T, block 256.
//   latch = 1 (early exit): a codeword that satisfies H e == s for the first time is
//           frozen: done bit set, iters = it, its outputs are no longer overwritten.
//   latch = 0 (fixed iterations): only record whether the FINAL decision satisfies.
// remaining[slot] += number of codewords still running.
__global__ __launch_bounds__(256) void k_finalize(int it, int latch, u64 *__restrict__ done, u64 *__restrict__ conv,
                                                  u64 *__restrict__ unsat, int *__restrict__ iters,
                                                  int *__restrict__ remaining)
{
    const int t = blockIdx.x, c = threadIdx.x, w = c >> 6, l = c & 63;
    const u64 uw = unsat[(size_t)t * 4 + w];
    const u64 dw = done[(size_t)t * 4 + w];
    __syncthreads();
    const u64 newly = ~dw & ~uw;
    if (latch) {
        if ((newly >> l) & 1) iters[(long)t * TW + c] = it;
        if (l == 0) {
            done[(size_t)t * 4 + w] = dw | newly;
            conv[(size_t)t * 4 + w] |= newly;
            unsat[(size_t)t * 4 + w] = 0;
            const int rem = __popcll(~(dw | newly));
            if (rem) atomicAdd(remaining, rem);
        }
    } else if (l == 0) {
        conv[(size_t)t * 4 + w] = ~uw & ~dw;  // dw = padding here
        unsat[(size_t)t * 4 + w] = 0;
    }
}

// ---------------------------------------------------------------------------
// K1  initial bit-to-check messages: v2c[tile][e][:] = LLR prior of the edge's column.
// grid (ceil(E/4), G), block 256 = 4 waves, wave = one edge row (1 KiB store).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_v2c(const int *__restrict__ col_idx, const float *__restrict__ prior,
                                                  float4 *__restrict__ v2c, long E)
{
    const int lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float p = prior[col_idx[e]];
    v2c[((size_t)blockIdx.y * E + e) * 64 + lane] = make_float4(p, p, p, p);
}

// ---------------------------------------------------------------------------
// K3  min-sum check-node update.
//   c2v_k = alpha * (-1)^(s + #{k' != k : v2c_k' <= 0}) * min_{k' != k} |v2c_k'|
// The reference package obtains the exclusive minimum by a forward and a backward
// running min; min is exact, so (min1, min2, first argmin) gives the identical
// value with ONE pass over the inputs.  Signs of all inputs are kept in a 64-bit
// mask per codeword (rows of degree <= 64); WIDE rows re-read the input instead.
// wave = (row, tile): lane handles 4 codewords (16 B loads / stores).
// grid (ceil(m/4), G), block 256 = 4 rows.
// ---------------------------------------------------------------------------
#define MS_ACC(x, K, mn1, mn2, ix, par, ng)                      \
    {                                                            \
        const float a_ = fabsf(x);                               \
        const unsigned n_ = (x) <= 0.0f;                         \
        par ^= n_;                                               \
        if (!WIDE) ng |= (u64)n_ << (K);                         \
        const bool lt_ = a_ < mn1;                               \
        mn2 = lt_ ? mn1 : ((a_ < mn2) ? a_ : mn2);               \
        ix = lt_ ? (K) : ix;                                     \
        mn1 = lt_ ? a_ : mn1;                                    \
    }

template <bool WIDE>
__global__ __launch_bounds__(256) void k_check_minsum(const int *__restrict__ row_ptr, const float4 *__restrict__ v2c,
                                                      float4 *__restrict__ c2v, const u64 *__restrict__ synd, int m,
                                                      long E, float alpha)
{
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= m) return;
    r = rfl(r);
    const int tl = blockIdx.y;
    const int e0 = rfl(row_ptr[r]);
    const int deg = rfl(row_ptr[r + 1]) - e0;
    const size_t base = ((size_t)tl * E + e0) * 64 + lane;
    const float4 *__restrict__ in = v2c + base;
    float4 *__restrict__ out = c2v + base;
    // syndrome bits of codewords 4*lane .. 4*lane+3 (natural plane layout)
    const u64 sw = synd[((size_t)tl * m + r) * 4 + (lane >> 4)];
    const unsigned sb = (unsigned)(sw >> ((lane & 15) * 4)) & 0xFu;

    float m1x = FLT_MAX, m1y = FLT_MAX, m1z = FLT_MAX, m1w = FLT_MAX;
    float m2x = FLT_MAX, m2y = FLT_MAX, m2z = FLT_MAX, m2w = FLT_MAX;
    int ix = 0, iy = 0, iz = 0, iw = 0;
    unsigned px = sb & 1, py = (sb >> 1) & 1, pz = (sb >> 2) & 1, pw = (sb >> 3) & 1;
    u64 nx = 0, ny = 0, nz = 0, nw = 0;

#pragma unroll 4
    for (int k = 0; k < deg; k++) {
        const float4 x = in[(size_t)k * 64];
        MS_ACC(x.x, k, m1x, m2x, ix, px, nx)
        MS_ACC(x.y, k, m1y, m2y, iy, py, ny)
        MS_ACC(x.z, k, m1z, m2z, iz, pz, nz)
        MS_ACC(x.w, k, m1w, m2w, iw, pw, nw)
    }
    const float nalpha = -alpha;
#pragma unroll 4
    for (int k = 0; k < deg; k++) {
        unsigned bx, by, bz, bw;
        if (WIDE) {
            const float4 x = in[(size_t)k * 64];
            bx = x.x <= 0.0f; by = x.y <= 0.0f; bz = x.z <= 0.0f; bw = x.w <= 0.0f;
        } else {
            bx = (unsigned)(nx >> k) & 1; by = (unsigned)(ny >> k) & 1;
            bz = (unsigned)(nz >> k) & 1; bw = (unsigned)(nw >> k) & 1;
        }
        float4 o;
        o.x = ((k == ix) ? m2x : m1x) * ((px ^ bx) ? nalpha : alpha);
        o.y = ((k == iy) ? m2y : m1y) * ((py ^ by) ? nalpha : alpha);
        o.z = ((k == iz) ? m2z : m1z) * ((pz ^ bz) ? nalpha : alpha);
        o.w = ((k == iw) ? m2w : m1w) * ((pw ^ bw) ? nalpha : alpha);
        out[(size_t)k * 64] = o;
    }
}

// ---------------------------------------------------------------------------
// K2  tanh-rule (sum-product) check-node update, LLR domain, fp32, COMPLEMENT form.
//   c2v_k = (-1)^(s + #{k' != k : x_k' < 0}) * 2 atanh( prod_{k' != k} tanh(|x_k'|/2) )
// computed without the 1-x cancellation that saturates the textbook form at |L|~17
// in fp32:   u_k = 1 - tanh(|x_k|/2) = 2 / (exp|x_k| + 1)
//            U   = 1 - prod(1 - u)   via  U' = U + u (1 - U)     (forward and backward)
//            |c2v_k| = log(2 / U_excl - 1),  U_excl = Upre + Usuf (1 - Upre)
// Exact for |L| up to ~88 (then u underflows to 0 and L = +inf, which is also what
// p = 0 priors feed in).  Same exclusive forward/backward sweep as the reference
// package; the CPU oracle's method 3 is this sequence op for op.
// Row values live in registers (compile-time unrolled to MAXDEG <= 64, predicated on
// the wave-uniform degree), signs in a 64-bit mask.  wave = (row, quarter tile).
// grid (rows in bucket, G), block 256 = the 4 quarters of one (row, tile).
// ---------------------------------------------------------------------------
template <int MAXDEG>
__global__ __launch_bounds__(256) void k_check_tanh(const int *__restrict__ list, const int *__restrict__ row_ptr,
                                                    const float *__restrict__ v2c, float *__restrict__ c2v,
                                                    const u64 *__restrict__ synd, int m, long E)
{
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = rfl(list[blockIdx.x]);
    const int tl = blockIdx.y;
    const int e0 = rfl(row_ptr[r]);
    const int deg = rfl(row_ptr[r + 1]) - e0;
    const size_t base = ((size_t)tl * E + e0) * TW + q * 64 + lane;
    float uu[MAXDEG], pre[MAXDEG];
#pragma unroll
    for (int k = 0; k < MAXDEG; k++)
        if (k < deg) uu[k] = v2c[base + (size_t)k * TW];
    u64 neg = 0;
    float U = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXDEG; k++)
        if (k < deg) {
            const float x = uu[k];
            neg |= (u64)(x < 0.0f) << k;
            const float u = 2.0f / (expf(fabsf(x)) + 1.0f);
            uu[k] = u;
            pre[k] = U;
            U = U + u * (1.0f - U);
        }
    const unsigned tot = ((unsigned)__popcll(neg) ^ (unsigned)(synd[((size_t)tl * m + r) * 4 + q] >> lane)) & 1u;
    U = 0.0f;
#pragma unroll
    for (int k = MAXDEG - 1; k >= 0; k--)
        if (k < deg) {
            const float Ut = pre[k] + U * (1.0f - pre[k]);
            const float Lm = logf(2.0f / Ut - 1.0f);
            c2v[base + (size_t)k * TW] = ((tot ^ (unsigned)(neg >> k)) & 1u) ? -Lm : Lm;
            U = U + uu[k] * (1.0f - U);
        }
}

// Any-degree fallback: the forward sweep parks Upre in c2v itself (as the reference
// package parks its prefix products), the backward sweep re-reads v2c and recomputes u.
// grid (rows in list, G), block 256.
__global__ __launch_bounds__(256) void k_check_tanh_generic(const int *__restrict__ list,
                                                            const int *__restrict__ row_ptr,
                                                            const float *__restrict__ v2c, float *c2v,
                                                            const u64 *__restrict__ synd, int m, long E)
{
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = rfl(list[blockIdx.x]);
    const int tl = blockIdx.y;
    const int e0 = rfl(row_ptr[r]);
    const int deg = rfl(row_ptr[r + 1]) - e0;
    const size_t base = ((size_t)tl * E + e0) * TW + q * 64 + lane;
    float U = 0.0f;
    unsigned par = (unsigned)(synd[((size_t)tl * m + r) * 4 + q] >> lane) & 1u;
    for (int k = 0; k < deg; k++) {
        const float x = v2c[base + (size_t)k * TW];
        c2v[base + (size_t)k * TW] = U;
        par ^= (unsigned)(x < 0.0f);
        const float u = 2.0f / (expf(fabsf(x)) + 1.0f);
        U = U + u * (1.0f - U);
    }
    U = 0.0f;
    for (int k = deg - 1; k >= 0; k--) {
        const float x = v2c[base + (size_t)k * TW];
        const float p = c2v[base + (size_t)k * TW];
        const float Ut = p + U * (1.0f - p);
        const float Lm = logf(2.0f / Ut - 1.0f);
        c2v[base + (size_t)k * TW] = ((par ^ (unsigned)(x < 0.0f)) & 1u) ? -Lm : Lm;
        const float u = 2.0f / (expf(fabsf(x)) + 1.0f);
        U = U + u * (1.0f - U);
    }
}

// ---------------------------------------------------------------------------
// K4  variable-node update + posterior + hard decision.
//   prefix : v2c_k = prior + sum_{k'<k} c2v_k'      (ascending row)
//   total  : L = prior + sum_k c2v_k ; e = [L <= 0]
//   suffix : v2c_k += sum_{k'>k} c2v_k'             (accumulated from the last edge)
// Column values live in registers (unrolled to MAXD, predicated on the uniform
// degree); columns are bucketed by degree on the host.  wave = (column, half
// tile): lane handles 2 codewords (8 B), the two halves of a 1 KiB edge row are
// fetched by adjacent waves of one block.
// grid (ceil(count/2), G), block 256 = 2 columns x 2 halves.
// write_out: also emit hard-decision planes (merged under the done mask) and, if
// `post` is non-null, the posterior of every not-yet-frozen codeword.
// ---------------------------------------------------------------------------
template <int MAXD>
__global__ __launch_bounds__(256) void k_var(const int *__restrict__ list, int count, const int *__restrict__ col_ptr,
                                             const int *__restrict__ csc_edge, const float *__restrict__ prior,
                                             const float *__restrict__ c2v, float *__restrict__ v2c,
                                             float *__restrict__ post, u64 *__restrict__ hard,
                                             const u64 *__restrict__ done, int n, long E, int write_out)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 2 + (w >> 1);
    if (slot >= count) return;
    const int half = w & 1;
    const int tl = blockIdx.y;
    const int v = rfl(list[slot]);
    const int cb = rfl(col_ptr[v]);
    const int d = rfl(col_ptr[v + 1]) - cb;
    const size_t lane_off = (size_t)half * 128 + lane * 2;
    const size_t tile_off = (size_t)tl * E;

    float2 mm[MAXD], pp[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; k++)
        if (k < d) {
            const int e = rfl(csc_edge[cb + k]);
            mm[k] = *(const float2 *)(c2v + (tile_off + e) * TW + lane_off);
        }
    const float pr = prior[v];
    float2 temp = make_float2(pr, pr);
#pragma unroll
    for (int k = 0; k < MAXD; k++)
        if (k < d) {
            pp[k] = temp;
            temp.x += mm[k].x;
            temp.y += mm[k].y;
        }
    float2 suf = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = MAXD - 1; k >= 0; k--)
        if (k < d) {
            const int e = rfl(csc_edge[cb + k]);
            float2 o;
            o.x = pp[k].x + suf.x;
            o.y = pp[k].y + suf.y;
            *(float2 *)(v2c + (tile_off + e) * TW + lane_off) = o;
            suf.x += mm[k].x;
            suf.y += mm[k].y;
        }
    if (write_out) {
        // codewords c = 128*half + 2*lane + i  ->  plane word 2*half + (lane>>5), bit 2*(lane&31)+i
        const u64 b0 = __ballot(temp.x <= 0.0f), b1 = __ballot(temp.y <= 0.0f);
        const u64 w0 = spread32(b0) | (spread32(b1) << 1);
        const u64 w1 = spread32(b0 >> 32) | (spread32(b1 >> 32) << 1);
        const size_t hi = ((size_t)tl * n + v) * 4 + 2 * half;
        const u64 d0 = done[(size_t)tl * 4 + 2 * half], d1 = done[(size_t)tl * 4 + 2 * half + 1];
        if (lane == 0) {
            hard[hi] = (hard[hi] & d0) | (w0 & ~d0);
            hard[hi + 1] = (hard[hi + 1] & d1) | (w1 & ~d1);
        }
        if (post) {
            const u64 dw = (lane >> 5) ? d1 : d0;
            const unsigned db = (unsigned)(dw >> (2 * (lane & 31))) & 3u;
            float *p = post + ((size_t)tl * n + v) * TW + lane_off;
            if (db == 0)
                *(float2 *)p = temp;
            else {
                if (!(db & 1)) p[0] = temp.x;
                if (!(db & 2)) p[1] = temp.y;
            }
        }
    }
}

// Any-degree fallback (columns wider than the largest bucket): prefix parked in
// v2c, second sweep re-reads c2v.  Same launch geometry as k_var.
__global__ __launch_bounds__(256) void k_var_generic(const int *__restrict__ list, int count,
                                                     const int *__restrict__ col_ptr, const int *__restrict__ csc_edge,
                                                     const float *__restrict__ prior, const float *__restrict__ c2v,
                                                     float *v2c, float *__restrict__ post, u64 *__restrict__ hard,
                                                     const u64 *__restrict__ done, int n, long E, int write_out)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 2 + (w >> 1);
    if (slot >= count) return;
    const int half = w & 1;
    const int tl = blockIdx.y;
    const int v = rfl(list[slot]);
    const int cb = rfl(col_ptr[v]);
    const int d = rfl(col_ptr[v + 1]) - cb;
    const size_t lane_off = (size_t)half * 128 + lane * 2;
    const size_t tile_off = (size_t)tl * E;
    const float pr = prior[v];
    float2 temp = make_float2(pr, pr);
    for (int k = 0; k < d; k++) {
        const int e = csc_edge[cb + k];
        const float2 mk = *(const float2 *)(c2v + (tile_off + e) * TW + lane_off);
        *(float2 *)(v2c + (tile_off + e) * TW + lane_off) = temp;
        temp.x += mk.x;
        temp.y += mk.y;
    }
    float2 suf = make_float2(0.0f, 0.0f);
    for (int k = d - 1; k >= 0; k--) {
        const int e = csc_edge[cb + k];
        const float2 mk = *(const float2 *)(c2v + (tile_off + e) * TW + lane_off);
        float2 o = *(float2 *)(v2c + (tile_off + e) * TW + lane_off);
        o.x += suf.x;
        o.y += suf.y;
        *(float2 *)(v2c + (tile_off + e) * TW + lane_off) = o;
        suf.x += mk.x;
        suf.y += mk.y;
    }
    if (write_out) {
        const u64 b0 = __ballot(temp.x <= 0.0f), b1 = __ballot(temp.y <= 0.0f);
        const u64 w0 = spread32(b0) | (spread32(b1) << 1);
        const u64 w1 = spread32(b0 >> 32) | (spread32(b1 >> 32) << 1);
        const size_t hi = ((size_t)tl * n + v) * 4 + 2 * half;
        const u64 d0 = done[(size_t)tl * 4 + 2 * half], d1 = done[(size_t)tl * 4 + 2 * half + 1];
        if (lane == 0) {
            hard[hi] = (hard[hi] & d0) | (w0 & ~d0);
            hard[hi + 1] = (hard[hi + 1] & d1) | (w1 & ~d1);
        }
        if (post) {
            const u64 dw = (lane >> 5) ? d1 : d0;
            const unsigned db = (unsigned)(dw >> (2 * (lane & 31))) & 3u;
            float *p = post + ((size_t)tl * n + v) * TW + lane_off;
            if (!(db & 1)) p[0] = temp.x;
            if (!(db & 2)) p[1] = temp.y;
        }
    }
}

struct Bucket {
    int maxd;   // unroll bound (0 = generic kernel)
    int off;    // offset into the list
    int count;  // nodes in the bucket
};

}  // namespace

// ===========================================================================
// handle
// ===========================================================================
struct scaldpc_bp {
    int m = 0, n = 0;
    long E = 0;
    int max_row_deg = 0;
    // device graph
    int *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr;
    int *d_var_list = nullptr, *d_row_list = nullptr;
    std::vector<Bucket> var_buckets, row_buckets;
    // priors
    float *d_prior = nullptr;
    bool have_prior = false;
    // workspace (state arrays sized for cap_tiles, messages for cap_group tiles)
    int cap_tiles = 0, cap_group = 0, tile_group = 0;
    float *d_v2c = nullptr, *d_c2v = nullptr, *d_post = nullptr;
    u64 *d_synd = nullptr, *d_recv = nullptr, *d_hard = nullptr, *d_done = nullptr, *d_conv = nullptr,
        *d_unsat = nullptr;
    int *d_iters = nullptr, *d_remaining = nullptr;
    int cap_remaining = 0;
    bool post_alloc = false;
    // host-I/O staging
    uint8_t *d_in = nullptr, *d_out_bits = nullptr, *d_out_conv = nullptr;
    float *d_out_llr = nullptr;
    int *d_out_iters = nullptr;
    size_t cap_in = 0, cap_out_bits = 0, cap_out_llr = 0, cap_out_b = 0;
    int *h_remaining = nullptr;  // pinned
    hipStream_t own_stream = nullptr;
    // last decode geometry (for scaldpc_bp_time_kernels)
    int last_group = 0;
    std::mutex mu;
};

namespace {

int build_buckets(const std::vector<int> &deg, const int *bounds, int nb, bool keep_isolated,
                  std::vector<int> &list, std::vector<Bucket> &out)
{
    // bucket b holds nodes with bounds[b-1] < deg <= bounds[b]; beyond the last bound -> generic (maxd 0)
    std::vector<std::vector<int>> tmp(nb + 1);
    for (int i = 0; i < (int)deg.size(); i++) {
        // an isolated check has nothing to send; an isolated VARIABLE still owes its
        // posterior (= prior) and hard decision, so it stays in the smallest bucket
        if (deg[i] == 0 && !keep_isolated) continue;
        int b = 0;
        while (b < nb && deg[i] > bounds[b]) b++;
        tmp[b].push_back(i);
    }
    list.clear();
    out.clear();
    for (int b = 0; b <= nb; b++) {
        if (tmp[b].empty()) continue;
        Bucket bk;
        bk.maxd = b < nb ? bounds[b] : 0;
        bk.off = (int)list.size();
        bk.count = (int)tmp[b].size();
        list.insert(list.end(), tmp[b].begin(), tmp[b].end());
        out.push_back(bk);
    }
    return 0;
}

template <typename T>
int grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return 0;
    dev_free(*p);
    *cap = 0;
    SC_TRY(dev_alloc(p, need));
    *cap = need;
    return 0;
}

int ensure_workspace(scaldpc_bp *h, int T, int G, bool want_post, int max_iter)
{
    if (T > h->cap_tiles) {
        dev_free(h->d_synd); dev_free(h->d_recv); dev_free(h->d_hard); dev_free(h->d_done);
        dev_free(h->d_conv); dev_free(h->d_unsat); dev_free(h->d_iters); dev_free(h->d_post);
        h->cap_tiles = 0;
        h->post_alloc = false;
        SC_TRY(dev_alloc(&h->d_synd, (size_t)T * h->m * 4));
        SC_TRY(dev_alloc(&h->d_recv, (size_t)T * h->n * 4));
        SC_TRY(dev_alloc(&h->d_hard, (size_t)T * h->n * 4));
        SC_TRY(dev_alloc(&h->d_done, (size_t)T * 4));
        SC_TRY(dev_alloc(&h->d_conv, (size_t)T * 4));
        SC_TRY(dev_alloc(&h->d_unsat, (size_t)T * 4));
        SC_TRY(dev_alloc(&h->d_iters, (size_t)T * TW));
        h->cap_tiles = T;
    }
    if (want_post && !h->post_alloc) {
        SC_TRY(dev_alloc(&h->d_post, (size_t)h->cap_tiles * h->n * TW));
        h->post_alloc = true;
    }
    if (G > h->cap_group) {
        dev_free(h->d_v2c); dev_free(h->d_c2v);
        h->cap_group = 0;
        SC_TRY(dev_alloc(&h->d_v2c, (size_t)G * h->E * TW));
        SC_TRY(dev_alloc(&h->d_c2v, (size_t)G * h->E * TW));
        h->cap_group = G;
    }
    if (max_iter + 2 > h->cap_remaining) {
        dev_free(h->d_remaining);
        if (h->h_remaining) (void)hipHostFree(h->h_remaining);
        h->h_remaining = nullptr;
        h->cap_remaining = 0;
        SC_TRY(dev_alloc(&h->d_remaining, (size_t)max_iter + 2));
        SC_HIP(hipHostMalloc((void **)&h->h_remaining, sizeof(int) * ((size_t)max_iter + 2), hipHostMallocDefault));
        h->cap_remaining = max_iter + 2;
    }
    return 0;
}

#define LAUNCH_CHECK() SC_HIP(hipGetLastError())

int launch_check(scaldpc_bp *h, int method, float alpha, int G, const u64 *synd_g, hipStream_t s)
{
    if (method == SCALDPC_BP_MIN_SUM) {
        dim3 grid((h->m + 3) / 4, G);
        if (h->max_row_deg <= 64)
            hipLaunchKernelGGL(k_check_minsum<false>, grid, dim3(256), 0, s, h->d_row_ptr, (const float4 *)h->d_v2c,
                               (float4 *)h->d_c2v, synd_g, h->m, h->E, alpha);
        else
            hipLaunchKernelGGL(k_check_minsum<true>, grid, dim3(256), 0, s, h->d_row_ptr, (const float4 *)h->d_v2c,
                               (float4 *)h->d_c2v, synd_g, h->m, h->E, alpha);
        LAUNCH_CHECK();
        return 0;
    }
    for (const Bucket &b : h->row_buckets) {
        dim3 grid(b.count, G);
        const int *list = h->d_row_list + b.off;
#define TANH_CASE(D)                                                                                             \
    case D:                                                                                                      \
        hipLaunchKernelGGL(k_check_tanh<D>, grid, dim3(256), 0, s, list, h->d_row_ptr, h->d_v2c, h->d_c2v, synd_g, \
                           h->m, h->E);                                                                          \
        break;
        switch (b.maxd) {
            TANH_CASE(2)
            TANH_CASE(4)
            TANH_CASE(8)
            TANH_CASE(16)
            TANH_CASE(32)
            TANH_CASE(64)
            default:
                hipLaunchKernelGGL(k_check_tanh_generic, grid, dim3(256), 0, s, list, h->d_row_ptr, h->d_v2c, h->d_c2v,
                                   synd_g, h->m, h->E);
        }
#undef TANH_CASE
        LAUNCH_CHECK();
    }
    return 0;
}

int launch_var(scaldpc_bp *h, int G, float *post_g, u64 *hard_g, const u64 *done_g, int write_out, hipStream_t s)
{
    for (const Bucket &b : h->var_buckets) {
        dim3 grid((b.count + 1) / 2, G);
        const int *list = h->d_var_list + b.off;
#define VAR_CASE(D)                                                                                               \
    case D:                                                                                                       \
        hipLaunchKernelGGL(k_var<D>, grid, dim3(256), 0, s, list, b.count, h->d_col_ptr, h->d_csc_edge, h->d_prior, \
                           h->d_c2v, h->d_v2c, post_g, hard_g, done_g, h->n, h->E, write_out);                     \
        break;
        switch (b.maxd) {
            VAR_CASE(1)
            VAR_CASE(2)
            VAR_CASE(4)
            VAR_CASE(8)
            VAR_CASE(16)
            VAR_CASE(32)
            default:
                hipLaunchKernelGGL(k_var_generic, grid, dim3(256), 0, s, list, b.count, h->d_col_ptr, h->d_csc_edge,
                                   h->d_prior, h->d_c2v, h->d_v2c, post_g, hard_g, done_g, h->n, h->E, write_out);
        }
#undef VAR_CASE
        LAUNCH_CHECK();
    }
    return 0;
}

float alpha_for(float alpha, int it)
{
    // ms_scaling_factor == 0 -> 1 - 2^-iter
    return alpha == 0.0f ? (float)(1.0 - std::pow(2.0, -1.0 * it)) : alpha;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *scaldpc_last_error(void) { return last_error().c_str(); }
int scaldpc_version(void) { return SCALDPC_VERSION; }

int scaldpc_device_count(int *count)
{
    if (!count) return fail(SCALDPC_EINVAL, "count is NULL");
    SC_HIP(hipGetDeviceCount(count));
    return 0;
}

int scaldpc_set_device(int device)
{
    SC_HIP(hipSetDevice(device));
    return 0;
}

int scaldpc_bp_create(int32_t m, int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx,
                      scaldpc_bp **out)
{
    if (!out) return fail(SCALDPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (m <= 0 || n <= 0 || nnz < 0 || !row_ptr || (nnz && !col_idx))
        return fail(SCALDPC_EINVAL, "bad graph arguments (m=%d n=%d nnz=%lld)", m, n, (long long)nnz);
    if (nnz > 0x7fffffffLL) return fail(SCALDPC_EINVAL, "nnz too large");
    if (row_ptr[0] != 0 || row_ptr[m] != nnz) return fail(SCALDPC_EINVAL, "row_ptr does not span [0, nnz]");
    std::vector<int> rdeg(m), cdeg(n, 0);
    for (int r = 0; r < m; r++) {
        if (row_ptr[r + 1] < row_ptr[r]) return fail(SCALDPC_EINVAL, "row_ptr not monotone at row %d", r);
        rdeg[r] = row_ptr[r + 1] - row_ptr[r];
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return fail(SCALDPC_EINVAL, "col_idx out of range at edge %d", e);
            if (e > row_ptr[r] && col_idx[e] <= col_idx[e - 1])
                return fail(SCALDPC_EINVAL, "col_idx not strictly ascending in row %d", r);
            cdeg[col_idx[e]]++;
        }
    }
    // CSC permutation: edges grouped by column, ascending row (row-major scan keeps rows ascending)
    std::vector<int> col_ptr(n + 1, 0), csc_edge((size_t)nnz), fill(n, 0);
    for (int j = 0; j < n; j++) col_ptr[j + 1] = col_ptr[j] + cdeg[j];
    for (int r = 0; r < m; r++)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            int j = col_idx[e];
            csc_edge[(size_t)col_ptr[j] + fill[j]++] = e;
        }

    scaldpc_bp *h = new (std::nothrow) scaldpc_bp();
    if (!h) return fail(SCALDPC_ENOMEM, "out of host memory");
    h->m = m;
    h->n = n;
    h->E = nnz;
    h->max_row_deg = m ? *std::max_element(rdeg.begin(), rdeg.end()) : 0;

    static const int vb[] = {1, 2, 4, 8, 16, 32};
    static const int rb[] = {2, 4, 8, 16, 32, 64};
    std::vector<int> vlist, rlist;
    build_buckets(cdeg, vb, 6, true, vlist, h->var_buckets);
    build_buckets(rdeg, rb, 6, false, rlist, h->row_buckets);

    int rc = 0;
    auto up = [&](int **d, const int *src, size_t cnt) -> int {
        SC_TRY(dev_alloc(d, cnt));
        if (cnt) SC_HIP(hipMemcpy(*d, src, cnt * sizeof(int), hipMemcpyHostToDevice));
        return 0;
    };
    if (!rc) rc = up(&h->d_row_ptr, row_ptr, (size_t)m + 1);
    if (!rc) rc = up(&h->d_col_idx, col_idx, (size_t)nnz);
    if (!rc) rc = up(&h->d_col_ptr, col_ptr.data(), (size_t)n + 1);
    if (!rc) rc = up(&h->d_csc_edge, csc_edge.data(), (size_t)nnz);
    if (!rc) rc = up(&h->d_var_list, vlist.data(), vlist.size());
    if (!rc) rc = up(&h->d_row_list, rlist.data(), rlist.size());
    if (!rc) rc = dev_alloc(&h->d_prior, (size_t)n);
    if (!rc && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(SCALDPC_EHIP, "hipStreamCreate failed");
    if (rc) {
        scaldpc_bp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int scaldpc_bp_set_channel_probs(scaldpc_bp *h, const double *probs)
{
    if (!h || !probs) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<float> llr(h->n);
    for (int j = 0; j < h->n; j++) {
        if (!(probs[j] >= 0.0 && probs[j] <= 1.0))
            return fail(SCALDPC_EINVAL, "channel_probs[%d] = %g is not a probability", j, probs[j]);
        // same expression, in fp32, as the oracle's f32 instantiation: log((1-p)/p)
        const float p = (float)probs[j];
        llr[j] = logf((1.0f - p) / p);
    }
    SC_HIP(hipMemcpy(h->d_prior, llr.data(), sizeof(float) * h->n, hipMemcpyHostToDevice));
    h->have_prior = true;
    return 0;
}

int scaldpc_bp_set_tile_group(scaldpc_bp *h, int32_t tiles)
{
    if (!h || tiles < 0) return fail(SCALDPC_EINVAL, "bad tile group");
    std::lock_guard<std::mutex> lk(h->mu);
    h->tile_group = tiles;
    return 0;
}

int scaldpc_bp_decode_batch(scaldpc_bp *h, const uint8_t *in, int32_t input_kind, int32_t batch, int32_t max_iter,
                            int32_t method, float alpha, uint32_t flags, void *stream, uint8_t *out_bits,
                            float *out_llr, int32_t *out_iters, uint8_t *out_conv)
{
    if (!h || !in || !out_bits) return fail(SCALDPC_EINVAL, "NULL argument");
    if (batch <= 0) return fail(SCALDPC_EINVAL, "batch must be positive (got %d)", batch);
    if (input_kind != SCALDPC_IN_SYNDROME && input_kind != SCALDPC_IN_RECEIVED)
        return fail(SCALDPC_EINVAL, "unknown input kind %d", input_kind);
    if (method != SCALDPC_BP_PRODUCT_SUM && method != SCALDPC_BP_MIN_SUM)
        return fail(SCALDPC_EINVAL, "unknown bp method %d", method);
    if (!(alpha >= 0.0f)) return fail(SCALDPC_EINVAL, "ms_scaling_factor must be >= 0");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->have_prior) return fail(SCALDPC_EINVAL, "channel probabilities not set");
    if (max_iter <= 0) max_iter = h->n;
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO;
    const bool early = flags & SCALDPC_F_EARLY_EXIT;
    if ((flags & SCALDPC_F_ASYNC) && (!dev_io || early))
        return fail(SCALDPC_EINVAL, "SCALDPC_F_ASYNC needs DEVICE_IO and no EARLY_EXIT (early exit polls the device)");
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;

    const int T = (batch + TW - 1) / TW;
    const int G = (h->tile_group > 0 && h->tile_group < T) ? h->tile_group : T;
    const int len = input_kind == SCALDPC_IN_SYNDROME ? h->m : h->n;
    SC_TRY(ensure_workspace(h, T, G, out_llr != nullptr, max_iter));

    // ---- stage input --------------------------------------------------------
    const uint8_t *din = in;
    if (!dev_io) {
        SC_TRY(grow(&h->d_in, &h->cap_in, (size_t)batch * len));
        SC_HIP(hipMemcpyAsync(h->d_in, in, (size_t)batch * len, hipMemcpyHostToDevice, s));
        din = h->d_in;
    }
    if (input_kind == SCALDPC_IN_SYNDROME) {
        hipLaunchKernelGGL(k_pack_bits, dim3(h->m, T), dim3(256), 0, s, din, h->m, batch, h->d_synd);
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_pack_bits, dim3(h->n, T), dim3(256), 0, s, din, h->n, batch, h->d_recv);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_parity<false>, dim3((h->m + 255) / 256, T), dim3(256), 0, s, h->d_row_ptr, h->d_col_idx,
                           h->d_recv, h->m, h->n, h->d_synd, (u64 *)nullptr);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_init_state, dim3(T), dim3(256), 0, s, batch, max_iter, h->d_done, h->d_conv, h->d_unsat,
                       h->d_iters);
    LAUNCH_CHECK();
    SC_HIP(hipMemsetAsync(h->d_hard, 0, sizeof(u64) * (size_t)T * h->n * 4, s));

    // ---- iterate, one tile group at a time -----------------------------------
    const int poll_every = 4;
    for (int g0 = 0; g0 < T; g0 += G) {
        const int g = std::min(G, T - g0);
        const u64 *synd_g = h->d_synd + (size_t)g0 * h->m * 4;
        u64 *hard_g = h->d_hard + (size_t)g0 * h->n * 4;
        u64 *done_g = h->d_done + (size_t)g0 * 4;
        u64 *conv_g = h->d_conv + (size_t)g0 * 4;
        u64 *unsat_g = h->d_unsat + (size_t)g0 * 4;
        int *iters_g = h->d_iters + (size_t)g0 * TW;
        float *post_g = out_llr ? h->d_post + (size_t)g0 * h->n * TW : nullptr;
        if (h->E) {
            hipLaunchKernelGGL(k_init_v2c, dim3((unsigned)((h->E + 3) / 4), g), dim3(256), 0, s, h->d_col_idx,
                               h->d_prior, (float4 *)h->d_v2c, h->E);
            LAUNCH_CHECK();
        }
        if (early) SC_HIP(hipMemsetAsync(h->d_remaining, 0, sizeof(int) * ((size_t)max_iter + 2), s));
        for (int it = 1; it <= max_iter; it++) {
            const bool last = it == max_iter;
            SC_TRY(launch_check(h, method, alpha_for(alpha, it), g, synd_g, s));
            SC_TRY(launch_var(h, g, post_g, hard_g, done_g, (early || last) ? 1 : 0, s));
            if (early || last) {
                hipLaunchKernelGGL(k_parity<true>, dim3((h->m + 255) / 256, g), dim3(256), 0, s, h->d_row_ptr,
                                   h->d_col_idx, hard_g, h->m, h->n, const_cast<u64 *>(synd_g), unsat_g);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(k_finalize, dim3(g), dim3(256), 0, s, it, early ? 1 : 0, done_g, conv_g, unsat_g,
                                   iters_g, h->d_remaining + it);
                LAUNCH_CHECK();
            }
            if (early && !last && (it % poll_every == 0 || it == 1)) {
                SC_HIP(hipMemcpyAsync(h->h_remaining + it, h->d_remaining + it, sizeof(int), hipMemcpyDeviceToHost, s));
                SC_HIP(hipStreamSynchronize(s));
                if (h->h_remaining[it] == 0) break;
            }
        }
    }
    h->last_group = std::min(G, T);

    // ---- outputs --------------------------------------------------------------
    uint8_t *dbits = out_bits;
    float *dllr = out_llr;
    int *diters = out_iters;
    uint8_t *dconv = out_conv;
    if (!dev_io) {
        SC_TRY(grow(&h->d_out_bits, &h->cap_out_bits, (size_t)batch * h->n));
        dbits = h->d_out_bits;
        if (out_llr) {
            SC_TRY(grow(&h->d_out_llr, &h->cap_out_llr, (size_t)batch * h->n));
            dllr = h->d_out_llr;
        }
        if ((size_t)batch > h->cap_out_b) {
            dev_free(h->d_out_iters);
            dev_free(h->d_out_conv);
            h->cap_out_b = 0;
            SC_TRY(dev_alloc(&h->d_out_iters, (size_t)batch));
            SC_TRY(dev_alloc(&h->d_out_conv, (size_t)batch));
            h->cap_out_b = batch;
        }
        diters = out_iters ? h->d_out_iters : nullptr;
        dconv = out_conv ? h->d_out_conv : nullptr;
    }
    hipLaunchKernelGGL(k_unpack_bits, dim3((h->n + 255) / 256, batch), dim3(256), 0, s, h->d_hard,
                       input_kind == SCALDPC_IN_RECEIVED ? h->d_recv : (const u64 *)nullptr, h->n, dbits);
    LAUNCH_CHECK();
    if (out_llr) {
        hipLaunchKernelGGL(k_unpack_llr, dim3((h->n + 31) / 32, TW / 32, T), dim3(32, 8), 0, s, h->d_post, h->n, batch,
                           dllr);
        LAUNCH_CHECK();
    }
    if (diters || dconv) {
        hipLaunchKernelGGL(k_unpack_state, dim3(T), dim3(256), 0, s, h->d_conv, h->d_iters, batch, diters, dconv);
        LAUNCH_CHECK();
    }
    if (!dev_io) {
        SC_HIP(hipMemcpyAsync(out_bits, dbits, (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
        if (out_llr) SC_HIP(hipMemcpyAsync(out_llr, dllr, sizeof(float) * (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
        if (out_iters) SC_HIP(hipMemcpyAsync(out_iters, diters, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, s));
        if (out_conv) SC_HIP(hipMemcpyAsync(out_conv, dconv, (size_t)batch, hipMemcpyDeviceToHost, s));
    }
    if (!(flags & SCALDPC_F_ASYNC)) SC_HIP(hipStreamSynchronize(s));
    return 0;
}

int scaldpc_bp_time_kernels(scaldpc_bp *h, int32_t iters, int32_t method, float alpha, void *stream, float *ms,
                            int32_t *launches)
{
    if (!h || !ms || !launches || iters <= 0) return fail(SCALDPC_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->last_group <= 0) return fail(SCALDPC_EINVAL, "no previous decode to time");
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    const int g = h->last_group;
    std::vector<hipEvent_t> ev((size_t)iters * 3);
    for (auto &e : ev) SC_HIP(hipEventCreate(&e));
    int rc = 0;
    for (int it = 0; it < iters && !rc; it++) {
        SC_HIP(hipEventRecord(ev[3 * it + 0], s));
        rc = launch_check(h, method, alpha_for(alpha, it + 1), g, h->d_synd, s);
        SC_HIP(hipEventRecord(ev[3 * it + 1], s));
        if (!rc) rc = launch_var(h, g, nullptr, h->d_hard, h->d_done, 0, s);
        SC_HIP(hipEventRecord(ev[3 * it + 2], s));
    }
    if (!rc) {
        SC_HIP(hipStreamSynchronize(s));
        ms[0] = ms[1] = 0.0f;
        for (int it = 0; it < iters; it++) {
            float a = 0, b = 0;
            SC_HIP(hipEventElapsedTime(&a, ev[3 * it + 0], ev[3 * it + 1]));
            SC_HIP(hipEventElapsedTime(&b, ev[3 * it + 1], ev[3 * it + 2]));
            ms[0] += a;
            ms[1] += b;
        }
        launches[0] = iters * (method == SCALDPC_BP_MIN_SUM ? 1 : (int)h->row_buckets.size());
        launches[1] = iters * (int)h->var_buckets.size();
        launches[2] = g * TW;  // codewords swept per pass
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

void scaldpc_bp_destroy(scaldpc_bp *h)
{
    if (!h) return;
    dev_free(h->d_row_ptr); dev_free(h->d_col_idx); dev_free(h->d_col_ptr); dev_free(h->d_csc_edge);
    dev_free(h->d_var_list); dev_free(h->d_row_list); dev_free(h->d_prior);
    dev_free(h->d_v2c); dev_free(h->d_c2v); dev_free(h->d_post);
    dev_free(h->d_synd); dev_free(h->d_recv); dev_free(h->d_hard); dev_free(h->d_done);
    dev_free(h->d_conv); dev_free(h->d_unsat); dev_free(h->d_iters); dev_free(h->d_remaining);
    dev_free(h->d_in); dev_free(h->d_out_bits); dev_free(h->d_out_conv); dev_free(h->d_out_llr);
    dev_free(h->d_out_iters);
    if (h->h_remaining) (void)hipHostFree(h->h_remaining);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

}  // extern "C"

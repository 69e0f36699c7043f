// libscaldpc -- q-ary min-sum decoders on MI355X (gfx950).
//
// Replaces the in-tree Rust decoders of the reference's `simulate_rs` crate
// (simulate-with-python/simulate_rs/src/):
//   Decoder::new / min_sum / into_llr        decoder.rs:494-553, 560-666, 668-692
//   DecoderSpecial::new / min_sum            decoder_special.rs:387-464, 471-617
// reached from Python through pydecoder.rs:24-65 / 96-145.
//
// Arithmetic is the reference's, in its order, in f32: per check the minimum over all
// assignments d with sum d = 0 (over the integers) of S - alpha_j[d_j], S summed left to
// right from 0.0 exactly as `.sum()` does (decoder.rs:600-610); per variable channel +
// sum(c2v * h), minus self, normalised by the first minimum (decoder.rs:634-652); hard
// decision = first argmin of the total at the last iteration.  Minima are exact, so the
// order in which assignments are enumerated does not matter; everything else is
// add/subtract in the reference's order => hard decisions bit-exact with the oracle.
//
// Parallelisation: lane = codeword (batch innermost), thread = (node, codeword).
//   msg : float [edge][W][Bp]     one array, updated in place (v2c <-> c2v)
//   llr : float [var][Q][Bp]
// The enumeration indexes the alphabet with per-lane run-time digits, which rules out
// registers; per-thread alpha / beta vectors are staged in LDS laid out [slot][thread],
// so that whatever slot each lane picks, lane l always hits bank l (conflict free).
// This path is ALU/LDS bound (Q^(DC-1) assignments per check), not HBM bound; no
// roofline claim is made for it.
#include "scaldpc_common.h"
#include "scaldpc_logf.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

using namespace scaldpc;
typedef unsigned long long u64;
#include "scaldpc_qary_special.h"
#include "scaldpc_qary_rows.h"

namespace {

constexpr int QERR_PMF = 3;        // decoder.rs:683-684 assert

// decoder.rs:668-692 on the device: llr[q] = ln(max_p / p[q]) in f32, with glibc's logf restated
// for the device (scaldpc_logf.h) and the correctly rounded f32 division, so the LLRs are bit for bit
// what the reference's f32::ln gives on the host -- for host and device inputs alike.
// pmf: [batch][nv][Q] -> llr [nv][Q][Bp].  thread = (variable, codeword).
// A row that does not sum to 1 +- 1e-3 (or has no maximum: all NaN) is the reference's assert
// (decoder.rs:683-684): the smallest offending (codeword, variable) is left in *first_bad.
__global__ void k_q_into_llr(const float *__restrict__ pmf, int nv, int Q, int batch, long Bp,
                             float *__restrict__ llr, int *__restrict__ err, u64 *__restrict__ first_bad, int kind)
{
    const long b = (long)blockIdx.y * blockDim.x + threadIdx.x;
    const int v = blockIdx.x;
    if (b >= Bp) return;
    if (b >= batch) {  // padding lanes decode a harmless all-equal message
        for (int q = 0; q < Q; q++) llr[((size_t)v * Q + q) * Bp + b] = 0.0f;
        return;
    }
    const float *p = pmf + ((size_t)b * nv + v) * Q;
    float sum = 0.0f, mx = 0.0f;
    bool have = false;
    for (int q = 0; q < Q; q++) {
        sum += p[q];
        if (p[q] == p[q] && (!have || p[q] > mx)) {
            mx = p[q];
            have = true;
        }
    }
    if (!have || !(sum < 1.0f + 0.001f) || !(sum > 1.0f - 0.001f)) {
        atomicMax(err, QERR_PMF);
        // key: codeword, then alphabet (0 = coefficient rows, 1 = row-sum rows), then variable, then "no maximum"
        atomicMax(first_bad, ~(((u64)b << 32) | ((u64)kind << 31) | ((u64)v << 1) | (have ? 0ull : 1ull)));  // (kept inverted: see scaldpc_qary::d_status)
    }
    // measured channel outputs repeat a handful of rows: no point caching across lanes, the double
    // pipe is idle anyway (18 double operations per symbol)
    for (int q = 0; q < Q; q++) llr[((size_t)v * Q + q) * Bp + b] = glibc_logf(mx / p[q]);
}

// The same conversion through an LDS tile: the input is [codeword][variable][Q] (a codeword's pmf rows are contiguous), the
// output [variable][Q][codeword] -- with thread = (variable, codeword) and lane = codeword every lane read its own 12-byte
// row from a different cache line (28 us for config 4's 1024 x 450 x 3 floats).  Here a workgroup takes 64 codewords x VT
// variables, one wave per variable: the waves load each codeword's VT * Q contiguous floats with neighbouring lanes (one
// or two sectors per row), then wave w, lane = codeword, converts variable w from LDS (row stride 33: conflict free) and
// writes llr with 64 codewords per store.  Same arithmetic, same error key, as many waves as before.
// grid (ceil(nv / VT), Bp / 64), block 64 * VT, VT = max(1, 32 / Q) (at most 10).
// With col_ptr != nullptr the wave also writes the variable's first variable-to-check messages (decoder.rs:567-573:
// v2c = channel * h, i.e. the LLR row, mirrored where h < 0) to every edge of its variable -- k_q_init's job, without the
// launch and without reading the LLRs back (vbase = index of this alphabet's first variable in the graph, W = message row width).
// A SECOND alphabet can ride in the same launch (DecoderSpecial: the coefficient rows and the row-sum rows): blocks nb0 .. of
// grid.x convert pmf1 (nv1 rows of Q1 symbols, VT1 per block, kind 1, first variable vbase1) -- one launch less in a call
// that is made of ~10 us launches; the block is sized for the larger VT, the waves beyond a segment's VT only help load.
__global__ void k_q_into_llr_tiled(const float *__restrict__ pmf0, int nv0, int Q0, int VT0, int batch, long Bp,
                                   float *__restrict__ llr0, int *__restrict__ err, u64 *__restrict__ first_bad, int kind0,
                                   const int *__restrict__ col_ptr = nullptr, const int *__restrict__ csc_edge = nullptr,
                                   const int *__restrict__ edge_h = nullptr, float *__restrict__ msg = nullptr, int W = 0,
                                   int vbase0 = 0, int nb0 = 0x7fffffff, const float *__restrict__ pmf1 = nullptr, int nv1 = 0,
                                   int Q1 = 0, int VT1 = 0, float *__restrict__ llr1 = nullptr, int vbase1 = 0)
{
    __shared__ float tile[64 * 33];
    const bool seg1 = (int)blockIdx.x >= nb0;
    const float *__restrict__ pmf = seg1 ? pmf1 : pmf0;
    float *__restrict__ llr = seg1 ? llr1 : llr0;
    const int nv = seg1 ? nv1 : nv0, Q = seg1 ? Q1 : Q0, VT = seg1 ? VT1 : VT0, kind = seg1 ? 1 : kind0, vbase = seg1 ? vbase1 : vbase0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int v0 = ((int)blockIdx.x - (seg1 ? nb0 : 0)) * VT;
    const long b0 = (long)blockIdx.y * 64;
    const int nvv = min(VT, nv - v0), width = nvv * Q;  // floats per codeword in this tile (<= 32)
    // (all threads of the block over the tile's 64 x width floats: every load instruction has 64 active lanes)
    for (int idx = threadIdx.x; idx < 64 * width; idx += blockDim.x) {
        const int c = idx / width, l = idx - c * width;
        if (b0 + c < batch) tile[c * 33 + l] = pmf[((size_t)(b0 + c) * nv + v0) * Q + l];
    }
    __syncthreads();
    if (w >= nvv) return;
    const int v = v0 + w;
    const long b = b0 + lane;
    const int c0 = col_ptr ? col_ptr[vbase + v] : 0, c1 = col_ptr ? col_ptr[vbase + v + 1] : 0;
    if (b >= batch) {  // padding lanes decode a harmless all-equal message
        for (int q = 0; q < Q; q++) llr[((size_t)v * Q + q) * Bp + b] = 0.0f;
        for (int t = c0; t < c1; t++)
            for (int q = 0; q < Q; q++) msg[((size_t)csc_edge[t] * W + q) * Bp + b] = 0.0f;
        return;
    }
    const float *p = tile + lane * 33 + w * Q;
    float sum = 0.0f, mx = 0.0f;
    bool have = false;
    for (int q = 0; q < Q; q++) {
        sum += p[q];
        if (p[q] == p[q] && (!have || p[q] > mx)) {
            mx = p[q];
            have = true;
        }
    }
    if (!have || !(sum < 1.0f + 0.001f) || !(sum > 1.0f - 0.001f)) {
        atomicMax(err, QERR_PMF);
        atomicMax(first_bad, ~(((u64)b << 32) | ((u64)kind << 31) | ((u64)v << 1) | (have ? 0ull : 1ull)));  // (kept inverted: see scaldpc_qary::d_status)
    }
    float *own = tile + lane * 33 + w * Q;  // (this thread's slots of the tile: probabilities in, LLRs out)
    for (int q = 0; q < Q; q++) {
        const float l = glibc_logf(mx / p[q]);
        llr[((size_t)v * Q + q) * Bp + b] = l;
        own[q] = l;
    }
    for (int t = c0; t < c1; t++) {
        const int e = csc_edge[t];
        const bool rev = edge_h[e] < 0;
        for (int q = 0; q < Q; q++) msg[((size_t)e * W + q) * Bp + b] = own[rev ? Q - 1 - q : q];
    }
}

// The same conversion on rows as they stand: pmf [rows][Q] -> llr [rows][Q] (scaldpc_qary_into_llr).
// bad[0] = smallest row index that fails the sum test (or has no maximum), as k_q_into_llr's key.
__global__ void k_q_into_llr_rows(const float *__restrict__ pmf, long rows, int Q, float *__restrict__ llr,
                                  u64 *__restrict__ first_bad)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = pmf + (size_t)r * Q;
    float sum = 0.0f, mx = 0.0f;
    bool have = false;
    for (int q = 0; q < Q; q++) {
        sum += p[q];
        if (p[q] == p[q] && (!have || p[q] > mx)) {
            mx = p[q];
            have = true;
        }
    }
    if (!have || !(sum < 1.0f + 0.001f) || !(sum > 1.0f - 0.001f)) atomicMin(first_bad, ((u64)r << 1) | (have ? 0ull : 1ull));
    for (int q = 0; q < Q; q++) llr[(size_t)r * Q + q] = glibc_logf(mx / p[q]);
}

// decoder.rs:567-573: v2c = channel * h.  thread = (edge, codeword).
__global__ void k_q_init(const int *__restrict__ edge_var, const int *__restrict__ edge_h,
                         const int *__restrict__ var_q, const long *__restrict__ var_off,
                         const float *__restrict__ llr, float *__restrict__ msg, int W, long Bp)
{
    const long b = (long)blockIdx.y * blockDim.x + threadIdx.x;
    const int e = blockIdx.x;
    if (b >= Bp) return;
    const int v = edge_var[e], Q = var_q[v];
    const float *ch = llr + var_off[v] * Bp + b;
    const bool rev = edge_h[e] < 0;
    for (int q = 0; q < Q; q++) msg[((size_t)e * W + q) * Bp + b] = ch[(size_t)(rev ? Q - 1 - q : q) * Bp];
}

// Check-node update of Decoder (decoder.rs:585-631), finite-support enumeration
// (FiniteDValueIterator, decoder.rs:281-401), index 0 fastest.
// block = T threads = T codewords of one check; LDS: A[k*Q][T], Bt[k*Q][T] floats, fin[k*Q][T] bytes.
// WORD: the per-lane registers that hold one 8-bit digit per edge of the check (finite-symbol counts, the
// enumeration index, the chosen symbols): u64 for checks of up to 8 edges (every size the reference registers,
// lib.rs:32-75), unsigned __int128 for 9..16 -- Decoder is const-generic in DC (decoder.rs:417-438).
template <typename WORD>
__global__ void k_q_check(const int *__restrict__ row_ptr, float *msg, int Q, int B, long Bp, int batch, int maxdc,
                          int *__restrict__ err)
{
    extern __shared__ unsigned char smem[];
    const int T = blockDim.x, tid = threadIdx.x;
    float *A = (float *)smem;
    float *Bt = A + (size_t)maxdc * Q * T;
    unsigned char *fin = (unsigned char *)(Bt + (size_t)maxdc * Q * T);
    const int c = blockIdx.x;
    const long b = (long)blockIdx.y * T + tid;
    if (b >= batch) return;  // padding lanes: no barrier below, every thread owns its LDS column
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0;
    if (k == 0) {
        if (tid == 0) atomicMax(err, QERR_NO_CONFIG);
        return;
    }
    WORD nums = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) {
        int cnt = 0;
        for (int q = 0; q < Q; q++) {
            const float x = msg[((size_t)(e0 + j) * Q + q) * Bp + b];
            A[(size_t)(j * Q + q) * T + tid] = x;
            Bt[(size_t)(j * Q + q) * T + tid] = INFINITY;
            if (finite_f(x)) fin[(size_t)(j * Q + cnt++) * T + tid] = (unsigned char)q;
        }
        nums |= (WORD)cnt << (8 * j);
        bad |= cnt == 0;
    }
    if (bad) {
        atomicMax(err, QERR_NO_FINITE);
    } else {
        WORD idx = 0;
        int nconf = 0;
        for (;;) {
            int dsum = 0;
            float S = 0.0f;
            WORD qs = 0;
            for (int j = 0; j < k - 1; j++) {
                const int ij = (int)(idx >> (8 * j)) & 255;
                const int q = fin[(size_t)(j * Q + ij) * T + tid];
                qs |= (WORD)q << (8 * j);
                dsum += q - B;
                S += A[(size_t)(j * Q + q) * T + tid];
            }
            const int dl = -dsum;
            if (dl >= -B && dl <= B) {
                const int ql = dl + B;
                qs |= (WORD)ql << (8 * (k - 1));
                S += A[(size_t)((k - 1) * Q + ql) * T + tid];
                if (finite_f(S)) {
                    nconf++;
                    for (int j = 0; j < k; j++) {
                        const int q = (int)(qs >> (8 * j)) & 255;
                        const size_t o = (size_t)(j * Q + q) * T + tid;
                        Bt[o] = fminf(S - A[o], Bt[o]);
                    }
                }
            }
            int j = 0;
            for (; j < k - 1; j++) {
                const int ij = (int)(idx >> (8 * j)) & 255, nj = (int)(nums >> (8 * j)) & 255;
                if (ij + 1 < nj) {
                    idx += (WORD)1 << (8 * j);
                    break;
                }
                idx &= ~((WORD)255 << (8 * j));
            }
            if (j >= k - 1) break;
        }
        if (nconf == 0) atomicMax(err, QERR_NO_CONFIG);
    }
    for (int j = 0; j < k; j++)
        for (int q = 0; q < Q; q++)
            msg[((size_t)(e0 + j) * Q + q) * Bp + b] = Bt[(size_t)(j * Q + q) * T + tid];
}


// ---------------------------------------------------------------------------
// Small and medium batches (<= 256; single `min_sum` calls are the reference's usual pattern): lane = codeword
// would leave 63 lanes idle while one walks Q^(DC-1) assignments.  Here a WAVE owns one
// (check, codeword): the lanes split the assignment space (assignment c goes to lane
// c mod 64, stepped through mixed-radix digits), each keeps private running minima in LDS
// ([slot][lane], conflict free), and the 64 partial minima of every slot are combined with
// wave shuffles.  min is exact and every S is summed in the same order as before, so the
// messages are bit-identical to the lane = codeword kernels'.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}

// generic Decoder check (decoder.rs:585-631).  grid (R, batch), block 64.
// LDS: A[k*Q] floats (shared), fin[k*Q] + num[k] bytes (shared), Bt[k*Q][64] floats (per lane).
__global__ __launch_bounds__(64) void k_q_check_wave(const int *__restrict__ row_ptr, float *msg, int Q, int B, long Bp,
                                                     int maxdc, int *__restrict__ err)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x;
    float *A = (float *)smem;
    float *Bt = A + maxdc * Q;
    unsigned char *fin = (unsigned char *)(Bt + (size_t)maxdc * Q * 64);
    unsigned char *num = fin + maxdc * Q;
    const int c = blockIdx.x;
    const long b = blockIdx.y;
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0;
    if (k == 0) {
        if (lane == 0) atomicMax(err, QERR_NO_CONFIG);
        return;
    }
    for (int i = lane; i < k * Q; i += 64) A[i] = msg[((size_t)(e0 + i / Q) * Q + i % Q) * Bp + b];
    for (int i = 0; i < k * Q; i++) Bt[(size_t)i * 64 + lane] = INFINITY;
    __syncthreads();
    if (lane < k) {
        int cnt = 0;
        for (int q = 0; q < Q; q++)
            if (finite_f(A[lane * Q + q])) fin[lane * Q + cnt++] = (unsigned char)q;
        num[lane] = (unsigned char)cnt;
    }
    __syncthreads();
    bool bad = false;
    unsigned long long total = 1;
    for (int j = 0; j < k; j++) bad |= num[j] == 0;
    for (int j = 0; j < k - 1; j++) total *= num[j];
    if (bad) {
        if (lane == 0) atomicMax(err, QERR_NO_FINITE);
    } else {
        // digits of this lane's first assignment and of the stride 64, index 0 fastest
        u64 idx = 0, stp = 0;
        {
            unsigned a = (unsigned)lane, st = 64;  // (both start below 65: 32-bit division, not the 64-bit library routine)
            for (int j = 0; j < k - 1; j++) {
                const unsigned nj = num[j];
                idx |= (u64)(a % nj) << (8 * j);
                a /= nj;
                stp |= (u64)(st % nj) << (8 * j);
                st /= nj;
            }
        }
        int nconf = 0;
        for (unsigned long long cfg = lane; cfg < total; cfg += 64) {
            int dsum = 0;
            float S = 0.0f;
            u64 qs = 0;
            for (int j = 0; j < k - 1; j++) {
                const int q = fin[j * Q + ((int)(idx >> (8 * j)) & 255)];
                qs |= (u64)q << (8 * j);
                dsum += q - B;
                S += A[j * Q + q];
            }
            const int dl = -dsum;
            if (dl >= -B && dl <= B) {
                const int ql = dl + B;
                qs |= (u64)ql << (8 * (k - 1));
                S += A[(k - 1) * Q + ql];
                if (finite_f(S)) {
                    nconf++;
                    for (int j = 0; j < k; j++) {
                        const int q = (int)(qs >> (8 * j)) & 255;
                        float *bb = &Bt[(size_t)(j * Q + q) * 64 + lane];
                        *bb = fminf(S - A[j * Q + q], *bb);
                    }
                }
            }
            // idx += stride (mixed radix, one conditional subtraction per digit)
            int carry = 0;
            u64 nidx = 0;
            for (int j = 0; j < k - 1; j++) {
                int d = ((int)(idx >> (8 * j)) & 255) + ((int)(stp >> (8 * j)) & 255) + carry;
                carry = d >= num[j];
                if (carry) d -= num[j];
                nidx |= (u64)d << (8 * j);
            }
            idx = nidx;
        }
        const u64 any = __ballot(nconf > 0);
        if (!any && lane == 0) atomicMax(err, QERR_NO_CONFIG);
    }
    // minimum over the 64 lanes' partial results, TRANSPOSED: lane L folds whole rows i = L, L + 64, ... of Bt (64 LDS reads
    // each, rotated by the lane index so that the 64 lanes hit 32 different banks) instead of 6 dependent ds_bpermute steps
    // per entry (k * Q entries: 630 of them for a degree-7 check over Q = 15)
    __syncthreads();
    for (int i = lane; i < k * Q; i += 64) {
        const float *row = Bt + (size_t)i * 64;
        float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
#pragma unroll 4
        for (int l = 0; l < 64; l += 4) {
            m0 = vmin(m0, row[(l + lane) & 63]);
            m1 = vmin(m1, row[(l + 1 + lane) & 63]);
            m2 = vmin(m2, row[(l + 2 + lane) & 63]);
            m3 = vmin(m3, row[(l + 3 + lane) & 63]);
        }
        msg[((size_t)(e0 + i / Q) * Q + i % Q) * Bp + b] = vmin(vmin(m0, m1), vmin(m2, m3));
    }
}

// DecoderSpecial check (decoder_special.rs:506-563), wave per (check, codeword).
// LDS: Ab[nb*QB], As[QS] floats (shared), Bb[nb*QB][64], Bs[QS][64] floats (per lane).
__global__ __launch_bounds__(64) void k_q_special_check_wave(const int *__restrict__ row_ptr, float *msg, int B, int BSUM,
                                                             int W, long Bp, int nbm, int skip_nb)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x;
    const int QB = 2 * B + 1, QS = 2 * BSUM + 1;
    float *Ab = (float *)smem;
    float *As = Ab + nbm * QB;
    float *Bb = As + QS;
    float *Bs = Bb + (size_t)nbm * QB * 64;
    const int c = blockIdx.x;
    const long b = blockIdx.y;
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0, nb = k - 1;
    if (nb == skip_nb) return;  // rows of this degree belong to k_q_special_check_tree
    for (int i = lane; i < nb * QB; i += 64) Ab[i] = msg[((size_t)(e0 + i / QB) * W + i % QB) * Bp + b];
    for (int i = lane; i < QS; i += 64) As[i] = msg[((size_t)(e0 + nb) * W + i) * Bp + b];
    for (int i = 0; i < nb * QB; i++) Bb[(size_t)i * 64 + lane] = INFINITY;
    for (int i = 0; i < QS; i++) Bs[(size_t)i * 64 + lane] = INFINITY;
    __syncthreads();
    unsigned long long total = 1;
    for (int j = 0; j < nb; j++) total *= QB;
    u64 dq = 0, stp = 0;
    {
        unsigned a = (unsigned)lane, st = 64;  // (both start below 65: 32-bit division)
        const unsigned uq = (unsigned)QB;
        for (int j = 0; j < nb; j++) {
            dq |= (u64)(a % uq) << (8 * j);
            a /= uq;
            stp |= (u64)(st % uq) << (8 * j);
            st /= uq;
        }
    }
    for (unsigned long long cfg = lane; cfg < total; cfg += 64) {
        int dsum = 0;
        float S = 0.0f;
        for (int j = 0; j < nb; j++) {
            const int q = (int)(dq >> (8 * j)) & 255;
            dsum += q - B;
            S += Ab[j * QB + q];
        }
        const int os = -dsum + BSUM;
        S += As[os];
        for (int j = 0; j < nb; j++) {
            const int q = (int)(dq >> (8 * j)) & 255;
            float *bb = &Bb[(size_t)(j * QB + q) * 64 + lane];
            *bb = fminf(*bb, S - Ab[j * QB + q]);
        }
        Bs[(size_t)os * 64 + lane] = fminf(Bs[(size_t)os * 64 + lane], S - As[os]);
        int carry = 0;
        u64 ndq = 0;
        for (int j = 0; j < nb; j++) {
            int d = ((int)(dq >> (8 * j)) & 255) + ((int)(stp >> (8 * j)) & 255) + carry;
            carry = d >= QB;
            if (carry) d -= QB;
            ndq |= (u64)d << (8 * j);
        }
        dq = ndq;
    }
    // minima over the 64 lanes, transposed (see k_q_check_wave): the row's nb * QB coefficient entries, then its QS sum entries
    __syncthreads();
    for (int i = lane; i < nb * QB + QS; i += 64) {
        const float *row = i < nb * QB ? Bb + (size_t)i * 64 : Bs + (size_t)(i - nb * QB) * 64;
        float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
#pragma unroll 4
        for (int l = 0; l < 64; l += 4) {
            m0 = vmin(m0, row[(l + lane) & 63]);
            m1 = vmin(m1, row[(l + 1 + lane) & 63]);
            m2 = vmin(m2, row[(l + 2 + lane) & 63]);
            m3 = vmin(m3, row[(l + 3 + lane) & 63]);
        }
        const float v = vmin(vmin(m0, m1), vmin(m2, m3));
        if (i < nb * QB)
            msg[((size_t)(e0 + i / QB) * W + i % QB) * Bp + b] = v;
        else
            msg[((size_t)(e0 + nb) * W + (i - nb * QB)) * Bp + b] = v;
    }
}

// Variable-node update (decoder.rs:634-658 / decoder_special.rs:566-609).
// thread = (variable, codeword); LDS: sum[Qmax][T], tmp[Qmax][T].
__global__ void k_q_var(int v0, const int *__restrict__ col_ptr, const int *__restrict__ csc_edge,
                        const int *__restrict__ edge_h, const int *__restrict__ var_q,
                        const long *__restrict__ var_off, const float *__restrict__ llr, float *msg, int W, long Bp,
                        int batch, int Qmax, int last, signed char *__restrict__ out)
{
    extern __shared__ unsigned char smem[];
    const int T = blockDim.x, tid = threadIdx.x;
    float *sum = (float *)smem;
    float *tmp = sum + (size_t)Qmax * T;
    const int v = v0 + blockIdx.x;
    const long b = (long)blockIdx.y * T + tid;
    if (b >= batch) return;
    const int Q = var_q[v], Bv = (Q - 1) / 2;
    const float *ch = llr + var_off[v] * Bp + b;
    for (int q = 0; q < Q; q++) sum[(size_t)q * T + tid] = ch[(size_t)q * Bp];
    const int c0 = col_ptr[v], c1 = col_ptr[v + 1];
    for (int t = c0; t < c1; t++) {
        const int e = csc_edge[t];
        const bool rev = edge_h[e] < 0;
        const float *in = msg + (size_t)e * W * Bp + b;
        for (int q = 0; q < Q; q++) sum[(size_t)q * T + tid] = sum[(size_t)q * T + tid] + in[(size_t)(rev ? Q - 1 - q : q) * Bp];
    }
    for (int t = c0; t < c1; t++) {
        const int e = csc_edge[t];
        const bool rev = edge_h[e] < 0;
        float *io = msg + (size_t)e * W * Bp + b;
        // prim_out = (sum - c2v*h) * h   (qary_sub_with_mult_in_gf then mult_in_gf)
        for (int q = 0; q < Q; q++) {
            const int qi = rev ? Q - 1 - q : q;
            tmp[(size_t)qi * T + tid] = sum[(size_t)q * T + tid] - io[(size_t)qi * Bp];
        }
        float mv = INFINITY;
        int ma = 0;
        for (int q = 0; q < Q; q++) {
            const float x = tmp[(size_t)q * T + tid];
            if (x < mv) {
                mv = x;
                ma = q;
            }
        }
        const float mn = tmp[(size_t)ma * T + tid];
        for (int q = 0; q < Q; q++) io[(size_t)q * Bp] = tmp[(size_t)q * T + tid] - mn;
    }
    if (last) {
        float mv = INFINITY;
        int ma = 0;
        for (int q = 0; q < Q; q++) {
            const float x = sum[(size_t)q * T + tid];
            if (x < mv) {
                mv = x;
                ma = q;
            }
        }
        out[(size_t)v * Bp + b] = (signed char)(ma - Bv);
    }
}

// The same update with everything in registers, for the plain decoder with alphabets Q = 3, 5, 7, 15 and columns of at
// most DMAX checks: every incoming message is loaded ONCE (the generic kernel reads each twice, with an LDS round trip
// between global accesses), all of a column's loads are issued before the first add.  Same additions and subtractions in
// the same order, the same first-minimum rule: identical symbols.  llr is [var][Q][Bp] here (one alphabet).
// grid (N, Bp/64), block 64.
//   v: variable (graph index: column of col_ptr, row of `out`);  llr: this variable's Q rows;  W: width of a message row
template <int Q, int DMAX>
__device__ __forceinline__ void var_small_body(int v, const float *__restrict__ llr, const int *__restrict__ col_ptr,
                                               const int *__restrict__ csc_edge, const int *__restrict__ edge_h, float *msg, int W,
                                               long Bp, long b, int last, signed char *__restrict__ out)
{
    const int c0 = col_ptr[v], deg = col_ptr[v + 1] - c0;
    float sum[Q], in[DMAX][Q];
    int ed[DMAX];
    bool rv[DMAX];
#pragma unroll
    for (int q = 0; q < Q; q++) sum[q] = llr[(size_t)q * Bp + b];
#pragma unroll
    for (int t = 0; t < DMAX; t++) {
        ed[t] = 0;
        rv[t] = false;
        if (t < deg) {
            ed[t] = csc_edge[c0 + t];
            rv[t] = edge_h[ed[t]] < 0;
#pragma unroll
            for (int q = 0; q < Q; q++) in[t][q] = msg[((size_t)ed[t] * W + q) * Bp + b];
        }
    }
#pragma unroll
    for (int t = 0; t < DMAX; t++)
        if (t < deg) {
#pragma unroll
            for (int q = 0; q < Q; q++) sum[q] = sum[q] + (rv[t] ? in[t][Q - 1 - q] : in[t][q]);
        }
#pragma unroll
    for (int t = 0; t < DMAX; t++)
        if (t < deg) {
            float tmp[Q];  // tmp[qi] = sum[q] - c2v[qi], qi = q mirrored where h < 0
#pragma unroll
            for (int q = 0; q < Q; q++) tmp[q] = (rv[t] ? sum[Q - 1 - q] : sum[q]) - in[t][q];
            float mv = INFINITY, mn = tmp[0];  // first strict minimum; default index 0 (NaN never selected)
#pragma unroll
            for (int q = 0; q < Q; q++)
                if (tmp[q] < mv) {
                    mv = tmp[q];
                    mn = tmp[q];
                }
#pragma unroll
            for (int q = 0; q < Q; q++) msg[((size_t)ed[t] * W + q) * Bp + b] = tmp[q] - mn;
        }
    if (last) {
        float mv = INFINITY;
        int ma = 0;
#pragma unroll
        for (int q = 0; q < Q; q++)
            if (sum[q] < mv) {
                mv = sum[q];
                ma = q;
            }
        out[(size_t)v * Bp + b] = (signed char)(ma - (Q - 1) / 2);
    }
}

template <int Q, int DMAX>
__global__ __launch_bounds__(64) void k_q_var_small(const int *__restrict__ col_ptr, const int *__restrict__ csc_edge,
                                                    const int *__restrict__ edge_h, const float *__restrict__ llr, float *msg,
                                                    long Bp, int batch, int last, signed char *__restrict__ out)
{
    const int v = blockIdx.x;
    const long b = (long)blockIdx.y * 64 + threadIdx.x;
    if (b >= batch) return;
    var_small_body<Q, DMAX>(v, llr + (size_t)v * Q * Bp, col_ptr, csc_edge, edge_h, msg, Q, Bp, b, last, out);
}

// DecoderSpecial (decoder_special.rs:566-609): the first BV variables over QA symbols (columns of at most DA checks), the
// row-sum variables behind them over QS symbols, one check each; message rows are W = max(QA, QS) wide.
// grid (N, Bp/64), block 64.
template <int QA, int DA, int QS>
__global__ __launch_bounds__(64) void k_q_var_small_special(const int *__restrict__ col_ptr, const int *__restrict__ csc_edge,
                                                            const int *__restrict__ edge_h, const float *__restrict__ llr,
                                                            float *msg, int BV, int W, long Bp, int batch, int last,
                                                            signed char *__restrict__ out)
{
    const int v = blockIdx.x;
    const long b = (long)blockIdx.y * 64 + threadIdx.x;
    if (b >= batch) return;
    if (v < BV)
        var_small_body<QA, DA>(v, llr + (size_t)v * QA * Bp, col_ptr, csc_edge, edge_h, msg, W, Bp, b, last, out);
    else
        var_small_body<QS, 1>(v, llr + ((size_t)BV * QA + (size_t)(v - BV) * QS) * Bp, col_ptr, csc_edge, edge_h, msg, W, Bp, b, last, out);
}

// [N][Bp] -> [batch][N]
__global__ void k_q_unpack(const signed char *__restrict__ in, int N, int batch, long Bp, signed char *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (v < N && b < batch) out[(size_t)b * N + v] = in[(size_t)v * Bp + b];
}

}  // namespace

struct scaldpc_qary {
    bool special = false;
    int R = 0, N = 0, B = 0, BSUM = 0, Q = 0, QS = 0, W = 0, iterations = 0;
    int E = 0, maxdc = 0, mindc = 0, maxdv = 0;
    long llr_rows = 0;  // total alphabet rows over all variables
    int *d_row_ptr = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr, *d_edge_var = nullptr, *d_edge_h = nullptr,
        *d_var_q = nullptr;
    long *d_var_off = nullptr;
    std::vector<int> h_var_q;
    std::vector<long> h_var_off;
    long cap_bp = 0;
    float *d_msg = nullptr, *d_llr = nullptr, *d_pmf = nullptr, *d_pmf2 = nullptr;
    signed char *d_hard = nullptr, *d_out = nullptr;
    size_t cap_pmf = 0, cap_pmf2 = 0, cap_out = 0;
    // one 16-byte status block per handle, zeroed by ONE fill and read back by ONE copy per call: [0] = the bitwise complement of
    // the smallest (codeword, variable) key whose pmf row fails the sum test (0: none; kept inverted so that "none" is zero
    // and the kernels lower the key with atomicMax), [1] = the call's error code (low word)
    u64 *d_status = nullptr;
    int *d_err = nullptr;        // = (int *)(d_status + 1)
    u64 *d_first_bad = nullptr;  // = d_status
    hipStream_t own_stream = nullptr;
    int device = 0;      // the device the handle was created on; every entry point runs there
    int kn_wave = -1;    // -1: wave-parallel enumeration for batches <= 256 and the special decoder; 0 / 1 force
    int kn_unroll = 1;   // register-resident unrolled enumeration for small alphabets
    int kn_tree = 1;     // special decoder: tree-walk check kernel for the Kyber shape (QB = 5, 6 coefficient edges)
    int kn_dp = 1;       // special decoder, same shape: min-plus recursion instead of the enumeration (k_q_special_check_dp); batches >= kn_dp_min
    int kn_dp_min = 5;      // (below, one wave per (check, codeword) of the tree walk is as fast or faster: profiles/r04/kyber_form_sweep.log)
    int kn_dp_split = 64;    // up to this batch the row's edges are split over four waves (same log)
    int kn_dp_split2 = 192;  // ... and up to this one over two
    // measurement aid (bench.py): with "timing" = 1 every check / variable launch of a call is bracketed by HIP events
    // on the launch stream; scaldpc_qary_last_timing reads the sums.  Off by default: the product path records nothing.
    int kn_llr_tiled = 1;  // probability -> LLR conversion through an LDS tile (coalesced reads); A/B knob "llr_tiled"
    int kn_var_small = 1;  // register-resident variable update for Q = 3 / 5 / 7 / 15 and columns of at most 4 checks (A/B knob "var_small")
    int kn_timing = 0;
    std::vector<hipEvent_t> tev;
    float stat_ms_check = 0.f, stat_ms_var = 0.f, stat_ms_call = 0.f;
    int stat_iters = 0, stat_kernel = -1, stat_batch = 0;
    std::mutex mu;
};

namespace {

int qary_build(int R, int N, int B, int BSUM, bool special, const int8_t *H, int iterations, scaldpc_qary **out)
{
    if (!out) return fail(SCALDPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (R <= 0 || N <= 0 || B < 1 || !H || iterations < 0)
        return fail(SCALDPC_EINVAL, "bad q-ary decoder arguments (R=%d N=%d B=%d)", R, N, B);
    if (B > 127) return fail(SCALDPC_EDEGREE, "B=%d: hard decisions are int8 (decoder.rs i8)", B);
    if (special) {
        if (BSUM < 1 || BSUM % B != 0)
            return fail(SCALDPC_EINVAL, "BSUM (%d) must be multiple of B (%d)", BSUM, B);  // decoder_special.rs:388-392
        if (N <= R) return fail(SCALDPC_EINVAL, "special decoder needs N > R");
        if (BSUM > 127) return fail(SCALDPC_EDEGREE, "BSUM=%d: hard decisions are int8", BSUM);
    }
    const int BV = special ? N - R : N;
    std::vector<int> row_ptr(R + 1, 0), col_cnt(N, 0), edge_var, edge_h;
    for (int r = 0; r < R; r++) {
        for (int c = 0; c < N; c++) {
            const int h = H[(size_t)r * N + c];
            if (!h) continue;
            if (h != 1 && h != -1) return fail(SCALDPC_EINVAL, "H[%d][%d] = %d: entries must be in {-1,0,1}", r, c, h);
            edge_var.push_back(c);
            edge_h.push_back(h);
            col_cnt[c]++;
        }
        row_ptr[r + 1] = (int)edge_var.size();
    }
    const int E = (int)edge_var.size();
    int maxdc = 0;
    for (int r = 0; r < R; r++) maxdc = std::max(maxdc, row_ptr[r + 1] - row_ptr[r]);
    // one 8-bit digit per edge of a check in a register word: 64 bits for degree <= 8 (all kernels), 128 bits for 9..16
    // (the lane-per-codeword kernel only: Decoder is const-generic in DC, decoder.rs:417-438; the reference registers 4 and 7)
    if (maxdc > (special ? 8 : 16))
        return fail(SCALDPC_EDEGREE, "check degree %d > %d is not supported by the enumeration kernels", maxdc, special ? 8 : 16);
    if (special) {
        for (int r = 0; r < R; r++) {
            const int k = row_ptr[r + 1] - row_ptr[r];
            if (k < 1) return fail(SCALDPC_EINVAL, "special decoder: check %d is empty", r);
            for (int j = 0; j < k - 1; j++)
                if (edge_var[row_ptr[r] + j] >= BV)
                    return fail(SCALDPC_EINVAL, "special decoder: H is not of the form [H' | I] (row %d)", r);
            if (edge_var[row_ptr[r] + k - 1] < BV)
                return fail(SCALDPC_EINVAL, "special decoder: row %d has no row-sum variable (H != [H' | I])", r);
            if ((k - 1) * B > BSUM)
                return fail(SCALDPC_EINVAL, "special decoder: (degree-1)*B = %d exceeds BSUM = %d in row %d", (k - 1) * B,
                            BSUM, r);
        }
        for (int v = BV; v < N; v++)
            if (col_cnt[v] != 1) return fail(SCALDPC_EINVAL, "special decoder: row-sum variable %d has degree %d", v, col_cnt[v]);
    }
    std::vector<int> col_ptr(N + 1, 0), csc_edge(E), fill(N, 0);
    for (int v = 0; v < N; v++) col_ptr[v + 1] = col_ptr[v] + col_cnt[v];
    for (int e = 0; e < E; e++) csc_edge[col_ptr[edge_var[e]] + fill[edge_var[e]]++] = e;

    scaldpc_qary *h = new (std::nothrow) scaldpc_qary();
    if (!h) return fail(SCALDPC_ENOMEM, "out of host memory");
    h->special = special;
    h->R = R; h->N = N; h->B = B; h->BSUM = BSUM;
    h->Q = 2 * B + 1;
    h->QS = special ? 2 * BSUM + 1 : h->Q;
    h->W = std::max(h->Q, h->QS);
    h->iterations = iterations;
    h->E = E;
    h->maxdc = maxdc;
    h->mindc = maxdc;
    for (int v = 0; v < N; v++) h->maxdv = std::max(h->maxdv, col_cnt[v]);
    for (int r = 0; r < R; r++) h->mindc = std::min(h->mindc, row_ptr[r + 1] - row_ptr[r]);
    h->h_var_q.resize(N);
    h->h_var_off.resize(N);
    long off = 0;
    for (int v = 0; v < N; v++) {
        h->h_var_q[v] = v < BV ? h->Q : h->QS;
        h->h_var_off[v] = off;
        off += h->h_var_q[v];
    }
    h->llr_rows = off;
    int rc = 0;
    auto up = [&](int **d, const int *src, size_t cnt) -> int {
        SC_TRY(dev_alloc(d, cnt));
        if (cnt) SC_HIP(hipMemcpy(*d, src, cnt * sizeof(int), hipMemcpyHostToDevice));
        return 0;
    };
    if (!rc) rc = up(&h->d_row_ptr, row_ptr.data(), R + 1);
    if (!rc) rc = up(&h->d_col_ptr, col_ptr.data(), N + 1);
    if (!rc) rc = up(&h->d_csc_edge, csc_edge.data(), E);
    if (!rc) rc = up(&h->d_edge_var, edge_var.data(), E);
    if (!rc) rc = up(&h->d_edge_h, edge_h.data(), E);
    if (!rc) rc = up(&h->d_var_q, h->h_var_q.data(), N);
    if (!rc) rc = dev_alloc(&h->d_var_off, (size_t)N);
    if (!rc && hipMemcpy(h->d_var_off, h->h_var_off.data(), sizeof(long) * N, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(SCALDPC_EHIP, "hipMemcpy failed");
    if (!rc) rc = dev_alloc(&h->d_status, 2);
    if (!rc) {
        h->d_first_bad = h->d_status;
        h->d_err = (int *)(h->d_status + 1);
    }
    if (!rc && hipGetDevice(&h->device) != hipSuccess) rc = fail(SCALDPC_EHIP, "hipGetDevice failed");
    if (const char *e = getenv("SCALDPC_QARY_WAVE")) h->kn_wave = atoi(e) != 0;  // the environment is read once per handle
    if (getenv("SCALDPC_QARY_NO_UNROLL")) h->kn_unroll = 0;
    if (getenv("SCALDPC_QARY_NO_TREE")) h->kn_tree = 0;
    if (!rc && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(SCALDPC_EHIP, "hipStreamCreate failed");
    if (rc) {
        scaldpc_qary_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

template <typename T>
int growq(T **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return 0;
    dev_free(*p);
    *cap = 0;
    SC_TRY(dev_alloc(p, need));
    *cap = need;
    return 0;
}

int qary_run(scaldpc_qary *h, const float *pmf_b, const float *pmf_s, int batch, uint32_t flags, void *stream,
             int8_t *out)
{
    if (!h || !pmf_b || !out || (h->special && !pmf_s)) return fail(SCALDPC_EINVAL, "NULL argument");
    if (batch <= 0) return fail(SCALDPC_EINVAL, "batch must be positive");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device);
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO;
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    const long Bp = ((long)batch + 63) / 64 * 64;
    const int BV = h->special ? h->N - h->R : h->N;
    if (Bp > h->cap_bp) {
        dev_free(h->d_msg); dev_free(h->d_llr); dev_free(h->d_hard);
        h->cap_bp = 0;
        SC_TRY(dev_alloc(&h->d_msg, (size_t)std::max(h->E, 1) * h->W * Bp));
        SC_TRY(dev_alloc(&h->d_llr, (size_t)h->llr_rows * Bp));
        SC_TRY(dev_alloc(&h->d_hard, (size_t)h->N * Bp));
        h->cap_bp = Bp;
    }
    SC_HIP(hipMemsetAsync(h->d_status, 0, 2 * sizeof(u64), s));
    const int TB = 64;
    // probabilities -> LLRs on the device (host inputs are staged as they are: [batch][var][Q] floats)
    const float *dp_b = pmf_b, *dp_s = pmf_s;
    if (!dev_io) {
        const size_t nb = (size_t)batch * BV * h->Q, ns = h->special ? (size_t)batch * h->R * h->QS : 0;
        SC_TRY(growq(&h->d_pmf, &h->cap_pmf, nb));
        SC_HIP(hipMemcpyAsync(h->d_pmf, pmf_b, nb * sizeof(float), hipMemcpyHostToDevice, s));
        dp_b = h->d_pmf;
        if (h->special) {
            SC_TRY(growq(&h->d_pmf2, &h->cap_pmf2, ns));
            SC_HIP(hipMemcpyAsync(h->d_pmf2, pmf_s, ns * sizeof(float), hipMemcpyHostToDevice, s));
            dp_s = h->d_pmf2;
        }
    }
    // (alphabets of up to 32 symbols go through the LDS-tiled form: coalesced reads of [codeword][variable][Q])
    // the tiled conversion also writes the first variable-to-check messages (k_q_init's job) when every alphabet takes it
    const bool fused_init = h->kn_llr_tiled && h->Q <= 32 && (!h->special || h->QS <= 32) && h->E > 0;
    auto into_llr = [&](const float *dp, int nv, int Q, float *llr, int kind) {
        if (Q <= 32 && h->kn_llr_tiled) {
            const int VT = std::max(1, 32 / Q);
            if (fused_init)
                hipLaunchKernelGGL(k_q_into_llr_tiled, dim3((nv + VT - 1) / VT, Bp / 64), dim3(64 * VT), 0, s, dp, nv, Q, VT, batch,
                                   Bp, llr, h->d_err, h->d_first_bad, kind, (const int *)h->d_col_ptr, (const int *)h->d_csc_edge,
                                   (const int *)h->d_edge_h, h->d_msg, h->W, kind ? BV : 0);
            else
                hipLaunchKernelGGL(k_q_into_llr_tiled, dim3((nv + VT - 1) / VT, Bp / 64), dim3(64 * VT), 0, s, dp, nv, Q, VT, batch,
                                   Bp, llr, h->d_err, h->d_first_bad, kind);
        } else
            hipLaunchKernelGGL(k_q_into_llr, dim3(nv, Bp / TB), dim3(TB), 0, s, dp, nv, Q, batch, Bp, llr, h->d_err, h->d_first_bad,
                               kind);
    };
    if (h->special && fused_init) {  // both alphabets in one launch
        const int VT0 = std::max(1, 32 / h->Q), VT1 = std::max(1, 32 / h->QS), nb0 = (BV + VT0 - 1) / VT0, nb1 = (h->R + VT1 - 1) / VT1;
        hipLaunchKernelGGL(k_q_into_llr_tiled, dim3(nb0 + nb1, Bp / 64), dim3(64 * std::max(VT0, VT1)), 0, s, dp_b, BV, h->Q, VT0, batch, Bp,
                           h->d_llr, h->d_err, h->d_first_bad, 0, (const int *)h->d_col_ptr, (const int *)h->d_csc_edge,
                           (const int *)h->d_edge_h, h->d_msg, h->W, 0, nb0, dp_s, h->R, h->QS, VT1,
                           h->d_llr + (size_t)BV * h->Q * Bp, BV);
        SC_HIP(hipGetLastError());
    } else {
        into_llr(dp_b, BV, h->Q, h->d_llr, 0);
        SC_HIP(hipGetLastError());
        if (h->special) {
            into_llr(dp_s, h->R, h->QS, h->d_llr + (size_t)BV * h->Q * Bp, 1);
            SC_HIP(hipGetLastError());
        }
    }
    if (h->E && !fused_init) {
        hipLaunchKernelGGL(k_q_init, dim3(h->E, Bp / TB), dim3(TB), 0, s, h->d_edge_var, h->d_edge_h, h->d_var_q,
                           h->d_var_off, h->d_llr, h->d_msg, h->W, Bp);
        SC_HIP(hipGetLastError());
    }
    // threads per block of the enumeration kernels: as many (<= 64) as fit 64 KB of LDS
    size_t per_thread = h->special ? (size_t)2 * ((h->maxdc - 1) * h->Q + h->QS) * 4 : (size_t)h->maxdc * h->Q * 9;
    int T = 64;
    while (T > 8 && per_thread * T > 64 * 1024) T >>= 1;
    if (per_thread * T > 64 * 1024)
        return fail(SCALDPC_EDEGREE, "alphabet/degree too large for the LDS-staged enumeration (%zu B per codeword)",
                    per_thread);
    const int iters = std::max(1, h->iterations);  // the loop body runs at least once (decoder.rs:578-579)
    // small batch: wave per (check, codeword), lanes share the assignment space
    const size_t wave_lds = h->special
                                ? (size_t)(((h->maxdc - 1) * h->Q + h->QS) * 65) * 4
                                : (size_t)h->maxdc * h->Q * 4 * 65 + (size_t)h->maxdc * h->Q + h->maxdc + 16;
    // measured: wave mode 0.69 vs 3.2 ms at batch 64 (config-4 decoder), 24 vs 70 ms (Kyber SW6);
    // a tie at batch 1024, where one codeword per lane keeps global accesses coalesced
    // the special decoder (15625 assignments per check at the Kyber shape) prefers wave mode at
    // every batch size measured (93 vs 153 ms at batch 256)
    bool wave_mode = (batch <= 256 || h->special) && wave_lds <= 64 * 1024;
    if (h->kn_wave >= 0) wave_mode = h->kn_wave != 0 && wave_lds <= 64 * 1024;
    if (h->maxdc > 8) wave_mode = false;  // 64-bit digit words in the wave kernels
    // small alphabets: fully unrolled register enumeration (any batch size)
    int unrolled = 0;
    if (!h->special && h->kn_unroll) {
        if (h->Q == 3 && h->maxdc <= 7) unrolled = 3;
        if (h->Q == 5 && h->maxdc <= 5) unrolled = 5;
    }
    // special decoder, Kyber shape (B = 2, rows of up to 6 coefficient edges + the row-sum edge): tree-walk kernel
    const int tree_nb = (h->special && h->kn_tree && (h->kn_wave != 0) && h->Q == 5 && h->maxdc - 1 == 6 && wave_lds <= 64 * 1024) ? 6 : 0;
    // ... and from a few codewords on, the min-plus recursion (lane = codeword) instead of any enumeration
    const int dp_nb = (h->special && h->kn_dp && (h->kn_wave != 0) && h->Q == 5 && h->maxdc - 1 == 6 && wave_lds <= 64 * 1024 &&
                       batch >= h->kn_dp_min) ? 6 : 0;
    // which check kernel this call runs (scaldpc_qary_last_timing's info[1])
    const int kernel_id = !h->E ? -1 : (unrolled == 3 && h->kn_dp) ? 8 : unrolled == 3 ? 0 : unrolled == 5 ? 1 : (h->special && dp_nb) ? 7 : (h->special && tree_nb) ? 2 : (wave_mode && h->special) ? 3
                          : wave_mode ? 4 : h->special ? 5 : 6;
    const bool timing = h->kn_timing != 0;
    if (timing) {
        while (h->tev.size() < (size_t)2 * iters + 2) {
            hipEvent_t e;
            SC_HIP(hipEventCreate(&e));
            h->tev.push_back(e);
        }
        SC_HIP(hipEventRecord(h->tev[2 * iters + 1], s));  // start of the call's device work is behind us: into_llr + init
    }
    for (int it = 1; it <= iters; it++) {
        if (timing) SC_HIP(hipEventRecord(h->tev[2 * (it - 1)], s));
        if (h->E) {
#define QUNROLLED(QQ, KK)                                                                                         \
    hipLaunchKernelGGL((k_q_check_unrolled<QQ, KK>), dim3(h->R, Bp / 64), dim3(64), 0, s, h->d_row_ptr, h->d_msg, Bp, \
                       batch, h->d_err)
            if (unrolled == 3 && h->kn_dp) {
                hipLaunchKernelGGL((k_q_check_dp<3, 7>), dim3(h->R, Bp / 64), dim3(64), 0, s, h->d_row_ptr, h->d_msg, Bp, batch, h->d_err);
            } else if (unrolled == 3) {
                QUNROLLED(3, 7);
            } else if (unrolled == 5) {
                QUNROLLED(5, 5);
            }
#undef QUNROLLED
            else if (h->special && dp_nb) {
                if (batch <= h->kn_dp_split)  // (a few dozen codewords: four waves per (check, 64 codewords))
                    hipLaunchKernelGGL((k_q_special_check_dp<5, 6, 4>), dim3(h->R, Bp / 64), dim3(256), 0, s, h->d_row_ptr, h->d_msg,
                                       h->BSUM, h->W, Bp, batch);
                else if (batch <= h->kn_dp_split2)  // (a few hundred: two)
                    hipLaunchKernelGGL((k_q_special_check_dp<5, 6, 2>), dim3(h->R, Bp / 64), dim3(128), 0, s, h->d_row_ptr, h->d_msg,
                                       h->BSUM, h->W, Bp, batch);
                else
                    hipLaunchKernelGGL((k_q_special_check_dp<5, 6, 1>), dim3(h->R, Bp / 64), dim3(64), 0, s, h->d_row_ptr, h->d_msg,
                                       h->BSUM, h->W, Bp, batch);
                if (h->mindc - 1 != dp_nb || h->maxdc - 1 != dp_nb)
                    hipLaunchKernelGGL(k_q_special_check_wave, dim3(h->R, batch), dim3(64), wave_lds, s, h->d_row_ptr, h->d_msg,
                                       h->B, h->BSUM, h->W, Bp, h->maxdc - 1, dp_nb);
            } else if (h->special && tree_nb) {
                // the Kyber shape: tree walk for the rows of 6 coefficient edges, the generic wave kernel for any others
                const size_t tree_lds = ((size_t)tree_nb * h->Q + h->QS + (size_t)(tree_nb * h->Q + h->QS) * 64) * 4;
                hipLaunchKernelGGL((k_q_special_check_tree<5, 6>), dim3(h->R, batch), dim3(64), tree_lds, s, h->d_row_ptr, h->d_msg,
                                   h->BSUM, h->W, Bp);
                if (h->mindc - 1 != tree_nb || h->maxdc - 1 != tree_nb)
                    hipLaunchKernelGGL(k_q_special_check_wave, dim3(h->R, batch), dim3(64), wave_lds, s, h->d_row_ptr, h->d_msg,
                                       h->B, h->BSUM, h->W, Bp, h->maxdc - 1, tree_nb);
            } else if (wave_mode && h->special)
                hipLaunchKernelGGL(k_q_special_check_wave, dim3(h->R, batch), dim3(64), wave_lds, s, h->d_row_ptr, h->d_msg,
                                   h->B, h->BSUM, h->W, Bp, h->maxdc - 1, -1);
            else if (wave_mode)
                hipLaunchKernelGGL(k_q_check_wave, dim3(h->R, batch), dim3(64), wave_lds, s, h->d_row_ptr, h->d_msg, h->Q,
                                   h->B, Bp, h->maxdc, h->d_err);
            else if (h->special)
                hipLaunchKernelGGL(k_q_special_check, dim3(h->R, Bp / T), dim3(T), per_thread * T, s, h->d_row_ptr,
                                   h->d_msg, h->B, h->BSUM, h->W, Bp, batch, h->maxdc - 1);
            else
                if (h->maxdc <= 8)
                    hipLaunchKernelGGL(k_q_check<u64>, dim3(h->R, Bp / T), dim3(T), per_thread * T, s, h->d_row_ptr, h->d_msg,
                                       h->Q, h->B, Bp, batch, h->maxdc, h->d_err);
                else
                    hipLaunchKernelGGL(k_q_check<unsigned __int128>, dim3(h->R, Bp / T), dim3(T), per_thread * T, s, h->d_row_ptr,
                                       h->d_msg, h->Q, h->B, Bp, batch, h->maxdc, h->d_err);
            SC_HIP(hipGetLastError());
        }
        if (timing) SC_HIP(hipEventRecord(h->tev[2 * (it - 1) + 1], s));
#define QVAR_SMALL(QQ)                                                                                              \
    hipLaunchKernelGGL((k_q_var_small<QQ, 4>), dim3(h->N, Bp / 64), dim3(64), 0, s, h->d_col_ptr, h->d_csc_edge, h->d_edge_h, \
                       h->d_llr, h->d_msg, Bp, batch, it == iters ? 1 : 0, h->d_hard)
        const bool vs = !h->special && h->kn_var_small && h->maxdv <= 4;
        if (vs && h->Q == 3)
            QVAR_SMALL(3);
        else if (vs && h->Q == 5)
            QVAR_SMALL(5);
        else if (vs && h->Q == 7)
            QVAR_SMALL(7);
        else if (vs && h->Q == 15)  // (B = 7: the reference's criterion and unit-test decoders)
            QVAR_SMALL(15);
        else if (h->special && h->kn_var_small && h->Q == 5 && h->QS == 25 && h->maxdv <= 4)  // the Kyber SW6 classes (lib.rs:54-75)
            hipLaunchKernelGGL((k_q_var_small_special<5, 4, 25>), dim3(h->N, Bp / 64), dim3(64), 0, s, h->d_col_ptr, h->d_csc_edge,
                               h->d_edge_h, h->d_llr, h->d_msg, BV, h->W, Bp, batch, it == iters ? 1 : 0, h->d_hard);
        else
            hipLaunchKernelGGL(k_q_var, dim3(h->N, Bp / TB), dim3(TB), (size_t)2 * h->W * TB * 4, s, 0, h->d_col_ptr,
                               h->d_csc_edge, h->d_edge_h, h->d_var_q, h->d_var_off, h->d_llr, h->d_msg, h->W, Bp, batch, h->W,
                               it == iters ? 1 : 0, h->d_hard);
#undef QVAR_SMALL
        SC_HIP(hipGetLastError());
    }
    if (timing) SC_HIP(hipEventRecord(h->tev[2 * iters], s));
    signed char *dout = (signed char *)out;
    if (!dev_io) {
        SC_TRY(growq(&h->d_out, &h->cap_out, (size_t)batch * h->N));
        dout = h->d_out;
    }
    hipLaunchKernelGGL(k_q_unpack, dim3((h->N + 255) / 256, batch), dim3(256), 0, s, h->d_hard, h->N, batch, Bp, dout);
    SC_HIP(hipGetLastError());
    u64 status[2] = {0, 0};
    SC_HIP(hipMemcpyAsync(status, h->d_status, sizeof(status), hipMemcpyDeviceToHost, s));
    if (!dev_io) SC_HIP(hipMemcpyAsync(out, dout, (size_t)batch * h->N, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    const int err = (int)(unsigned)status[1];
    const u64 bad = ~status[0];
    if (timing) {
        h->stat_ms_check = h->stat_ms_var = 0.f;
        for (int it = 0; it < iters; it++) {
            float a = 0.f, b = 0.f;
            SC_HIP(hipEventElapsedTime(&a, h->tev[2 * it], h->tev[2 * it + 1]));
            SC_HIP(hipEventElapsedTime(&b, h->tev[2 * it + 1], h->tev[2 * it + 2]));
            h->stat_ms_check += a;
            h->stat_ms_var += b;
        }
        SC_HIP(hipEventElapsedTime(&h->stat_ms_call, h->tev[2 * iters + 1], h->tev[2 * iters]));
        h->stat_iters = iters;
        h->stat_kernel = kernel_id;
        h->stat_batch = batch;
    }
    if (err == QERR_PMF) {
        const int bb = (int)(bad >> 32), vv = (int)((bad & 0x7fffffffull) >> 1) + (((bad >> 31) & 1) ? BV : 0);
        if (bad & 1) return fail(SCALDPC_EPMF, "No maximum probability found (codeword %d, variable %d)", bb, vv);
        return fail(SCALDPC_EPMF, "channel output of codeword %d, variable %d does not sum to 1 +- 1e-3 (decoder.rs:683-684)",
                    bb, vv);
    }
    if (err == QERR_NO_CONFIG)
        return fail(SCALDPC_ENOCONF, "a check node admits no finite configuration (decoder.rs:618)");
    if (err == QERR_NO_FINITE)
        return fail(SCALDPC_ENOCONF, "a message has no finite entry (the reference would not terminate, decoder.rs:368-375)");
    return 0;
}

}  // namespace

extern "C" {

int scaldpc_qary_create(int32_t R, int32_t N, int32_t B, const int8_t *H, int32_t iterations, scaldpc_qary **out)
{
    return qary_build(R, N, B, 0, false, H, iterations, out);
}

int scaldpc_qary_special_create(int32_t R, int32_t N, int32_t B, int32_t BSUM, const int8_t *H, int32_t iterations,
                                scaldpc_qary **out)
{
    return qary_build(R, N, B, BSUM, true, H, iterations, out);
}

int scaldpc_qary_into_llr(const float *pmf, int64_t rows, int32_t Q, uint32_t flags, void *stream, float *llr)
{
    if (!pmf || !llr) return fail(SCALDPC_EINVAL, "NULL argument");
    if (rows <= 0 || Q <= 0) return fail(SCALDPC_EINVAL, "rows and Q must be positive");
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO;
    hipStream_t s = (hipStream_t)stream;  // NULL: the default stream (this call owns no handle)
    const size_t cnt = (size_t)rows * Q;
    float *d_p = nullptr, *d_l = nullptr;
    u64 *d_bad = nullptr;
    auto cleanup = [&]() {
        dev_free(d_bad);
        if (!dev_io) {
            dev_free(d_p);
            dev_free(d_l);
        }
    };
    int rc = dev_alloc(&d_bad, 1);
    if (!rc && !dev_io) {
        rc = dev_alloc(&d_p, cnt);
        if (!rc) rc = dev_alloc(&d_l, cnt);
    }
    if (rc) {
        cleanup();
        return rc;
    }
    u64 bad = ~0ull;
    hipError_t e = hipMemsetAsync(d_bad, 0xFF, sizeof(u64), s);
    if (e == hipSuccess && !dev_io) e = hipMemcpyAsync(d_p, pmf, cnt * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_q_into_llr_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, dev_io ? pmf : d_p, (long)rows, Q,
                           dev_io ? llr : d_l, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !dev_io) e = hipMemcpyAsync(llr, d_l, cnt * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(u64), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return fail(SCALDPC_EHIP, "into_llr failed: %s", hipGetErrorString(e));
    if (bad != ~0ull) {
        if (bad & 1) return fail(SCALDPC_EPMF, "No maximum probability found (row %lld)", (long long)(bad >> 1));
        return fail(SCALDPC_EPMF, "channel output row %lld does not sum to 1 +- 1e-3 (decoder.rs:683-684)", (long long)(bad >> 1));
    }
    return 0;
}

int scaldpc_qary_configure(scaldpc_qary *h, const char *key, const char *value)
{
    if (!h || !key || !value) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!strcmp(key, "wave"))
        h->kn_wave = atoi(value) < 0 ? -1 : atoi(value) != 0;
    else if (!strcmp(key, "unroll"))
        h->kn_unroll = atoi(value) != 0;
    else if (!strcmp(key, "tree"))
        h->kn_tree = atoi(value) != 0;
    else if (!strcmp(key, "dp"))
        h->kn_dp = atoi(value) != 0;
    else if (!strcmp(key, "dp_min"))
        h->kn_dp_min = std::max(1, atoi(value));
    else if (!strcmp(key, "dp_split"))
        h->kn_dp_split = std::max(0, atoi(value));
    else if (!strcmp(key, "dp_split2"))
        h->kn_dp_split2 = std::max(0, atoi(value));
    else if (!strcmp(key, "timing"))
        h->kn_timing = atoi(value) != 0;
    else if (!strcmp(key, "llr_tiled"))
        h->kn_llr_tiled = atoi(value) != 0;
    else if (!strcmp(key, "var_small"))
        h->kn_var_small = atoi(value) != 0;
    else
        return fail(SCALDPC_EINVAL, "unknown knob %s", key);
    return 0;
}

int scaldpc_qary_last_timing(scaldpc_qary *h, float *ms, int32_t *info)
{
    if (!h || !ms || !info) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->stat_kernel < -1 || h->stat_iters == 0) return fail(SCALDPC_EINVAL, "no timed call yet: configure(\"timing\", \"1\") first");
    ms[0] = h->stat_ms_check;
    ms[1] = h->stat_ms_var;
    ms[2] = h->stat_ms_call;
    info[0] = h->stat_iters;
    info[1] = h->stat_kernel;
    info[2] = h->stat_batch;
    info[3] = h->maxdc;
    return 0;
}

int scaldpc_qary_min_sum_batch(scaldpc_qary *h, const float *pmf, int32_t batch, uint32_t flags, void *stream,
                               int8_t *out)
{
    if (h && h->special) return fail(SCALDPC_EINVAL, "this handle is a special decoder: use scaldpc_qary_special_min_sum_batch");
    return qary_run(h, pmf, nullptr, batch, flags, stream, out);
}

int scaldpc_qary_special_min_sum_batch(scaldpc_qary *h, const float *pmf_b, const float *pmf_sum, int32_t batch,
                                       uint32_t flags, void *stream, int8_t *out)
{
    if (h && !h->special) return fail(SCALDPC_EINVAL, "this handle is not a special decoder");
    return qary_run(h, pmf_b, pmf_sum, batch, flags, stream, out);
}

void scaldpc_qary_destroy(scaldpc_qary *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    dev_free(h->d_status);
    dev_free(h->d_row_ptr); dev_free(h->d_col_ptr); dev_free(h->d_csc_edge); dev_free(h->d_edge_var);
    dev_free(h->d_edge_h); dev_free(h->d_var_q); dev_free(h->d_var_off); dev_free(h->d_msg); dev_free(h->d_llr);
    dev_free(h->d_pmf); dev_free(h->d_pmf2); dev_free(h->d_hard); dev_free(h->d_out);
    for (auto &e : h->tev) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

}  // extern "C"

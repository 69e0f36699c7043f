// The check-node kernels of DecoderSpecial (decoder_special.rs:471-617) and the f32::min helpers they share with the
// other q-ary kernels: included by scaldpc_qary.hip (the product) and, as it stands, by
// profiles/microbench/qary_dp_equivalence.hip, which holds the tree-walk and the min-plus kernel to a plain
// enumeration message for message (tests/test_qary_gpu.py::test_special_check_kernels_equal_the_enumeration_bit_for_bit).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>

namespace {

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) < INFINITY; }  // false for inf and NaN

// f32::min / fminf (a NaN operand is ignored) as ONE instruction.  The compiler's lowering of fminf puts a
// canonicalising `v_max_f32 x, x, x` in front of `v_min_f32` for every operand it cannot prove quiet (sNaN
// must come out quiet under IEEE rules): 115 extra instructions per 25 assignments in the enumeration kernels,
// a quarter of their VALU work.  v_min_f32 itself already returns the other operand when one is a quiet NaN
// (the only NaNs arithmetic produces here: inf - inf), which is all f32::min asks for.
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// min of three in ONE instruction (v_min3_f32 = v_min_f32 of v_min_f32: a quiet NaN operand is ignored, as in vmin)
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// minimum of N values, two per v_min3_f32 (min is exact and order-free)
template <int N>
__device__ __forceinline__ float fold_min(const float (&v)[N])
{
    float m = v[0];
#pragma unroll
    for (int i = 1; i + 1 < N; i += 2) m = vmin3(m, v[i], v[i + 1]);
    if constexpr (N % 2 == 0) m = vmin(m, v[N - 1]);
    return m;
}


// Check-node update of DecoderSpecial (decoder_special.rs:506-563): the first k-1 edges
// are B-variables (alphabet QB), the last is the row-sum variable (alphabet QS); ALL
// (2B+1)^(k-1) assignments are visited (SimpleDValueIterator, :226-275), no finiteness
// filter; f32::min semantics (NaN ignored) = fminf.
// LDS: Ab[nbm*QB][T], As[QS][T], Bb[nbm*QB][T], Bs[QS][T].
__global__ void k_q_special_check(const int *__restrict__ row_ptr, float *msg, int B, int BSUM, int W, long Bp,
                                  int batch, int nbm)
{
    extern __shared__ unsigned char smem[];
    const int T = blockDim.x, tid = threadIdx.x;
    const int QB = 2 * B + 1, QS = 2 * BSUM + 1;
    float *Ab = (float *)smem;
    float *As = Ab + (size_t)nbm * QB * T;
    float *Bb = As + (size_t)QS * T;
    float *Bs = Bb + (size_t)nbm * QB * T;
    const int c = blockIdx.x;
    const long b = (long)blockIdx.y * T + tid;
    if (b >= batch) return;
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0, nb = k - 1;
    for (int j = 0; j < nb; j++)
        for (int q = 0; q < QB; q++) {
            Ab[(size_t)(j * QB + q) * T + tid] = msg[((size_t)(e0 + j) * W + q) * Bp + b];
            Bb[(size_t)(j * QB + q) * T + tid] = INFINITY;
        }
    for (int q = 0; q < QS; q++) {
        As[(size_t)q * T + tid] = msg[((size_t)(e0 + nb) * W + q) * Bp + b];
        Bs[(size_t)q * T + tid] = INFINITY;
    }
    u64 dq = 0;  // digit j = d_j + B, 8 bits each, all start at 0 (= -B)
    for (;;) {
        int dsum = 0;
        float S = 0.0f;
        for (int j = 0; j < nb; j++) {
            const int q = (int)(dq >> (8 * j)) & 255;
            dsum += q - B;
            S += Ab[(size_t)(j * QB + q) * T + tid];
        }
        const size_t os = (size_t)(-dsum + BSUM) * T + tid;
        S += As[os];
        for (int j = 0; j < nb; j++) {
            const int q = (int)(dq >> (8 * j)) & 255;
            const size_t o = (size_t)(j * QB + q) * T + tid;
            Bb[o] = fminf(Bb[o], S - Ab[o]);
        }
        Bs[os] = fminf(Bs[os], S - As[os]);
        int j = 0;
        for (; j < nb; j++) {
            const int q = (int)(dq >> (8 * j)) & 255;
            if (q < 2 * B) {
                dq += 1ull << (8 * j);
                break;
            }
            dq &= ~(255ull << (8 * j));
        }
        if (j >= nb) break;
    }
    for (int j = 0; j < nb; j++)
        for (int q = 0; q < QB; q++) msg[((size_t)(e0 + j) * W + q) * Bp + b] = Bb[(size_t)(j * QB + q) * T + tid];
    for (int q = 0; q < QS; q++) msg[((size_t)(e0 + nb) * W + q) * Bp + b] = Bs[(size_t)q * T + tid];
}

// ---------------------------------------------------------------------------
// DecoderSpecial check update for rows of NB coefficient edges over an alphabet of QB symbols --
// the Kyber decoders' shape (lib.rs:54-75: B = 2 => QB = 5, SW = 6 => NB = 6: 5^6 = 15 625
// assignments per check) -- as a TREE walk with everything but the commits in registers.
//
// What the reference computes (decoder_special.rs:506-563), per assignment d_0..d_{NB-1}:
//     S = ((((0 + a_0[d_0]) + a_1[d_1]) + ...) + a_{NB-1}[d_{NB-1}]) + a_s[-sum d]        f32, this order
//     beta_j[d_j] = min(beta_j[d_j], S - a_j[d_j])  for every edge j,  beta_s[-sum d] likewise
// i.e. NB + 1 additions, NB + 1 subtractions, NB + 1 minima = 3 (NB + 1) = 21 f32 operations per
// assignment (328 125 per check and iteration).  min is exact and order-free; the sum is not.  This kernel
// ENUMERATES the assignments (the recursion that avoids it, exact because it keeps the order of the additions, is
// k_q_special_check_dp below: the default from a handful of codewords on), but not independently of each other:
//   * the partial sums of a common prefix are shared (the same additions in the same order, fewer of
//     them): a lane owns the first NB - 3 digits (its work items) and unrolls the last three, so an
//     assignment costs 2 additions instead of NB + 1;
//   * the lane's own digits keep their running minima in per-lane LDS tables (dynamic indices), read and
//     committed once per work item;
//   * the three unrolled digits index their alpha / minima with compile-time indices: registers for the
//     whole kernel;
//   * the row-sum symbol of an assignment is T0 - (d_{NB-3} + d_{NB-2} + d_{NB-1}): inside a work item it
//     moves through a window of 3 QB - 2 neighbouring symbols, which is loaded to / committed from
//     registers once per item.
// Round 4: MIN-MARGINALS OF S (see QEnum above): x -> fl(x - a) is monotone, so
//     beta_j[d] = min over assignments with d_j = d of fl(S - a_j[d]) = fl( (min over them of S) - a_j[d] )   bit for bit,
// and the walk only folds sums: the unrolled block of QB d5-values lowers the two minima that belong to ITS digits
// (d5's, the row-sum symbol's) with each S and hands ONE folded minimum up to d4's, d3's and the lane's digits.
// Per assignment: 2 adds + 2 mins + (2 v_min3 + 1 v_min) / QB = ~4.7 VALU operations with register operands instead of
// the 16 of the subtract-per-candidate form (the reference's own count: 21; the generic wave kernel above: ~20 LDS
// accesses and ~100 integer / address operations); the NB * QB + QS subtractions happen once per ROW, at the commit.
// wave = (check, codeword); LDS: Ab[NB*QB] + As[QS] floats (shared), per-lane tables
// Bb[NB*QB][64] and Bs[QS][64]; the partial minima of the 64 lanes are combined at the end by a
// transposed walk over the tables (exact), so the messages are bit-identical to the other kernels'.
// grid (R, batch), block 64.  Rows whose degree is not NB + 1 are left to k_q_special_check_wave.
// ---------------------------------------------------------------------------
template <int QB, int NB>
__global__ __launch_bounds__(64) void k_q_special_check_tree(const int *__restrict__ row_ptr, float *msg, int BSUM, int W,
                                                             long Bp)
{
    static_assert(NB >= 3, "needs at least three coefficient edges");
    constexpr int B = (QB - 1) / 2, NL = NB - 3, WIN = 3 * QB - 2;
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x;
    const int QS = 2 * BSUM + 1;
    float *Ab = (float *)smem;               // [NB][QB]
    float *As = Ab + NB * QB;                // [QS]
    float *Bb = As + QS;                     // [NB * QB][64]   per-lane tables of the digits (the unrolled digits' rows only at the end)
    float *Bs = Bb + (size_t)NB * QB * 64;   // [QS][64]
    const int c = blockIdx.x;
    const long b = blockIdx.y;
    const int e0 = row_ptr[c], nb = row_ptr[c + 1] - e0 - 1;
    if (nb != NB) return;
    for (int i = lane; i < NB * QB; i += 64) Ab[i] = msg[((size_t)(e0 + i / QB) * W + i % QB) * Bp + b];
    for (int i = lane; i < QS; i += 64) As[i] = msg[((size_t)(e0 + NB) * W + i) * Bp + b];
    for (int i = 0; i < NL * QB; i++) Bb[(size_t)i * 64 + lane] = INFINITY;  // (the unrolled digits' rows are written at the end)
    for (int i = 0; i < QS; i++) Bs[(size_t)i * 64 + lane] = INFINITY;
    __syncthreads();
    // digits NB-3 .. NB-1 are unrolled (compile-time indices): their alphas and running minima are registers for the whole kernel
    float A3[QB], A4[QB], A5[QB], b3[QB], b4[QB], b5[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) {
        A3[q] = Ab[NL * QB + q];
        A4[q] = Ab[(NB - 2) * QB + q];
        A5[q] = Ab[(NB - 1) * QB + q];
        b3[q] = INFINITY;
        b4[q] = INFINITY;
        b5[q] = INFINITY;
    }
    int items = 1;
#pragma unroll
    for (int j = 0; j < NL; j++) items *= QB;
    for (int t = lane; t < items; t += 64) {
        int dg[NL > 0 ? NL : 1];
        float ml[NL > 0 ? NL : 1];
        float P = 0.0f, gl = INFINITY;  // gl: minimum of S over this work item (all assignments with the lane's digits)
        int dsum = 0, tt = t;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            dg[j] = tt % QB;
            tt /= QB;
            ml[j] = Bb[(size_t)(j * QB + dg[j]) * 64 + lane];  // continue from the lane's table entry
            P += Ab[j * QB + dg[j]];  // ((0 + a_0) + a_1) + ...
            dsum += dg[j] - B;
        }
        // Row-sum symbol of (.., d3, d4, d5): BSUM - (dsum + (d3-B) + (d4-B) + (d5-B)) = T0 - (d3 + d4 + d5): over the whole
        // work item it moves through a window of 3 QB - 2 neighbouring symbols.  Their alphas and the lane's running minima
        // are loaded ONCE per item and committed once (round 4: the window used to be re-loaded and committed for every d3 --
        // 29 LDS operations per 25 assignments, which bound the kernel once the arithmetic had shrunk to ~5 operations per
        // assignment).  The minima START from the lane's table entries, so the commit is a plain store.
        const int T0 = BSUM - dsum + 3 * B;
        float *const ps0 = &Bs[(size_t)T0 * 64 + lane];
        float aw[WIN], mw[WIN];
#pragma unroll
        for (int u = 0; u < WIN; u++) {
            aw[u] = As[T0 - u];
            mw[u] = ps0[-(ptrdiff_t)u * 64];
        }
        // Every S (built left to right, the reference's additions in the reference's order) lowers the minimum of its own d5
        // and of its row-sum symbol; the QB sums of one d4 are folded (two per v_min3_f32) into ONE number for d4's minimum,
        // the QB of those into one for d3's, and those into one for the lane's digits.
#pragma unroll
        for (int d3 = 0; d3 < QB; d3++) {
            const float P3 = P + A3[d3];
            float g4[QB];
#pragma unroll
            for (int d4 = 0; d4 < QB; d4++) {
                const float P4 = P3 + A4[d4];
                float Sv[QB];
#pragma unroll
                for (int d5 = 0; d5 < QB; d5++) {
                    Sv[d5] = (P4 + A5[d5]) + aw[d3 + d4 + d5];
                    b5[d5] = vmin(b5[d5], Sv[d5]);
                    mw[d3 + d4 + d5] = vmin(mw[d3 + d4 + d5], Sv[d5]);
                }
                g4[d4] = fold_min(Sv);
                b4[d4] = vmin(b4[d4], g4[d4]);
            }
            const float g3 = fold_min(g4);
            b3[d3] = vmin(b3[d3], g3);
            gl = vmin(gl, g3);
        }
#pragma unroll
        for (int u = 0; u < WIN; u++) ps0[-(ptrdiff_t)u * 64] = mw[u];
#pragma unroll
        for (int j = 0; j < NL; j++) Bb[(size_t)(j * QB + dg[j]) * 64 + lane] = vmin(ml[j], gl);
    }
    // Combine the 64 lanes' partial minima of S (exact: min is order-free), subtract the slot's alpha ONCE and write c2v in
    // place (a minimum that stayed +inf -- no assignment through the slot -- gives +inf, not inf - inf).  The tables
    // are [slot][lane] in LDS: lane s takes slot s and walks the 64 entries of its row -- rotated by its own
    // index, so that the lanes of a wave hit 64 different banks -- instead of a butterfly of 6 cross-lane
    // shuffles per slot (each a dependent LDS-crossbar round trip: 480 of them per wave were half a wave's
    // life, profiles/r02/sq_counters_kyber_tree.json).  The three unrolled digits' register minima go through
    // the table as well (rows NB-3 .. NB-1 of Bb).  (A two-phase combine over a table without those rows -- 10.2 KB of
    // LDS per wave instead of 14.3, 15 waves per CU instead of 11 -- was measured and is SLOWER: 1.97 -> 2.08 ms per
    // 256-codeword call, issue stalls 26 % -> 40 % of the wave cycles; profiles/r04/qary_min_marginals.log.)
#pragma unroll
    for (int q = 0; q < QB; q++) {
        Bb[(size_t)(NL * QB + q) * 64 + lane] = b3[q];
        Bb[(size_t)((NB - 2) * QB + q) * 64 + lane] = b4[q];
        Bb[(size_t)((NB - 1) * QB + q) * 64 + lane] = b5[q];
    }
    __syncthreads();
    const int nslots = NB * QB + QS;  // Bb and Bs are contiguous: one table of nslots rows
    for (int s = lane; s < nslots; s += 64) {
        const float *row = Bb + (size_t)s * 64;
        float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
#pragma unroll 4
        for (int l = 0; l < 64; l += 4) {
            m0 = vmin(m0, row[(l + lane) & 63]);
            m1 = vmin(m1, row[(l + 1 + lane) & 63]);
            m2 = vmin(m2, row[(l + 2 + lane) & 63]);
            m3 = vmin(m3, row[(l + 3 + lane) & 63]);
        }
        const float mS = vmin(vmin(m0, m1), vmin(m2, m3));
        const float v = finite_f(mS) ? mS - Ab[s] : INFINITY;  // (Ab and As are contiguous: slot s's alpha is Ab[s])
        if (s < NB * QB)
            msg[((size_t)(e0 + s / QB) * W + s % QB) * Bp + b] = v;
        else
            msg[((size_t)(e0 + NB) * W + (s - NB * QB)) * Bp + b] = v;
    }
}

// ---------------------------------------------------------------------------
// DecoderSpecial check update WITHOUT enumerating the assignments: a min-plus recursion over the edges, IN THE
// REFERENCE'S ORDER OF ADDITIONS, bit-identical to the enumeration (round 4).
//
// The reference forms, per assignment (decoder_special.rs:531-554),
//     S = ((((0 + a_0[d_0]) + a_1[d_1]) + ...) + a_{NB-1}[d_{NB-1}]) + a_s[-sum d]                  (f32, this order)
// and the min-marginal form above needs  M_j[d] = min over the assignments with d_j = d of S.  S is a chain of
// x -> fl(x + c) steps, each monotone non-decreasing in x.  Take two assignments that agree from edge k + 1 on and
// have the same sum of their first k + 1 digits (so the same row-sum symbol): the one with the smaller partial sum
// P_k = fl(..fl(a_0 + a_1).. + a_k) has the smaller (or equal) S, through the SAME later additions.  Hence
//     min over a class of prefixes of S = S continued from the class's minimal P_k                  -- bit for bit,
// the minimal partial sums obey  P_k[u] = min over q of fl(P_{k-1}[u - q] + a_k[q])  (u = digit sum so far), and
// the same holds with one digit pinned: start from V[u] = fl(P_{j-1}[u] + a_j[d]) and continue over k = j + 1 ..
// (u counts the OTHER edges' digits, which makes the recursion the same code for every d; d only enters through the
// scalar a_j[d] and through the row-sum symbol of the last step, BSUM + NB B - (u + d)).  NaN alphas (inf - inf of
// the variable update) drop out exactly as under f32::min: v_min ignores a NaN operand, and a class that has nothing
// else stays NaN, which the commit below turns into +inf like a class of +inf sums (the reference's beta stays at
// its initial +inf in both cases).
// Work per check: NB QB pinned recursions of at most NB - 1 steps over at most (QB-1)(NB-1)+1 entries -- about 5 300
// additions and 2 200 v_min3 at the Kyber shape (QB = 5, NB = 6) against 15 625 assignments x ~4.7 operations in the
// tree walk, and nothing leaves the registers: lane = codeword, block = 64 codewords of one check, no LDS.
// Rows whose degree is not NB + 1 are left to k_q_special_check_wave.
// ---------------------------------------------------------------------------
// out[u] = min over q of (in[u - q] + ak[q]): one min-plus step, candidates two per v_min3_f32
template <int LEN, int QB>
__device__ __forceinline__ void minplus_step(const float (&in)[LEN], const float (&ak)[QB], float (&out)[LEN + QB - 1])
{
#pragma unroll
    for (int u = 0; u < LEN + QB - 1; u++) {
        float m = 0.0f, pend = 0.0f;
        bool have = false, hp = false;  // (compile-time after unrolling)
#pragma unroll
        for (int q = 0; q < QB; q++) {
            if (u - q < 0 || u - q >= LEN) continue;
            const float c = in[u - q] + ak[q];
            if (!have) {
                m = c;
                have = true;
            } else if (!hp) {
                pend = c;
                hp = true;
            } else {
                m = vmin3(m, pend, c);
                hp = false;
            }
        }
        out[u] = hp ? vmin(m, pend) : m;
    }
}

// min over u of (V[u] + asw[u + D])
template <int LEN, int D, int WN>
__device__ __forceinline__ float minplus_close(const float (&V)[LEN], const float (&asw)[WN])
{
    static_assert(LEN + D <= WN, "window");
    float c[LEN];
#pragma unroll
    for (int u = 0; u < LEN; u++) c[u] = V[u] + asw[u + D];
    return fold_min(c);
}

// the steps k = K .. NB-1 of a pinned recursion, then the row-sum symbol's alpha: returns M_j[d]
template <int QB, int NB, int K, int LEN>
struct DpTail {
    static __device__ __forceinline__ float run(const float (&V)[LEN], const float (&a)[NB][QB],
                                                const float (&asw)[(QB - 1) * NB + 1], int d)
    {
        if constexpr (K < NB) {
            float Vn[LEN + QB - 1];
            minplus_step<LEN, QB>(V, a[K], Vn);
            return DpTail<QB, NB, K + 1, LEN + QB - 1>::run(Vn, a, asw, d);
        } else {
            constexpr int WN = (QB - 1) * NB + 1;
            static_assert(QB == 5, "the close-out below is written for five symbols");
            switch (d) {  // (wave-uniform; the pinned symbol shifts the window of row-sum alphas)
                case 0: return minplus_close<LEN, 0, WN>(V, asw);
                case 1: return minplus_close<LEN, 1, WN>(V, asw);
                case 2: return minplus_close<LEN, 2, WN>(V, asw);
                case 3: return minplus_close<LEN, 3, WN>(V, asw);
                default: return minplus_close<LEN, 4, WN>(V, asw);
            }
        }
    }
};

// edge J: its QB pinned recursions if J is in JSET (P = the minimal partial sums over the edges before it, by digit sum), then
// on to J + 1 while JSET has anything beyond (bit NB of JSET = the row-sum variable's own messages)
template <int QB, int NB, int J, int LEN, unsigned JSET>
struct DpEdge {
    static __device__ __forceinline__ void run(const float (&P)[LEN], const float (&a)[NB][QB],
                                               const float (&asw)[(QB - 1) * NB + 1], float *edge0, size_t qstride, size_t estride,
                                               float *sum_top, bool store)
    {
        static_assert(QB == 5, "the symbol select below is written for five symbols");
        if constexpr ((JSET >> J) & 1u) {
#pragma unroll 1
            for (int d = 0; d < QB; d++) {
                const float ajd = d == 0 ? a[J][0] : d == 1 ? a[J][1] : d == 2 ? a[J][2] : d == 3 ? a[J][3] : a[J][4];
                float V[LEN];
#pragma unroll
                for (int u = 0; u < LEN; u++) V[u] = P[u] + ajd;
                const float M = DpTail<QB, NB, J + 1, LEN>::run(V, a, asw, d);
                if (store) edge0[(size_t)J * estride + (size_t)d * qstride] = finite_f(M) ? M - ajd : INFINITY;
            }
        }
        if constexpr ((JSET >> (J + 1)) != 0u) {
            float Pn[LEN + QB - 1];
            minplus_step<LEN, QB>(P, a[J], Pn);
            if constexpr (J + 1 < NB)
                DpEdge<QB, NB, J + 1, LEN + QB - 1, JSET>::run(Pn, a, asw, edge0, qstride, estride, sum_top, store);
            else {
                // the row-sum variable's own messages: symbol BSUM + NB B - w is reached by the assignments of digit sum w alone
#pragma unroll
                for (int w = 0; w < LEN + QB - 1; w++) {
                    const float M = Pn[w] + asw[w];
                    if (store) *(sum_top - (ptrdiff_t)w * (ptrdiff_t)qstride) = finite_f(M) ? M - asw[w] : INFINITY;
                }
            }
        }
    }
};

// PARTS = 1: grid (R, Bp / 64), block 64 -- a lane does the whole row of its codeword.
// PARTS = 2: block 128 -- two waves share a (check, 64 codewords): edges {0, 3, 4} and {1, 2, 5, row-sum}.
// PARTS = 4: grid (R, Bp / 64), block 256 -- the four waves of a block share ONE (check, 64 codewords) and split its
// edges: wave 0 pins edge 0, wave 1 edge 1, wave 2 edges 2 and 5 and writes the row-sum variable's messages, wave 3 edges 3
// and 4 (about a quarter of the additions each; a wave's prefix recursion up to its first edge is cheap next to the pinned
// ones).  Four times the waves of a quarter the length: what a call of up to 64 codewords needs to fill the chip (the
// launch is then 12 us, close to the empty-launch floor of this loop); halves are best from 65 to ~192 codewords, and from
// there on the whole-row form has waves enough and less redundant work (profiles/r04/kyber_form_sweep.log).  The messages are updated in place, so with
// PARTS > 1 every wave loads the whole row before the block's barrier and stores after it.
template <int QB, int NB, int PARTS>
__global__ __launch_bounds__(64 * PARTS) void k_q_special_check_dp(const int *__restrict__ row_ptr, float *msg, int BSUM, int W,
                                                                          long Bp, int batch)
{
    static_assert(NB == 6, "the split of the edges over the four waves below is written for six coefficient edges");
    constexpr int B = (QB - 1) / 2, WN = (QB - 1) * NB + 1;
    const int c = blockIdx.x, part = threadIdx.x >> 6;
    const long b = (long)blockIdx.y * 64 + (threadIdx.x & 63);  // (< Bp: the padding lanes compute on whatever is there and store nothing)
    const int e0 = row_ptr[c], nb = row_ptr[c + 1] - e0 - 1;
    if (nb != NB) return;  // (the whole block)
    float a[NB][QB], asw[WN];
#pragma unroll
    for (int j = 0; j < NB; j++)
#pragma unroll
        for (int q = 0; q < QB; q++) a[j][q] = msg[((size_t)(e0 + j) * W + q) * Bp + b];
    // asw[w] = alpha of the row-sum symbol that closes an assignment of digit sum w (digits q = d + B): BSUM + NB B - w
    const int top = BSUM + NB * B;  // (<= 2 BSUM: NB B <= BSUM is checked when the decoder is built, decoder_special.rs:388-392)
    float *const sum_top = msg + ((size_t)(e0 + NB) * W + top) * Bp + b;
#pragma unroll
    for (int w = 0; w < WN; w++) asw[w] = *(sum_top - (ptrdiff_t)w * (ptrdiff_t)Bp);
    if constexpr (PARTS > 1) __syncthreads();
    const bool store = b < batch;
    float *const edge0 = msg + (size_t)e0 * W * Bp + b;
    const float P0[1] = {0.0f};  // S starts at 0 (decoder_special.rs:536)
    if (store && part == PARTS / 2) {  // (the wave that writes the row-sum variable's messages)
        // row-sum symbols no assignment reaches keep the reference's initial +inf (decoder_special.rs:527)
        for (int t = 0; t < top - (WN - 1); t++) msg[((size_t)(e0 + NB) * W + t) * Bp + b] = INFINITY;
        for (int t = top + 1; t < 2 * BSUM + 1; t++) msg[((size_t)(e0 + NB) * W + t) * Bp + b] = INFINITY;
    }
    static_assert(PARTS == 1 || PARTS == 2 || PARTS == 4, "whole row, halves or quarters");
    if constexpr (PARTS == 1)
        DpEdge<QB, NB, 0, 1, 0x7Fu>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);
    else if constexpr (PARTS == 2) {
        if (part == 0)
            DpEdge<QB, NB, 0, 1, 0x19u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);  // edges 0, 3, 4
        else
            DpEdge<QB, NB, 0, 1, 0x66u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);  // edges 1, 2, 5, row-sum
    } else if (part == 0)
        DpEdge<QB, NB, 0, 1, 0x01u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);
    else if (part == 1)
        DpEdge<QB, NB, 0, 1, 0x02u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);
    else if (part == 2)
        DpEdge<QB, NB, 0, 1, 0x64u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);
    else
        DpEdge<QB, NB, 0, 1, 0x18u>::run(P0, a, asw, edge0, (size_t)Bp, (size_t)W * Bp, sum_top, store);
}

}  // namespace

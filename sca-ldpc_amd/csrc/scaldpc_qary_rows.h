// The register-resident check-node kernels of Decoder (decoder.rs:585-631) for small alphabets: the unrolled enumeration
// and the min-plus recursion that replaces it for Q = 3.  Included by scaldpc_qary.hip (the product) and, as it stands, by
// profiles/microbench/qary_dp_equivalence.hip, which holds both to a plain enumeration in the reference's form message for
// message.
#pragma once
#include "scaldpc_qary_special.h"

#include <utility>

namespace {
constexpr int QERR_NO_FINITE = 5;  // decoder.rs:368-375 would spin forever
constexpr int QERR_NO_CONFIG = 6;  // decoder.rs:618 assert

// ---------------------------------------------------------------------------
// Unrolled enumeration for small alphabets (the reference's own decoder sizes: Q = 3,
// DC <= 7; also Q = 5, DC <= 5).  Every digit is a template argument, so alpha / beta live in
// registers with compile-time indices -- no LDS, no index arithmetic.  S is built left to
// right through the recursion (the partial sum of the first j digits is shared by all
// assignments below it: the same additions in the same order as decoder.rs:600-610, fewer of
// them).  No finite-support filter is needed: an assignment through a non-finite alpha has
// S = inf or NaN, and v_min never lets those lower a minimum -- exactly the assignments
// FiniteDValueIterator / `cfg.sum.is_finite()` (decoder.rs:281-401, 612) would have skipped.
//
// MIN-MARGINALS OF S, ONE SUBTRACTION PER OUTPUT (round 4).  The reference computes, for every edge j and symbol d,
//     beta_j[d] = min over the assignments with d_j = d of fl(S - alpha_j[d])          (decoder.rs:621-627)
// with alpha_j[d] the SAME number in every candidate.  x -> fl(x - a) is monotone non-decreasing (exact subtraction is,
// and so is rounding), so the minimum of the candidates is the candidate of the minimum:
//     beta_j[d] = fl( M_j[d] - alpha_j[d] ),   M_j[d] = min over those assignments of S      -- bit for bit.
// (A minimum that stays +inf -- no assignment with a finite sum through (j, d) -- gives beta = +inf without forming
// inf - inf; a finite M implies a finite alpha_j[d], which is one of its summands.)  The enumeration therefore only
// folds sums: every node of the digit tree returns the minimum of S over its subtree, the node for digit q of edge J
// lowers M[J][q] with it, and the K subtractions per assignment (+ K minima) of the reference's form -- 7 x 729 each
// for config 4's checks, 3956 v_sub and 2973 v_min per row -- become Q x K subtractions per ROW and about three
// minima per assignment.  The oracle keeps the reference's form; every q-ary parity test holds this one to it.
// lane = codeword, thread = (check, codeword).
// ---------------------------------------------------------------------------
template <int Q, int K, int J, int... D>
struct QEnum {
    // S: the left-to-right sum of alpha over the digits D... chosen so far; returns min S over the subtree
    static __device__ __forceinline__ float run(const float (&A)[K][Q], float (&M)[K][Q], float S)
    {
        return run_q(A, M, S, std::make_integer_sequence<int, Q>());
    }
    template <int... Qs>
    static __device__ __forceinline__ float run_q(const float (&A)[K][Q], float (&M)[K][Q], float S, std::integer_sequence<int, Qs...>)
    {
        const float sub[Q] = {QEnum<Q, K, J + 1, D..., Qs>::run(A, M, S + A[J][Qs])...};
#pragma unroll
        for (int q = 0; q < Q; q++) M[J][q] = vmin(M[J][q], sub[q]);
        return fold_min(sub);
    }
};
// Last FREE digit (edge K-2; edge K-1's digit then follows from sum d = 0).
template <int Q, int K, int... D>
struct QEnum<Q, K, K - 2, D...> {
    static_assert(sizeof...(D) == K - 2, "digits of the edges before the last free one");
    static constexpr int B = (Q - 1) / 2;
    static constexpr int base = -((D - B) + ... + 0);
    static constexpr int dl(int q) { return base - (q - B); }           // digit of the last edge when edge K-2 takes q
    static constexpr bool ok(int q) { return dl(q) >= -B && dl(q) <= B; }
    static constexpr int nvalid()
    {
        int n = 0;
        for (int q = 0; q < Q; q++) n += ok(q) ? 1 : 0;
        return n;
    }
    static constexpr int slot(int q)  // valid q -> its slot among the valid ones
    {
        int n = 0;
        for (int t = 0; t < q; t++) n += ok(t) ? 1 : 0;
        return n;
    }
    static __device__ __forceinline__ float run(const float (&A)[K][Q], float (&M)[K][Q], float S)
    {
        constexpr int NV = nvalid();
        if constexpr (NV > 0) {
            float S2[NV];
            fill(A, M, S, S2, std::make_integer_sequence<int, Q>());
            return fold_min(S2);
        }
        return INFINITY;
    }
    template <int... Qs>
    static __device__ __forceinline__ void fill(const float (&A)[K][Q], float (&M)[K][Q], float S, float (&S2)[nvalid()], std::integer_sequence<int, Qs...>)
    {
        (one<Qs>(A, M, S, S2), ...);
    }
    template <int q>
    static __device__ __forceinline__ void one(const float (&A)[K][Q], float (&M)[K][Q], float S, float (&S2)[nvalid()])
    {
        if constexpr (ok(q)) {
            constexpr int ql = dl(q) + B;
            const float s2 = (S + A[K - 2][q]) + A[K - 1][ql];  // the reference's additions in the reference's order
            S2[slot(q)] = s2;
            M[K - 2][q] = vmin(M[K - 2][q], s2);
            M[K - 1][ql] = vmin(M[K - 1][ql], s2);
        }
    }
};
// last edge: its digit is fixed by sum d = 0 (reached directly only when K = 1)
template <int Q, int K, int... D>
struct QEnum<Q, K, K - 1, D...> {
    static constexpr int B = (Q - 1) / 2;
    static constexpr int dl = -((D - B) + ... + 0);
    static __device__ __forceinline__ float run(const float (&A)[K][Q], float (&M)[K][Q], float S)
    {
        if constexpr (dl >= -B && dl <= B) {
            constexpr int ql = dl + B;
            const float S2 = S + A[K - 1][ql];
            M[K - 1][ql] = vmin(M[K - 1][ql], S2);
            return S2;
        }
        return INFINITY;
    }
};

template <int Q, int K>
__device__ __forceinline__ void q_check_unrolled(float *msg, int e0, long Bp, long b, int *err)
{
    float A[K][Q], M[K][Q];
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            A[j][q] = msg[((size_t)(e0 + j) * Q + q) * Bp + b];
            M[j][q] = INFINITY;
        }
    QEnum<Q, K, 0>::run(A, M, 0.0f);
    // "at least one configuration" (decoder.rs:618 asserts it): an assignment with a finite sum makes M[0][d_0] finite, and
    // nothing else does
    bool any_conf = false;
#pragma unroll
    for (int q = 0; q < Q; q++) any_conf |= finite_f(M[0][q]);
    if (!any_conf) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < Q; q++) any |= finite_f(A[j][q]);
            bad |= !any;
        }
        atomicMax(err, bad ? QERR_NO_FINITE : QERR_NO_CONFIG);
    }
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int q = 0; q < Q; q++)
            msg[((size_t)(e0 + j) * Q + q) * Bp + b] = finite_f(M[j][q]) ? M[j][q] - A[j][q] : INFINITY;  // (see above: one subtraction per output)
}

// ---------------------------------------------------------------------------
// The same check update WITHOUT the enumeration: the min-plus recursion of k_q_special_check_dp (scaldpc_qary_special.h,
// where the argument is written out) for Decoder's constraint.  The K - 1 free digits q_j = d_j + B are summed left to
// right and the last edge's symbol follows from sum d = 0: q_last = K B - U, U = the free digits' sum -- an assignment
// exists only for U in [(K - 2) B, K B].  Every table below is therefore clipped, at compile time, to the digit sums that
// can still reach that window with the edges that are left (and that are not past it already): for config 4's checks
// (Q = 3, K = 7) ~700 additions per row instead of 729 assignments x ~6 operations.
//   P_k[u]   minimal partial sum over the assignments of edges 0 .. k-1 with digit sum u            (prefix)
//   V_k[u]   the same with edge J pinned to symbol D; u counts the other edges' digits               (pinned, k > J)
//   M_J[D]   = min over u of fl(V_{K-1}[u] + a_last[K B - u - D]);  beta_J[D] = fl(M - a_J[D])
// Non-finite alphas need no filter (see k_q_check_unrolled); the reference's "no configuration" assert reads off the last
// edge's minima (every assignment passes through one of them).
// ---------------------------------------------------------------------------
template <int Q, int K>
struct DpRange {
    static constexpr int B = (Q - 1) / 2, NB = K - 1, S = Q - 1, TL = (K - 2) * B, TH = K * B;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int cmin(int a, int b) { return a < b ? a : b; }
    // prefix over k edges
    static constexpr int plo(int k) { return cmax(0, TL - (NB - k) * S); }
    static constexpr int phi(int k) { return cmin(k * S, TH); }
    // pinned to symbol d, k edges done (the pinned one among them): digit sum of the other k - 1
    static constexpr int vlo(int k, int d) { return cmax(0, TL - d - (NB - k) * S); }
    static constexpr int vhi(int k, int d) { return cmin((k - 1) * S, TH - d); }
};

// out[u] = min over q of (in[u - q] + ak[q]) for u in [LO, HO], in covering [LI, HI]
template <int Q, int LI, int HI, int LO, int HO>
__device__ __forceinline__ void minplus_step_clipped(const float (&in)[HI - LI + 1], const float (&ak)[Q], float (&out)[HO - LO + 1])
{
#pragma unroll
    for (int u = LO; u <= HO; u++) {
        float m = INFINITY, pend = 0.0f;
        bool have = false, hp = false;  // (compile-time after unrolling)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (u - q < LI || u - q > HI) continue;
            const float c = in[u - q - LI] + ak[q];
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
        out[u - LO] = hp ? vmin(m, pend) : m;
    }
}

// edge J pinned to D, KK edges done: the remaining free edges, then the last edge's alpha -> M_J[D]
template <int Q, int K, int D, int KK>
struct GDpTail {
    using R = DpRange<Q, K>;
    static __device__ __forceinline__ float run(const float (&V)[R::vhi(KK, D) - R::vlo(KK, D) + 1], const float (&a)[K][Q])
    {
        constexpr int LI = R::vlo(KK, D), HI = R::vhi(KK, D);
        if constexpr (KK < R::NB) {
            constexpr int LO = R::vlo(KK + 1, D), HO = R::vhi(KK + 1, D);
            if constexpr (HO < LO)
                return INFINITY;
            else {
                float Vn[HO - LO + 1];
                minplus_step_clipped<Q, LI, HI, LO, HO>(V, a[KK], Vn);
                return GDpTail<Q, K, D, KK + 1>::run(Vn, a);
            }
        } else {
            float c[HI - LI + 1];
#pragma unroll
            for (int u = LI; u <= HI; u++) c[u - LI] = V[u - LI] + a[K - 1][K * R::B - u - D];
            return fold_min(c);
        }
    }
};

template <int Q, int K, int J>
struct GDpEdge {
    using R = DpRange<Q, K>;
    template <int D>
    static __device__ __forceinline__ void pinned(const float (&P)[R::phi(J) - R::plo(J) + 1], const float (&a)[K][Q], float *edge0,
                                                  size_t qstride)
    {
        constexpr int LO = R::vlo(J + 1, D), HO = R::vhi(J + 1, D);
        float M = INFINITY;
        if constexpr (HO >= LO) {
            float V[HO - LO + 1];
#pragma unroll
            for (int u = LO; u <= HO; u++) V[u - LO] = P[u - R::plo(J)] + a[J][D];
            M = GDpTail<Q, K, D, J + 1>::run(V, a);
        }
        edge0[((size_t)J * Q + D) * qstride] = finite_f(M) ? M - a[J][D] : INFINITY;
    }
    template <int... Ds>
    static __device__ __forceinline__ void all_pinned(const float (&P)[R::phi(J) - R::plo(J) + 1], const float (&a)[K][Q], float *edge0,
                                                      size_t qstride, std::integer_sequence<int, Ds...>)
    {
        (pinned<Ds>(P, a, edge0, qstride), ...);
    }
    // returns whether the check has a configuration with a finite sum
    static __device__ __forceinline__ bool run(const float (&P)[R::phi(J) - R::plo(J) + 1], const float (&a)[K][Q], float *edge0,
                                               size_t qstride)
    {
        if constexpr (J < R::NB) {
            all_pinned(P, a, edge0, qstride, std::make_integer_sequence<int, Q>());
            float Pn[R::phi(J + 1) - R::plo(J + 1) + 1];
            minplus_step_clipped<Q, R::plo(J), R::phi(J), R::plo(J + 1), R::phi(J + 1)>(P, a[J], Pn);
            return GDpEdge<Q, K, J + 1>::run(Pn, a, edge0, qstride);
        } else {
            // the last edge: symbol ql closes exactly the assignments of digit sum K B - ql
            bool any_conf = false;
#pragma unroll
            for (int ql = 0; ql < Q; ql++) {
                constexpr int KB = K * R::B;
                float M = INFINITY;
                if (KB - ql >= R::plo(J) && KB - ql <= R::phi(J)) M = P[KB - ql - R::plo(J)] + a[K - 1][ql];
                any_conf |= finite_f(M);
                edge0[((size_t)(K - 1) * Q + ql) * qstride] = finite_f(M) ? M - a[K - 1][ql] : INFINITY;
            }
            return any_conf;
        }
    }
};

template <int Q, int K>
__device__ __forceinline__ void q_check_dp(float *msg, int e0, long Bp, long b, int *err)
{
    static_assert(K >= 2, "at least one free edge");
    float a[K][Q];
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int q = 0; q < Q; q++) a[j][q] = msg[((size_t)(e0 + j) * Q + q) * Bp + b];
    const float P0[1] = {0.0f};  // S starts at 0 (decoder.rs:600)
    const bool any_conf = GDpEdge<Q, K, 0>::run(P0, a, msg + (size_t)e0 * Q * Bp + b, (size_t)Bp);
    if (!any_conf) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < Q; q++) any |= finite_f(a[j][q]);
            bad |= !any;
        }
        atomicMax(err, bad ? QERR_NO_FINITE : QERR_NO_CONFIG);
    }
}

// grid (R, Bp/64), block 64: lane = codeword.  Rows of fewer than 3 edges take the (tiny) unrolled enumeration.
template <int Q, int KMAX>
__global__ __launch_bounds__(64) void k_q_check_dp(const int *__restrict__ row_ptr, float *msg, long Bp, int batch, int *__restrict__ err)
{
    const int c = blockIdx.x;
    const long b = (long)blockIdx.y * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0;
#define QK(KK)                                                        \
    case KK:                                                          \
        if constexpr (KK <= KMAX) q_check_dp<Q, KK>(msg, e0, Bp, b, err); \
        break;
    switch (k) {
        case 1: q_check_unrolled<Q, 1>(msg, e0, Bp, b, err); break;
        case 2: q_check_unrolled<Q, 2>(msg, e0, Bp, b, err); break;
        QK(3) QK(4) QK(5) QK(6) QK(7) QK(8)
        default:
            if (threadIdx.x == 0) atomicMax(err, QERR_NO_CONFIG);  // k == 0 (k > KMAX never reaches this kernel)
    }
#undef QK
}

// grid (R, Bp/64), block 64.  Registers: 97 (Q = 3, DC = 7) / 115 (Q = 5, DC = 5) since round 4 -- four waves per SIMD.
// Until then a per-assignment configuration counter (v_cmp_class into an SGPR pair + add-with-carry for each of the 729
// assignments) had the compiler hold hundreds of masks: 294 registers, ONE wave per SIMD (two with a forced allocation
// and spills), and config 4's 2400 waves ran in two rounds: 48.9 -> 30.9 us per launch without it (see q_check_unrolled).
template <int Q, int KMAX>
__global__ __launch_bounds__(64) void k_q_check_unrolled(
    const int *__restrict__ row_ptr, float *msg, long Bp, int batch, int *__restrict__ err)
{
    const int c = blockIdx.x;
    const long b = (long)blockIdx.y * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int e0 = row_ptr[c], k = row_ptr[c + 1] - e0;
#define QK(KK)                                                              \
    case KK:                                                                \
        if constexpr (KK <= KMAX) q_check_unrolled<Q, KK>(msg, e0, Bp, b, err); \
        break;
    switch (k) {
        QK(1) QK(2) QK(3) QK(4) QK(5) QK(6) QK(7) QK(8)
        default:
            if (threadIdx.x == 0) atomicMax(err, QERR_NO_CONFIG);  // k == 0 (k > KMAX never reaches this kernel)
    }
#undef QK
}
}  // namespace

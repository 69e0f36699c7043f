// The two row updates of the min-sum record form compared message for message: `row_old` below is the first version
// (compare-select recurrences with per-lane state, ballots for the masks -- the recurrences of k_check_minsum_x word for
// word), check_minsum_row_rec is the PRODUCT's (scaldpc_bp_kernels.h, included as it stands: scalar lane masks,
// v_med3 / v_min on clamped magnitudes, v_writelane through the LLVM intrinsic).  Inputs full of ties, zeros, negative
// zeros, NaN, +-inf and FLT_MAX, EVERY degree 1 .. 64.  A difference is a bug: in round 3 this program found the
// VALU-writes-SGPR -> inline-asm v_writelane hazard on degree-1 rows.  It is a different translation unit from the
// product's (other register pressure, other schedule): the product kernels themselves face the oracle on every row
// degree in tests/test_bp_gpu.py::test_staircase_graph_every_degree, and their ISA is linted (tests/test_isa_lint.py).
// Run by tests/test_bp_gpu.py::test_record_row_update_equivalence.
// Build: make -C profiles/microbench rec_row_equivalence   (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off)
#include "../../sca-ldpc_amd/csrc/scaldpc_common.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
using namespace scaldpc;
#include "../../sca-ldpc_amd/csrc/scaldpc_bp_kernels.h"

namespace {
template <int DEG, bool FIRST>
__device__ __forceinline__ void row_old(const float *p, u64 synd_mask, float alpha, const float *__restrict__ prior,
                                                     const int *__restrict__ cidx, float *__restrict__ rec,
                                                     float *__restrict__ rec2, ulonglong2 *__restrict__ mask, int lane)
{
    unsigned par = (unsigned)(synd_mask >> lane) & 1u;
    float x[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) x[k] = FIRST ? prior[rfl(cidx[k])] : p[(size_t)k * TW];
    float m1 = FLT_MAX, m2 = FLT_MAX;
    int ix = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const float a = fabsf(x[k]);
        par ^= (unsigned)(x[k] <= 0.0f);
        const bool lt = a < m1;
        m2 = lt ? m1 : ((a < m2) ? a : m2);
        ix = lt ? k : ix;
        m1 = lt ? a : m1;
    }
    rec[lane] = m1 * alpha;
    rec2[lane] = m2 * alpha;
    u64 mneg = 0, marg = 0;  // lane k keeps edge k's two masks
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const u64 ng = __ballot((par ^ (unsigned)(x[k] <= 0.0f)) != 0);
        const u64 ag = __ballot(ix == k);
        if (lane == k) {
            mneg = ng;
            marg = ag;
        }
    }
    if (lane < DEG) mask[lane] = make_ulonglong2(mneg, marg);
}

template <int DEG>
__global__ void k(const float *msg, const u64 *synd, float alpha, float *recA, float *rec2A, ulonglong2 *maskA, float *recB, float *rec2B, ulonglong2 *maskB,
                  const int *iota)
{
    const int lane = threadIdx.x, r = blockIdx.x;
    const float *p = msg + (size_t)r * DEG * TW + lane;
    row_old<DEG, false>(p, synd[r], alpha, nullptr, nullptr, recA + r * TW, rec2A + r * TW, maskA + r * DEG, lane);
    // (the product lays its masks out by position in the variable pass's edge list; here edge k's position is k)
    check_minsum_row_rec<DEG>(p, synd[r], alpha, recB + r * TW, rec2B + r * TW, maskB + r * DEG, lane, iota);
}
template <int DEG>
int run(int rows, unsigned seed)
{
    srand(seed);
    std::vector<float> h((size_t)rows * DEG * TW);
    for (auto &v : h) {
        int t = rand() % 11;
        v = (float)(rand() % 7 - 3) * 0.5f;  // few distinct magnitudes: ties, zeros, negative zeros
        if (t == 0) v = -0.0f;
        if (t == 1) v = INFINITY;
        if (t == 2) v = -INFINITY;
        if (t == 3) v = NAN;
        if (t == 4) v = FLT_MAX;
    }
    std::vector<u64> sy(rows);
    for (auto &v : sy) v = ((u64)rand() << 40) ^ ((u64)rand() << 20) ^ rand();
    float *d, *ra, *r2a, *rb, *r2b; u64 *ds; ulonglong2 *ma, *mb; int *iota;
    int hi[64];
    for (int i = 0; i < 64; i++) hi[i] = i;
    hipMalloc(&iota, sizeof hi); hipMemcpy(iota, hi, sizeof hi, hipMemcpyHostToDevice);
    hipMalloc(&d, h.size() * 4); hipMalloc(&ds, rows * 8);
    hipMalloc(&ra, rows * TW * 4); hipMalloc(&r2a, rows * TW * 4); hipMalloc(&rb, rows * TW * 4); hipMalloc(&r2b, rows * TW * 4);
    hipMalloc(&ma, rows * DEG * 16); hipMalloc(&mb, rows * DEG * 16);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, sy.data(), rows * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<DEG>, dim3(rows), dim3(64), 0, 0, d, ds, 0.75f, ra, r2a, ma, rb, r2b, mb, iota);
    hipDeviceSynchronize();
    std::vector<float> A(rows * TW), A2(rows * TW), B(rows * TW), B2(rows * TW);
    std::vector<u64> MA(rows * DEG * 2), MB(rows * DEG * 2);
    hipMemcpy(A.data(), ra, A.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(A2.data(), r2a, A.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(B.data(), rb, A.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(B2.data(), r2b, A.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(MA.data(), ma, MA.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(MB.data(), mb, MB.size() * 8, hipMemcpyDeviceToHost);
    // compare the MESSAGES the variable pass would rebuild (an arg-min flag on another edge of the same magnitude is the same message)
    long bad = 0, badrec = 0, badneg = 0;
    for (int r = 0; r < rows; r++)
        for (int l = 0; l < TW; l++) {
            if (memcmp(&A[r * TW + l], &B[r * TW + l], 4) || memcmp(&A2[r * TW + l], &B2[r * TW + l], 4)) badrec++;
            for (int e = 0; e < DEG; e++) {
                const u64 na = MA[(r * DEG + e) * 2], aa = MA[(r * DEG + e) * 2 + 1], nb = MB[(r * DEG + e) * 2], ab = MB[(r * DEG + e) * 2 + 1];
                float va = ((aa >> l) & 1) ? A2[r * TW + l] : A[r * TW + l], vb = ((ab >> l) & 1) ? B2[r * TW + l] : B[r * TW + l];
                if (((na >> l) & 1) != ((nb >> l) & 1)) badneg++;
                if ((na >> l) & 1) va = -va;
                if ((nb >> l) & 1) vb = -vb;
                if (memcmp(&va, &vb, 4)) { if (bad < 6) printf("  row %d lane %d edge %d: x=%g old: m1a=%g m2a=%g neg=%d arg=%d | new: m1a=%g m2a=%g neg=%d arg=%d\n", r, l, e, h[((size_t)r * DEG + e) * TW + l], A[r*TW+l], A2[r*TW+l], (int)((na>>l)&1), (int)((aa>>l)&1), B[r*TW+l], B2[r*TW+l], (int)((nb>>l)&1), (int)((ab>>l)&1)); bad++; }
            }
        }
    printf("DEG %2d: %ld of %d messages differ, %ld records, %ld signs\n", DEG, bad, rows * TW * DEG, badrec, badneg);
    for (void *q : {(void *)d, (void *)ds, (void *)ra, (void *)r2a, (void *)rb, (void *)r2b, (void *)ma, (void *)mb, (void *)iota}) (void)hipFree(q);
    return bad != 0 || badrec != 0;
}
template <int DEG>
int run_all()
{
    int rc = run<DEG>(64, (unsigned)DEG);
    if constexpr (DEG < 64) rc |= run_all<DEG + 1>();
    return rc;
}
}  // namespace
int main()
{
    return run_all<1>();  // every degree the product instantiates
}

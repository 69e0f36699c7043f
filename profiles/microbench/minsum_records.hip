// Round 3 microbenchmark (VERDICT r02 task 8: "only revisit with a structurally new idea, and a microbenchmark first").
//
// Idea: a min-sum check node sends only TWO distinct magnitudes (the smallest and the second smallest incoming one) and
// a sign per edge.  Instead of writing deg messages per check (4 B per edge and codeword, read back by the variable
// pass as a gather), the check pass writes a RECORD per check -- min1[row][64], min2[row][64] -- plus, per edge and
// tile of 64 codewords, two 64-bit lane masks (sign, "this edge is the arg-min"): 8 B + 0.25 B per edge instead of 4 B
// per edge.  The variable pass rebuilds each message exactly (select + sign, same float as before, so nothing about
// bit-exactness changes) from the record of the edge's row (2 MB per tile for 4000 checks: L2-sized) and the masks.
// Fabric traffic per edge and iteration goes from 16 B to about 8.5 B; the question is what the record gathers cost.
//
// Geometry of the HQC-128 bench: R = 4000 checks of weight 50 + identity column, n = 21669, E = 204000,
// tiles of 64 codewords, T tiles per launch, message arrays [tile][edge][64].
//
// Build: hipcc -O3 --offload-arch=gfx950 -o minsum_records minsum_records.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int TW = 64;    // codewords per tile
constexpr int DEG = 51;   // check degree (50 + identity)
constexpr float ALPHA = 0.625f;

typedef unsigned long long u64;

struct Geo {
    int R, n, E, T;
};

// block -> (tile, block within tile).  mode 0: tile = blockIdx.y (every XCD sees every tile); mode 1: workgroups go
// round-robin over the 8 XCDs by linear id, so tile = xcd % T keeps a tile's records in T-th of the L2s.
__device__ __forceinline__ bool place(int mode, int T, int nblk, int &tile, int &blk)
{
    if (mode == 0) {
        tile = blockIdx.y;
        blk = blockIdx.x;
        return true;
    }
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    if (T >= 8) {  // T multiple of 8: tile = xcd + 8 * (slot / nblk)
        tile = xcd + 8 * (slot / nblk);
        blk = slot % nblk;
        return tile < T;
    }
    const int per = 8 / T;  // XCDs per tile
    tile = xcd % T;
    blk = slot * per + xcd / T;
    return blk < nblk;
}

// ---------------------------------------------------------------- baseline: message arrays both ways
__global__ __launch_bounds__(256) void k_check_base(const float *__restrict__ v2c, float *__restrict__ c2v,
                                                    const u64 *__restrict__ synd, Geo g, int mode, int nblk)
{
    int tile, blk;
    if (!place(mode, g.T, nblk, tile, blk)) return;
    const int lane = threadIdx.x & 63, r = blk * 4 + (threadIdx.x >> 6);
    if (r >= g.R) return;
    const size_t base = ((size_t)tile * g.E + (size_t)r * DEG) * TW + lane;
    float x[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) x[k] = v2c[base + (size_t)k * TW];
    float m1 = INFINITY, m2 = INFINITY;
    int arg = 0;
    unsigned s = (unsigned)((synd[(size_t)tile * g.R + r] >> lane) & 1) << 31;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const float a = fabsf(x[k]);
        s ^= __float_as_uint(x[k]) & 0x80000000u;
        if (a < m1) {
            m2 = m1;
            m1 = a;
            arg = k;
        } else if (a < m2)
            m2 = a;
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const float mag = ALPHA * (k == arg ? m2 : m1);
        c2v[base + (size_t)k * TW] = __uint_as_float(__float_as_uint(mag) | (s ^ (__float_as_uint(x[k]) & 0x80000000u)));
    }
}

template <int CAP>
__device__ __forceinline__ void var_base_body(const float *__restrict__ c2v, float *__restrict__ v2c, size_t tbase, int lane,
                                              int ev, int d, float pr)
{
    float m[CAP];
    float tot = pr;
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) {
            const int e = __builtin_amdgcn_readlane(ev, k);
            m[k] = c2v[tbase + (size_t)e * TW + lane];
        }
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) tot += m[k];
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) {
            const int e = __builtin_amdgcn_readlane(ev, k);
            v2c[tbase + (size_t)e * TW + lane] = tot - m[k];
        }
}

__global__ __launch_bounds__(256) void k_var_base(const float *__restrict__ c2v, float *__restrict__ v2c,
                                                  const int *__restrict__ colorder, const int *__restrict__ colptr,
                                                  const int *__restrict__ coledge, const float *__restrict__ prior, Geo g,
                                                  int mode, int nblk)
{
    int tile, blk;
    if (!place(mode, g.T, nblk, tile, blk)) return;
    const int lane = threadIdx.x & 63, ci = blk * 4 + (threadIdx.x >> 6);
    if (ci >= g.n) return;
    const int c = __builtin_amdgcn_readfirstlane(colorder[ci]);
    const int beg = __builtin_amdgcn_readfirstlane(colptr[c]), d = __builtin_amdgcn_readfirstlane(colptr[c + 1]) - beg;
    const int ev = lane < d ? coledge[beg + lane] : 0;
    const size_t tbase = (size_t)tile * g.E * TW;
    const float pr = prior[c];
    if (d <= 4)
        var_base_body<4>(c2v, v2c, tbase, lane, ev, d, pr);
    else if (d <= 16)
        var_base_body<16>(c2v, v2c, tbase, lane, ev, d, pr);
    else
        var_base_body<32>(c2v, v2c, tbase, lane, ev, d, pr);
}

// ---------------------------------------------------------------- records: min1 / min2 per check, two lane masks per edge
__global__ __launch_bounds__(256) void k_check_rec(const float *__restrict__ v2c, float *__restrict__ min1,
                                                   float *__restrict__ min2, ulonglong2 *__restrict__ masks,
                                                   const u64 *__restrict__ synd, Geo g, int mode, int nblk)
{
    int tile, blk;
    if (!place(mode, g.T, nblk, tile, blk)) return;
    const int lane = threadIdx.x & 63, r = blk * 4 + (threadIdx.x >> 6);
    if (r >= g.R) return;
    const size_t base = ((size_t)tile * g.E + (size_t)r * DEG) * TW + lane;
    float x[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) x[k] = v2c[base + (size_t)k * TW];
    float m1 = INFINITY, m2 = INFINITY;
    int arg = 0;
    unsigned s = (unsigned)((synd[(size_t)tile * g.R + r] >> lane) & 1) << 31;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const float a = fabsf(x[k]);
        s ^= __float_as_uint(x[k]) & 0x80000000u;
        if (a < m1) {
            m2 = m1;
            m1 = a;
            arg = k;
        } else if (a < m2)
            m2 = a;
    }
    const size_t rb = ((size_t)tile * g.R + r) * TW + lane;
    min1[rb] = ALPHA * m1;
    min2[rb] = ALPHA * m2;
    u64 mys = 0, mya = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const u64 sm = __ballot(((s ^ __float_as_uint(x[k])) & 0x80000000u) != 0);
        const u64 am = __ballot(arg == k);
        if (lane == k) {
            mys = sm;
            mya = am;
        }
    }
    if (lane < DEG) masks[(size_t)tile * g.E + (size_t)r * DEG + lane] = make_ulonglong2(mys, mya);
}

template <int CAP>
__device__ __forceinline__ void var_rec_body(const float *__restrict__ min1, const float *__restrict__ min2, float *__restrict__ v2c,
                                             size_t tbase, size_t rbase, int lane, int ev, u64 sv, u64 av, int d, float pr)
{
    float m[CAP];
    float tot = pr;
    const unsigned slo = (unsigned)sv, shi = (unsigned)(sv >> 32), alo = (unsigned)av, ahi = (unsigned)(av >> 32);
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) {
            const int e = __builtin_amdgcn_readlane(ev, k);
            const int r = e / DEG;
            const u64 sm = ((u64)(unsigned)__builtin_amdgcn_readlane((int)shi, k) << 32) | (unsigned)__builtin_amdgcn_readlane((int)slo, k);
            const u64 am = ((u64)(unsigned)__builtin_amdgcn_readlane((int)ahi, k) << 32) | (unsigned)__builtin_amdgcn_readlane((int)alo, k);
            float mag = min1[rbase + (size_t)r * TW + lane];
            if ((am >> lane) & 1) mag = min2[rbase + (size_t)r * TW + lane];
            m[k] = __uint_as_float(__float_as_uint(mag) | ((unsigned)((sm >> lane) & 1) << 31));
        }
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) tot += m[k];
#pragma unroll
    for (int k = 0; k < CAP; k++)
        if (k < d) {
            const int e = __builtin_amdgcn_readlane(ev, k);
            v2c[tbase + (size_t)e * TW + lane] = tot - m[k];
        }
}

__global__ __launch_bounds__(256) void k_var_rec(const float *__restrict__ min1, const float *__restrict__ min2,
                                                 const ulonglong2 *__restrict__ masks, float *__restrict__ v2c,
                                                 const int *__restrict__ colorder, const int *__restrict__ colptr,
                                                 const int *__restrict__ coledge, const float *__restrict__ prior, Geo g,
                                                 int mode, int nblk)
{
    int tile, blk;
    if (!place(mode, g.T, nblk, tile, blk)) return;
    const int lane = threadIdx.x & 63, ci = blk * 4 + (threadIdx.x >> 6);
    if (ci >= g.n) return;
    const int c = __builtin_amdgcn_readfirstlane(colorder[ci]);
    const int beg = __builtin_amdgcn_readfirstlane(colptr[c]), d = __builtin_amdgcn_readfirstlane(colptr[c + 1]) - beg;
    const int ev = lane < d ? coledge[beg + lane] : 0;
    ulonglong2 mk = make_ulonglong2(0, 0);
    if (lane < d) mk = masks[(size_t)tile * g.E + ev];
    const size_t tbase = (size_t)tile * g.E * TW, rbase = (size_t)tile * g.R * TW;
    const float pr = prior[c];
    if (d <= 4)
        var_rec_body<4>(min1, min2, v2c, tbase, rbase, lane, ev, mk.x, mk.y, d, pr);
    else if (d <= 16)
        var_rec_body<16>(min1, min2, v2c, tbase, rbase, lane, ev, mk.x, mk.y, d, pr);
    else
        var_rec_body<32>(min1, min2, v2c, tbase, rbase, lane, ev, mk.x, mk.y, d, pr);
}

// ----------------------------------------------------------------
static float time_loop(hipStream_t s, int reps, const std::function<void()> &f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms * 1000.f / reps;
}

int main(int argc, char **argv)
{
    const int R = 4000, NIN = 17669, W = 50;
    const int n = NIN + R, E = R * DEG;
    std::mt19937 rng(1);
    // CSR: row r = W distinct columns of Hin + identity column NIN + r (edge order = column order inside the row)
    std::vector<int> col(E);
    for (int r = 0; r < R; r++) {
        std::vector<int> pick;
        while ((int)pick.size() < W) {
            int c = rng() % NIN;
            if (std::find(pick.begin(), pick.end(), c) == pick.end()) pick.push_back(c);
        }
        std::sort(pick.begin(), pick.end());
        for (int k = 0; k < W; k++) col[r * DEG + k] = pick[k];
        col[r * DEG + W] = NIN + r;
    }
    std::vector<int> colptr(n + 1, 0), coledge(E), colorder(n);
    for (int e = 0; e < E; e++) colptr[col[e] + 1]++;
    for (int c = 0; c < n; c++) colptr[c + 1] += colptr[c];
    {
        std::vector<int> fill(colptr.begin(), colptr.end() - 1);
        for (int e = 0; e < E; e++) coledge[fill[col[e]]++] = e;
    }
    int maxd = 0;
    for (int c = 0; c < n; c++) maxd = std::max(maxd, colptr[c + 1] - colptr[c]);
    for (int c = 0; c < n; c++) colorder[c] = c;
    std::stable_sort(colorder.begin(), colorder.end(),
                     [&](int a, int b) { return colptr[a + 1] - colptr[a] > colptr[b + 1] - colptr[b]; });  // heaviest first
    printf("graph: R=%d n=%d E=%d, max column degree %d\n", R, n, E, maxd);
    if (maxd > 32) {
        printf("column degree above 32: not handled here\n");
        return 1;
    }
    std::vector<float> prior(n);
    for (int c = 0; c < n; c++) prior[c] = c < NIN ? 2.9f : 1.5f;

    int *d_colptr, *d_coledge, *d_colorder;
    float *d_prior;
    CK(hipMalloc(&d_colptr, sizeof(int) * (n + 1)));
    CK(hipMalloc(&d_coledge, sizeof(int) * E));
    CK(hipMalloc(&d_colorder, sizeof(int) * n));
    CK(hipMalloc(&d_prior, sizeof(float) * n));
    CK(hipMemcpy(d_colptr, colptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_coledge, coledge.data(), sizeof(int) * E, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_colorder, colorder.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prior, prior.data(), sizeof(float) * n, hipMemcpyHostToDevice));

    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int TMAX = 8;
    float *d_v2c, *d_c2v, *d_v2c_ref, *d_min1, *d_min2;
    ulonglong2 *d_masks;
    u64 *d_synd;
    const size_t msgs = (size_t)TMAX * E * TW;
    CK(hipMalloc(&d_v2c, msgs * 4));
    CK(hipMalloc(&d_v2c_ref, msgs * 4));
    CK(hipMalloc(&d_c2v, msgs * 4));
    CK(hipMalloc(&d_min1, (size_t)TMAX * R * TW * 4));
    CK(hipMalloc(&d_min2, (size_t)TMAX * R * TW * 4));
    CK(hipMalloc(&d_masks, (size_t)TMAX * E * sizeof(ulonglong2)));
    CK(hipMalloc(&d_synd, (size_t)TMAX * R * 8));
    {
        std::vector<float> h(msgs);
        std::normal_distribution<float> nd(1.5f, 2.0f);
        for (auto &v : h) v = nd(rng);
        CK(hipMemcpy(d_v2c, h.data(), msgs * 4, hipMemcpyHostToDevice));
        std::vector<u64> sy((size_t)TMAX * R);
        for (auto &v : sy) v = ((u64)rng() << 32) | rng();
        CK(hipMemcpy(d_synd, sy.data(), sy.size() * 8, hipMemcpyHostToDevice));
    }

    // ---- equality: one check + variable pass, both ways, from the same v2c
    {
        Geo g{R, n, E, TMAX};
        const int nbr = (R + 3) / 4, nbc = (n + 3) / 4;
        CK(hipMemcpy(d_v2c_ref, d_v2c, msgs * 4, hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(k_check_base, dim3(nbr, TMAX), dim3(256), 0, s, d_v2c_ref, d_c2v, d_synd, g, 0, nbr);
        hipLaunchKernelGGL(k_var_base, dim3(nbc, TMAX), dim3(256), 0, s, d_c2v, d_v2c_ref, d_colorder, d_colptr, d_coledge, d_prior, g, 0, nbc);
        std::vector<float> keep(msgs);
        CK(hipMemcpy(keep.data(), d_v2c, msgs * 4, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_check_rec, dim3(nbr, TMAX), dim3(256), 0, s, d_v2c, d_min1, d_min2, d_masks, d_synd, g, 0, nbr);
        hipLaunchKernelGGL(k_var_rec, dim3(nbc, TMAX), dim3(256), 0, s, d_min1, d_min2, d_masks, d_v2c, d_colorder, d_colptr, d_coledge, d_prior, g, 0, nbc);
        CK(hipStreamSynchronize(s));
        std::vector<float> a(msgs), b(msgs);
        CK(hipMemcpy(a.data(), d_v2c_ref, msgs * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_v2c, msgs * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < msgs; i++) diff += memcmp(&a[i], &b[i], 4) != 0;
        printf("one iteration both ways: %zu of %zu variable-to-check messages differ bit-wise\n", diff, msgs);
        CK(hipMemcpy(d_v2c, keep.data(), msgs * 4, hipMemcpyHostToDevice));
    }

    const int reps = 40;
    for (int T : {2, 4, 8}) {
        Geo g{R, n, E, T};
        const int nbr = (R + 3) / 4, nbc = (n + 3) / 4;
        for (int mode = 0; mode < 2; mode++) {
            dim3 gr, gc;
            if (mode == 0) {
                gr = dim3(nbr, T);
                gc = dim3(nbc, T);
            } else if (T >= 8) {
                gr = dim3(nbr * 8 * (T / 8));
                gc = dim3(nbc * 8 * (T / 8));
            } else {
                const int per = 8 / T;
                gr = dim3(((nbr + per - 1) / per) * 8);
                gc = dim3(((nbc + per - 1) / per) * 8);
            }
            auto cb = [&] { hipLaunchKernelGGL(k_check_base, gr, dim3(256), 0, s, d_v2c, d_c2v, d_synd, g, mode, nbr); };
            auto vb = [&] { hipLaunchKernelGGL(k_var_base, gc, dim3(256), 0, s, d_c2v, d_v2c, d_colorder, d_colptr, d_coledge, d_prior, g, mode, nbc); };
            auto cr = [&] { hipLaunchKernelGGL(k_check_rec, gr, dim3(256), 0, s, d_v2c, d_min1, d_min2, d_masks, d_synd, g, mode, nbr); };
            auto vr = [&] { hipLaunchKernelGGL(k_var_rec, gc, dim3(256), 0, s, d_min1, d_min2, d_masks, d_v2c, d_colorder, d_colptr, d_coledge, d_prior, g, mode, nbc); };
            const float pb = time_loop(s, reps, [&] { cb(); vb(); });
            const float tcb = time_loop(s, reps, cb), tvb = time_loop(s, reps, vb);
            const float pr = time_loop(s, reps, [&] { cr(); vr(); });
            const float tcr = time_loop(s, reps, cr), tvr = time_loop(s, reps, vr);
            const double alg = 16.0 * E * TW * T;  // bytes per pair, message-array scheme
            printf("T=%d tiles (%4d codewords) %-9s | messages: pair %7.1f us (check %6.1f, var %6.1f) %5.2f TB/s | records: pair %7.1f us (check %6.1f, var %6.1f) | %.2fx\n",
                   T, T * TW, mode ? "xcd-aware" : "plain", pb, tcb, tvb, alg / pb * 1e-6, pr, tcr, tvr, pb / pr);
        }
    }
    CK(hipGetLastError());
    return 0;
}

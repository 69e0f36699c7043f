// Microbenchmark (evidence for DESIGN.md, not product code): the register-resident min-sum row
// update of k_check_minsum_x (exact degree 51, all loads first, same recurrences) on the bench
// geometry -- 4000 rows x 51 edges, 256 codewords resident (208.9 MB) -- with
//   (a) 4-byte lanes: wave = (row, tile of 64 codewords), msg[tile][edge][64]          (today)
//   (b) 8-byte lanes: wave = (row, PAIR of tiles),      msg[pair][edge][64][2]  (two codewords per lane)
// and the gather side of it, a k_var-like column pass (degree 11, random edges):
//   (c) 4-byte lanes   (d) 8-byte lanes.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o check_lane_width check_lane_width.hip
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int DEG = 51, ROWS = 4000, CDEG = 11;
constexpr long E = (long)DEG * ROWS;

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <int W>  // W floats per lane
__global__ __launch_bounds__(256) void k_check(float *msg, float alpha)
{
    const int lane = threadIdx.x & 63;
    const int r = rfl((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (r >= ROWS) return;
    float *p = msg + ((size_t)blockIdx.y * E + (size_t)r * DEG) * 64 * W + lane * W;
    float x[DEG][W];
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (W == 1) x[k][0] = p[(size_t)k * 64];
        else { const float2 v = *(const float2 *)(p + (size_t)k * 128); x[k][0] = v.x; x[k][W - 1] = v.y; }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        float m1 = FLT_MAX, m2 = FLT_MAX;
        int ix = 0;
        unsigned par = 0;
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            const float a = fabsf(x[k][w]);
            par ^= (unsigned)(x[k][w] <= 0.0f);
            const bool lt = a < m1;
            m2 = lt ? m1 : ((a < m2) ? a : m2);
            ix = lt ? k : ix;
            m1 = lt ? a : m1;
        }
#pragma unroll
        for (int k = 0; k < DEG; k++) x[k][w] = ((k == ix) ? m2 : m1) * ((par ^ (unsigned)(x[k][w] <= 0.0f)) ? -alpha : alpha);
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (W == 1) p[(size_t)k * 64] = x[k][0];
        else *(float2 *)(p + (size_t)k * 128) = make_float2(x[k][0], x[k][W - 1]);
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_var(float *msg, const int *__restrict__ edges, int ncol)
{
    const int lane = threadIdx.x & 63;
    const int c = rfl((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (c >= ncol) return;
    const int *ce = edges + (size_t)c * CDEG;
    float *mt = msg + (size_t)blockIdx.y * E * 64 * W + lane * W;
    float mm[CDEG][W], pp[CDEG][W];
#pragma unroll
    for (int k = 0; k < CDEG; k++) {
        const size_t o = (size_t)rfl(ce[k]) * 64 * W;
        if (W == 1) mm[k][0] = mt[o];
        else { const float2 v = *(const float2 *)(mt + o); mm[k][0] = v.x; mm[k][W - 1] = v.y; }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        float temp = 0.25f;
#pragma unroll
        for (int k = 0; k < CDEG; k++) { pp[k][w] = temp; temp += mm[k][w]; }
        float suf = 0.0f;
#pragma unroll
        for (int k = CDEG - 1; k >= 0; k--) { pp[k][w] += suf; suf += mm[k][w]; }
    }
#pragma unroll
    for (int k = 0; k < CDEG; k++) {
        const size_t o = (size_t)rfl(ce[k]) * 64 * W;
        if (W == 1) mt[o] = pp[k][0];
        else *(float2 *)(mt + o) = make_float2(pp[k][0], pp[k][W - 1]);
    }
}

int main()
{
    const size_t bytes = (size_t)E * 256 * 4;  // 4 tiles of 64 codewords
    float *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0x3c, bytes));
    // a random edge permutation dealt to columns of degree CDEG (like the CSC gather)
    const int ncol = (int)(E / CDEG);
    std::vector<int> perm((size_t)ncol * CDEG);
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
    srand(1);
    for (size_t i = perm.size() - 1; i > 0; i--) { size_t j = (size_t)rand() % (i + 1); std::swap(perm[i], perm[j]); }
    int *de; CK(hipMalloc(&de, perm.size() * 4)); CK(hipMemcpy(de, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 50;
    auto timeit = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-34s %7.2f us/launch  %7.1f GB/s (read+write of 208.9 MB)\n", name, ms * 1e3 / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
        return 0;
    };
    const unsigned rb = (ROWS + 3) / 4, cb = (unsigned)((ncol + 3) / 4);
    for (int round = 0; round < 2; round++) {
        if (timeit("check, 4 B lanes (4 tiles)", [&] { hipLaunchKernelGGL(k_check<1>, dim3(rb, 4), dim3(256), 0, 0, d, 0.9f); })) return 1;
        if (timeit("check, 8 B lanes (2 tile pairs)", [&] { hipLaunchKernelGGL(k_check<2>, dim3(rb, 2), dim3(256), 0, 0, d, 0.9f); })) return 1;
        if (timeit("var,   4 B lanes (4 tiles)", [&] { hipLaunchKernelGGL(k_var<1>, dim3(cb, 4), dim3(256), 0, 0, d, de, ncol); })) return 1;
        if (timeit("var,   8 B lanes (2 tile pairs)", [&] { hipLaunchKernelGGL(k_var<2>, dim3(cb, 2), dim3(256), 0, 0, d, de, ncol); })) return 1;
    }
    return 0;
}

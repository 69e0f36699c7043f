// Microbenchmark (evidence for DESIGN.md section 7, not product code): would HALF tiles pay on the HQC-256
// graph (BASELINE config 3)?  There one 64-codeword tile (156.7 MB of messages) is a cache-resident group of
// its own, so the decode runs as ONE stream lane and every launch pays its own ramp-up and tail.  Half tiles
// = 32 codewords, a wave takes TWO rows (or two columns), lanes 0-31 the first, lanes 32-63 the second; the
// two halves of a tile are independent and could run as two lanes, one kernel out of phase.
//   (a) check pass, full tile     wave = (row, 64 codewords)          msg[edge][64]
//   (b) check pass, half tiles    wave = (2 rows, 32 codewords)       msg[half][edge][32], both halves back to back
//   (c) var pass, full tile       wave = (column, 64 cw), degree 11, edge ids through scalar loads
//   (d) var pass, half tiles      wave = (2 columns, 32 cw), edge ids per lane
//   (e) 50 iterations check -> var: full tile on one stream  vs  the two halves on two streams, out of phase
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o half_tile half_tile.hip
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int DEG = 51, ROWS = 12000, CDEG = 11;
constexpr long E = (long)DEG * ROWS;
__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ float tanh_compl(float a) { return 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a * 1.44269504f) + 1.0f); }
__device__ __forceinline__ float llr_from_compl(float U) { return __builtin_amdgcn_logf(fmaf(2.0f, __builtin_amdgcn_rcpf(U), -1.0f)) * 0.693147181f; }
__device__ __forceinline__ float compl_step(float U, float u) { return fmaf(u, 1.0f - U, U); }

template <int STRIDE>
__device__ __forceinline__ void tanh_row(float *p)  // the product kernel's row update (complement form), DEG edges
{
    float uu[DEG], pre[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) uu[k] = p[(size_t)k * STRIDE];
    unsigned acc = 0;
    float U = 0.0f;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const unsigned xb = __float_as_uint(uu[k]);
        acc ^= xb;
        const float u = tanh_compl(fabsf(uu[k]));
        uu[k] = __uint_as_float(__float_as_uint(u) | (xb & 0x80000000u));
        pre[k] = U;
        U = compl_step(U, u);
    }
    U = 0.0f;
#pragma unroll
    for (int k = DEG - 1; k >= 0; k--) {
        const float Lm = llr_from_compl(compl_step(pre[k], U));
        const unsigned sg = (acc ^ __float_as_uint(uu[k])) & 0x80000000u;
        U = compl_step(U, fabsf(uu[k]));
        p[(size_t)k * STRIDE] = __uint_as_float(__float_as_uint(Lm) ^ sg);
    }
}

__global__ __launch_bounds__(256) void k_check_full(float *msg)
{
    const int lane = threadIdx.x & 63;
    const int r = rfl((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (r >= ROWS) return;
    tanh_row<64>(msg + (size_t)r * DEG * 64 + lane);
}
__global__ __launch_bounds__(256) void k_check_half(float *msgh)  // one half: [edge][32]
{
    const int lane = threadIdx.x & 63;
    const int r = 2 * ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) + (lane >> 5);
    if (r >= ROWS) return;
    tanh_row<32>(msgh + (size_t)r * DEG * 32 + (lane & 31));
}

__global__ __launch_bounds__(256) void k_var_full(float *msg, const int *__restrict__ edges, int ncol)
{
    const int lane = threadIdx.x & 63;
    const int c = rfl((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (c >= ncol) return;
    const int *ce = edges + (size_t)c * CDEG;
    float *mt = msg + lane;
    float mm[CDEG], pp[CDEG];
#pragma unroll
    for (int k = 0; k < CDEG; k++) mm[k] = mt[(size_t)rfl(ce[k]) * 64];
    float temp = 0.25f;
#pragma unroll
    for (int k = 0; k < CDEG; k++) { pp[k] = temp; temp += mm[k]; }
    float suf = 0.0f;
#pragma unroll
    for (int k = CDEG - 1; k >= 0; k--) { mt[(size_t)rfl(ce[k]) * 64] = pp[k] + suf; suf += mm[k]; }
}
__global__ __launch_bounds__(256) void k_var_half(float *msgh, const int *__restrict__ edges, int ncol)
{
    const int lane = threadIdx.x & 63;
    const int c = 2 * ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) + (lane >> 5);
    if (c >= ncol) return;
    const int *ce = edges + (size_t)c * CDEG;  // per lane: two columns per wave
    float *mt = msgh + (lane & 31);
    float mm[CDEG], pp[CDEG];
#pragma unroll
    for (int k = 0; k < CDEG; k++) mm[k] = mt[(size_t)ce[k] * 32];
    float temp = 0.25f;
#pragma unroll
    for (int k = 0; k < CDEG; k++) { pp[k] = temp; temp += mm[k]; }
    float suf = 0.0f;
#pragma unroll
    for (int k = CDEG - 1; k >= 0; k--) { mt[(size_t)ce[k] * 32] = pp[k] + suf; suf += mm[k]; }
}

int main()
{
    const size_t bytes = (size_t)E * 64 * 4;  // one tile: 156.7 MB
    float *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0x3c, bytes));
    float *h0 = d, *h1 = d + (size_t)E * 32;  // the same memory seen as two halves
    const int ncol = (int)(E / CDEG);
    std::vector<int> perm((size_t)ncol * CDEG);
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
    srand(1);
    for (size_t i = perm.size() - 1; i > 0; i--) { size_t j = (size_t)rand() % (i + 1); std::swap(perm[i], perm[j]); }
    int *de; CK(hipMalloc(&de, perm.size() * 4)); CK(hipMemcpy(de, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t a, b, j; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    const int reps = 50;
    const unsigned rbf = (ROWS + 3) / 4, rbh = (ROWS / 2 + 3) / 4, cbf = (unsigned)((ncol + 3) / 4), cbh = (unsigned)((ncol / 2 + 3) / 4);
    auto timeit = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; i++) launch();
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(a, s0));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s0, j, 0));
        CK(hipEventRecord(b, s0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-64s %8.2f us per pass over the tile  %7.1f GB/s\n", name, ms * 1e3 / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
        return 0;
    };
    for (int round = 0; round < 2; round++) {
        if (timeit("(a) check, full tile", [&] { hipLaunchKernelGGL(k_check_full, dim3(rbf), dim3(256), 0, s0, d); })) return 1;
        if (timeit("(b) check, two halves back to back (one stream)", [&] {
                hipLaunchKernelGGL(k_check_half, dim3(rbh), dim3(256), 0, s0, h0);
                hipLaunchKernelGGL(k_check_half, dim3(rbh), dim3(256), 0, s0, h1); })) return 1;
        if (timeit("(c) var, full tile, scalar edge ids", [&] { hipLaunchKernelGGL(k_var_full, dim3(cbf), dim3(256), 0, s0, d, de, ncol); })) return 1;
        if (timeit("(d) var, two halves back to back, per-lane edge ids", [&] {
                hipLaunchKernelGGL(k_var_half, dim3(cbh), dim3(256), 0, s0, h0, de, ncol);
                hipLaunchKernelGGL(k_var_half, dim3(cbh), dim3(256), 0, s0, h1, de, ncol); })) return 1;
        if (timeit("(e1) iteration = check -> var, full tile, one stream", [&] {
                hipLaunchKernelGGL(k_check_full, dim3(rbf), dim3(256), 0, s0, d);
                hipLaunchKernelGGL(k_var_full, dim3(cbf), dim3(256), 0, s0, d, de, ncol); })) return 1;
        {   // two lanes: half 0 on s0, half 1 on s1, s1 started one kernel late
            bool first = true;
            if (timeit("(e2) iteration = check -> var, two halves on two streams", [&] {
                    hipLaunchKernelGGL(k_check_half, dim3(rbh), dim3(256), 0, s0, h0);
                    if (first) { (void)hipEventRecord(j, s0); (void)hipStreamWaitEvent(s1, j, 0); first = false; }
                    hipLaunchKernelGGL(k_check_half, dim3(rbh), dim3(256), 0, s1, h1);
                    hipLaunchKernelGGL(k_var_half, dim3(cbh), dim3(256), 0, s0, h0, de, ncol);
                    hipLaunchKernelGGL(k_var_half, dim3(cbh), dim3(256), 0, s1, h1, de, ncol); })) return 1;
        }
    }
    return 0;
}

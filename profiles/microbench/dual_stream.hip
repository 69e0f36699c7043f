// Microbenchmark (evidence for DESIGN.md, not product code): can an Infinity-Cache-resident
// in-place stream (normal loads/stores, 209 MB working set) and a non-temporal HBM stream
// (nt loads/stores, 3.3 GB working set) run CONCURRENTLY on two HIP streams without slowing
// each other down?  Same access shape as the check kernel (wave = 51 consecutive 256 B rows).
// build: hipcc --offload-arch=gfx950 -O3 -o dual_stream dual_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void rmw(float *buf, size_t nvec, int chunk)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t base = wave * (size_t)chunk * 64 + lane;
    if (base + (size_t)(chunk - 1) * 64 >= nvec) return;
    float acc = 0.f;
    for (int k = 0; k < chunk; k++) acc += NT ? __builtin_nontemporal_load(buf + base + (size_t)k * 64) : buf[base + (size_t)k * 64];
    for (int k = 0; k < chunk; k++) {
        if (NT) __builtin_nontemporal_store(acc + k, buf + base + (size_t)k * 64);
        else buf[base + (size_t)k * 64] = acc + k;
    }
}

int main()
{
    const size_t small = (size_t)(209e6) / (51 * 256) * (51 * 256), big = (size_t)(3344e6) / (51 * 256) * (51 * 256);
    float *a, *b; CK(hipMalloc(&a, small)); CK(hipMalloc(&b, big)); CK(hipMemset(a, 0, small)); CK(hipMemset(b, 0, big));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t ev[4]; for (auto &x : ev) CK(hipEventCreate(&x));
    const size_t na = small / 4, nb = big / 4;
    dim3 ga((unsigned)((na / (51 * 64) + 3) / 4)), gb((unsigned)((nb / (51 * 64) + 3) / 4));
    const int ra = 160, rb = 10;  // equal bytes per stream: 160 x 209 MB = 10 x 3344 MB
    auto run = [&](bool do_a, bool do_b, bool b_nt, const char *label) -> int {
        for (int w = 0; w < 2; w++) { if (do_a) hipLaunchKernelGGL(rmw<false>, ga, dim3(256), 0, s1, a, na, 51); if (do_b) { if (b_nt) hipLaunchKernelGGL(rmw<true>, gb, dim3(256), 0, s2, b, nb, 51); else hipLaunchKernelGGL(rmw<false>, gb, dim3(256), 0, s2, b, nb, 51); } }
        CK(hipDeviceSynchronize());
        if (do_a) CK(hipEventRecord(ev[0], s1));
        if (do_b) CK(hipEventRecord(ev[2], s2));
        for (int i = 0; i < ra; i++) {
            if (do_a) hipLaunchKernelGGL(rmw<false>, ga, dim3(256), 0, s1, a, na, 51);
            if (do_b && i < rb) { if (b_nt) hipLaunchKernelGGL(rmw<true>, gb, dim3(256), 0, s2, b, nb, 51); else hipLaunchKernelGGL(rmw<false>, gb, dim3(256), 0, s2, b, nb, 51); }
        }
        if (do_a) CK(hipEventRecord(ev[1], s1));
        if (do_b) CK(hipEventRecord(ev[3], s2));
        CK(hipDeviceSynchronize());
        float ma = 0, mb = 0;
        if (do_a) CK(hipEventElapsedTime(&ma, ev[0], ev[1]));
        if (do_b) CK(hipEventElapsedTime(&mb, ev[2], ev[3]));
        double ta = do_a ? 2.0 * small * ra / (ma * 1e-3) / 1e9 : 0, tb = do_b ? 2.0 * big * rb / (mb * 1e-3) / 1e9 : 0;
        double wall = (ma > mb ? ma : mb) * 1e-3;
        double agg = ((do_a ? 2.0 * small * ra : 0) + (do_b ? 2.0 * big * rb : 0)) / wall / 1e9;
        printf("%-46s cache-resident %7.0f GB/s (%6.1f ms)   HBM stream %7.0f GB/s (%6.1f ms)   aggregate %7.0f GB/s\n", label, ta, ma, tb, mb, agg);
        return 0;
    };
    if (run(true, false, false, "cache-resident alone")) return 1;
    if (run(false, true, true, "HBM nt alone")) return 1;
    if (run(false, true, false, "HBM normal alone")) return 1;
    if (run(true, true, true, "cache-resident + HBM nt concurrently")) return 1;
    if (run(true, true, false, "cache-resident + HBM normal concurrently")) return 1;
    return 0;
}

// Microbenchmark (evidence for DESIGN.md, not product code): in-place read-modify-write
// stream over a working set of S bytes, repeated, to see what the 256 MiB Infinity
// Cache (MALL) buys when the BP message array of one codeword group stays resident.
// build: hipcc --offload-arch=gfx950 -O3 -o rmw_stream rmw_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each wave owns `chunk` consecutive rows of 64 lanes x VEC floats: reads all, then writes all (like a check node)
template <typename V>
__global__ __launch_bounds__(256) void rmw(V *buf, size_t nvec, int chunk)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t base = wave * (size_t)chunk * 64 + lane;
    if (base + (size_t)(chunk - 1) * 64 >= nvec) return;
    float acc = 0.f;
    for (int k = 0; k < chunk; k++) { V x = buf[base + (size_t)k * 64]; acc += ((float *)&x)[0]; }
    for (int k = 0; k < chunk; k++) { V o; for (unsigned j = 0; j < sizeof(V) / 4; j++) ((float *)&o)[j] = acc + k; buf[base + (size_t)k * 64] = o; }
}

// ELL-style: step k of every wave touches one compact window: row index = k * nwaves + wave
template <typename V>
__global__ __launch_bounds__(256) void rmw_ell(V *buf, size_t nwaves, int chunk)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    float acc = 0.f;
    for (int k = 0; k < chunk; k++) { V x = buf[((size_t)k * nwaves + wave) * 64 + lane]; acc += ((float *)&x)[0]; }
    for (int k = 0; k < chunk; k++) { V o; for (unsigned j = 0; j < sizeof(V) / 4; j++) ((float *)&o)[j] = acc + k; buf[((size_t)k * nwaves + wave) * 64 + lane] = o; }
}

template <typename V>
int run_ell(const char *name, size_t bytes, int chunk, int reps)
{
    V *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
    size_t nvec = bytes / sizeof(V);
    size_t waves = nvec / ((size_t)chunk * 64);
    dim3 grid((unsigned)((waves + 3) / 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(rmw_ell<V>, grid, dim3(256), 0, 0, d, waves, chunk);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(rmw_ell<V>, grid, dim3(256), 0, 0, d, waves, chunk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double gbs = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
    printf("ELL %-8s S=%7.1f MB chunk=%3d  %8.1f us/launch  %8.1f GB/s (read+write)\n", name, bytes / 1e6, chunk, ms * 1e3 / reps, gbs);
    CK(hipFree(d));
    return 0;
}

template <typename V>
int run(const char *name, size_t bytes, int chunk, int reps)
{
    V *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
    size_t nvec = bytes / sizeof(V);
    size_t waves = nvec / ((size_t)chunk * 64);
    dim3 grid((unsigned)((waves + 3) / 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(rmw<V>, grid, dim3(256), 0, 0, d, nvec, chunk);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(rmw<V>, grid, dim3(256), 0, 0, d, nvec, chunk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double gbs = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-8s S=%7.1f MB chunk=%3d  %8.1f us/launch  %8.1f GB/s (read+write)\n", name, bytes / 1e6, chunk, ms * 1e3 / reps, gbs);
    CK(hipFree(d));
    return 0;
}

int main()
{
    const double sizes_mb[] = {104, 157, 209, 836, 1672, 3344};
    for (double mb : sizes_mb) {
        size_t bytes = (size_t)(mb * 1e6) / (51 * 1024) * (51 * 1024);
        int reps = mb < 500 ? 50 : 10;
        if (run<float>("dword", bytes, 51, reps)) return 1;
        if (run<float2>("dwordx2", bytes, 51, reps)) return 1;
        if (run<float4>("dwordx4", bytes, 51, reps)) return 1;
        if (run_ell<float>("dword", bytes, 51, reps)) return 1;
        if (run_ell<float2>("dwordx2", bytes, 51, reps)) return 1;
        if (run_ell<float4>("dwordx4", bytes, 51, reps)) return 1;
    }
    return 0;
}

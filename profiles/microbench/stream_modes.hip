// Read-only, write-only and mixed streams over a cache-resident (209 MB) and an HBM-sized (1 GB) buffer, in the access shape of the
// BP passes (a wave owns consecutive 256-B rows: 64 lanes x 4 B): is the in-place stream's ceiling (6.7 TB/s, half reads
// half writes) a limit on the total, or on the WRITES?  The record-form launch pair moves 136 MB of reads and 120 MB of
// writes per pair and lane (profiles/r04/pmc_traffic_hqc128_minsum.json).
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_modes stream_modes.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define HIPOK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

// a wave reads its first NR rows -- every load issued before the first use, as the product's row kernels do -- and writes its
// last nw rows of a block of `rows` rows
template <int NR>
__global__ __launch_bounds__(256) void k_stream(float *buf, size_t nrows, int rows, int nw, float magic)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t r0 = wave * (size_t)rows;
    if (r0 + rows > nrows) return;
    float *p = buf + r0 * 64 + lane;
    float x[NR > 0 ? NR : 1];
#pragma unroll
    for (int k = 0; k < NR; k++) x[k] = p[(size_t)k * 64];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NR; k++) acc += x[k];
    if (nw == 0) {
        if (acc == magic) p[0] = acc;  // (never: keeps the loads alive)
        return;
    }
    for (int k = rows - nw; k < rows; k++) p[(size_t)k * 64] = acc + (float)k;
}

template <int NR>
void launch(dim3 grid, float *buf, size_t nrows, int rows, int nw)
{
    hipLaunchKernelGGL(k_stream<NR>, grid, dim3(256), 0, 0, buf, nrows, rows, nw, -1.0f);
}
void launch_nr(int nr, dim3 grid, float *buf, size_t nrows, int rows, int nw)
{
    switch (nr) {
        case 0: launch<0>(grid, buf, nrows, rows, nw); break;
        case 14: launch<14>(grid, buf, nrows, rows, nw); break;
        case 25: launch<25>(grid, buf, nrows, rows, nw); break;
        case 50: launch<50>(grid, buf, nrows, rows, nw); break;
        default: launch<51>(grid, buf, nrows, rows, nw); break;
    }
}

int main()
{
    const int rows = 51;
    const size_t sizes[2] = {(size_t)209 << 20, (size_t)1 << 30};
    const struct { const char *name; int nr, nw; } modes[] = {
        {"read only", 51, 0}, {"write only", 0, 51}, {"in place (51 r + 51 w)", 51, 51}, {"check-like (51 r + 9 w)", 51, 9},
        {"var-like (14 r + 51 w)", 14, 51}, {"2 r : 1 w (50 r + 25 w)", 50, 25}, {"1 r : 2 w (25 r + 50 w)", 25, 50}};
    hipEvent_t a, b;
    HIPOK(hipEventCreate(&a));
    HIPOK(hipEventCreate(&b));
    for (size_t bytes : sizes) {
        const size_t nrows = bytes / 256 / rows * rows;
        float *buf;
        HIPOK(hipMalloc(&buf, nrows * 256));
        HIPOK(hipMemset(buf, 0, nrows * 256));
        const dim3 grid((unsigned)((nrows / rows + 3) / 4));
        printf("buffer %.0f MB\n", nrows * 256 / 1048576.0);
        for (const auto &m : modes) {
            const int reps = 20;
            for (int i = 0; i < 3; i++) launch_nr(m.nr, grid, buf, nrows, rows, m.nw);
            HIPOK(hipEventRecord(a, 0));
            for (int i = 0; i < reps; i++) launch_nr(m.nr, grid, buf, nrows, rows, m.nw);
            HIPOK(hipEventRecord(b, 0));
            HIPOK(hipEventSynchronize(b));
            float ms;
            HIPOK(hipEventElapsedTime(&ms, a, b));
            const double waves = (double)(nrows / rows), rd = waves * m.nr * 256.0 * reps, wr = waves * m.nw * 256.0 * reps, t = ms * 1e-3;
            printf("  %-26s reads %6.2f TB/s  writes %6.2f TB/s  total %6.2f TB/s   (%.1f us per launch)\n", m.name, rd / t / 1e12, wr / t / 1e12,
                   (rd + wr) / t / 1e12, ms * 1e3 / reps);
        }
        HIPOK(hipFree(buf));
    }
    return 0;
}

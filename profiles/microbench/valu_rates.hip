// Issue rates of the VALU instructions the q-ary min-plus kernels are made of or could be: v_add_f32, v_min_f32, v_min3_f32, the packed
// v_pk_add_f32 and the integer minima (non-negative floats order like their bit patterns), as wave-instructions per SIMD cycle with 1, 2, 4 and 8 waves per SIMD (independent chains of 16 registers).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(64) void k_rate(float *out, int iters, float seed)
{
    float x[16];
    f2 y[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = f2{x[2 * i], x[2 * i + 1]};
    const float c = seed * 0.5f;
    const f2 c2 = f2{c, c + 1.0f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(c));
            } else if (OP == 1) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_min_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(c));
            } else if (OP == 2) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(c), "v"(x[(i + 1) & 15]));
            } else if (OP == 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i]) : "v"(y[i]), "v"(c2));
            } else if (OP == 4) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(c));
            } else if (OP == 5) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(c), "v"(x[(i + 1) & 15]));
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_min3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(c), "v"(x[(i + 1) & 15]));
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    if (s == -1.0f) out[0] = s;
}

template <int OP>
double run(int waves_per_simd, float *out)
{
    const int iters = 2000;
    const int blocks = 256 * 4 * waves_per_simd;  // one wave per block
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 8 * (OP == 3 ? 8 : 16) * waves_per_simd;  // per SIMD
    return instr / (ms * 1e-3);                                                    // wave-instructions per second per SIMD
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    const char *names[7] = {"v_add_f32", "v_min_f32", "v_min3_f32", "v_pk_add_f32", "v_min_u32", "v_min3_u32", "v_min3_i32"};
    for (int w : {1, 2, 4, 8}) {
        const double r[7] = {run<0>(w, out), run<1>(w, out), run<2>(w, out), run<3>(w, out), run<4>(w, out), run<5>(w, out), run<6>(w, out)};
        printf("%d wave(s) per SIMD (G wave-instructions/s per SIMD):", w);
        for (int k = 0; k < 7; k++) printf("  %s %.2f", names[k], r[k] / 1e9);
        printf("\n");
    }
    return 0;
}

// The check-node kernels of DecoderSpecial at the Kyber shape (five symbols, six coefficient edges + the row-sum edge)
// compared MESSAGE FOR MESSAGE, as bit patterns:
//   host   a plain enumeration on the CPU in the reference's own form (decoder_special.rs:531-554 restated below: every
//          assignment forms S left to right and lowers beta_j[d_j] with S - a_j[d_j], f32::min semantics)
//   lane   k_q_special_check       the product's enumeration in the same form (codeword per lane)
//   tree   k_q_special_check_tree  the product's tree walk in min-marginal form
//   dp     k_q_special_check_dp    the product's min-plus recursion (no enumeration), whole row per lane and split over four waves
// all four kernels included from the product's header as it stands.  Inputs: smooth random LLRs over 20 binades (every
// addition rounds), heavy ties, impossible symbols (+inf), NaN alphas (the variable update's inf - inf), zeros, sums that
// overflow to +inf.  A difference is a bug in the kernel or a hole in the monotonicity argument of the header.
// Run by tests/test_qary_gpu.py::test_special_check_kernels_equal_the_enumeration_bit_for_bit.
// Build: make -C profiles/microbench qary_dp_equivalence
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
#include "../../sca-ldpc_amd/csrc/scaldpc_qary_special.h"

#define HIPOK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);           \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)

namespace {
constexpr int QB = 5, NB = 6, B = 2, BSUM = 12, QS = 2 * BSUM + 1, W = QS;
constexpr int R = 12, BATCH = 100;  // (a ragged batch: Bp = 128)
constexpr long Bp = 128;

u64 rng_state = 0x9E3779B97F4A7C15ull;
unsigned rnd()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(rng_state >> 33);
}
float unit() { return (float)(rnd() & 0xFFFFFF) / 16777216.0f; }

// one alpha of the given flavour
float draw(int flavour)
{
    switch (flavour) {
        case 0: return -logf(unit() + 1e-7f) * ldexpf(1.0f, (int)(rnd() % 20) - 10);  // smooth, 20 binades
        case 1: return 0.25f * (float)(rnd() % 8);                                    // ties everywhere
        case 2: return (rnd() % 4 == 0) ? INFINITY : -logf(unit() + 1e-7f);            // impossible symbols
        case 3: return (rnd() % 6 == 0) ? NAN : (rnd() % 6 == 0 ? INFINITY : 3.0f * unit());  // NaN alphas
        case 4: return (rnd() % 3 == 0) ? 0.0f : unit();                               // zeros (normalised messages have one)
        default: return (rnd() % 5 == 0) ? FLT_MAX * (0.3f + 0.5f * unit()) : 1e30f * unit();  // sums overflow
    }
}

// decoder_special.rs:531-554, one check of one codeword
void host_check(const float *a /* [NB][QB] */, const float *as /* [QS] */, float *bb, float *bs)
{
    for (int i = 0; i < NB * QB; i++) bb[i] = INFINITY;
    for (int i = 0; i < QS; i++) bs[i] = INFINITY;
    int d[NB] = {0, 0, 0, 0, 0, 0};  // digits q = d + B
    for (;;) {
        int dsum = 0;
        volatile float S = 0.0f;  // (volatile: every addition rounds to f32, whatever the host compiler would like)
        for (int j = 0; j < NB; j++) {
            dsum += d[j] - B;
            S = S + a[j * QB + d[j]];
        }
        const int t = -dsum + BSUM;
        S = S + as[t];
        for (int j = 0; j < NB; j++) {
            volatile float c = S - a[j * QB + d[j]];
            bb[j * QB + d[j]] = fminf(bb[j * QB + d[j]], c);
        }
        volatile float c = S - as[t];
        bs[t] = fminf(bs[t], c);
        int j = 0;
        for (; j < NB; j++) {
            if (d[j] < QB - 1) {
                d[j]++;
                break;
            }
            d[j] = 0;
        }
        if (j >= NB) break;
    }
}

size_t at(int c, int j, int q, long b) { return ((size_t)(c * (NB + 1) + j) * W + q) * Bp + b; }
}  // namespace

int main()
{
    const size_t n = (size_t)R * (NB + 1) * W * Bp;
    std::vector<int> row_ptr(R + 1);
    for (int c = 0; c <= R; c++) row_ptr[c] = c * (NB + 1);
    int *d_row_ptr;
    float *d_in, *d_work;
    HIPOK(hipMalloc(&d_row_ptr, sizeof(int) * (R + 1)));
    HIPOK(hipMalloc(&d_in, sizeof(float) * n));
    HIPOK(hipMalloc(&d_work, sizeof(float) * n));
    HIPOK(hipMemcpy(d_row_ptr, row_ptr.data(), sizeof(int) * (R + 1), hipMemcpyHostToDevice));
    const char *names[6] = {"smooth", "ties", "impossible", "nan", "zeros", "overflow"};
    int bad_total = 0;
    for (int flavour = 0; flavour < 6; flavour++) {
        std::vector<float> in(n, 7.0f), host(n), out[4];
        for (int c = 0; c < R; c++)
            for (long b = 0; b < BATCH; b++) {
                for (int j = 0; j < NB; j++)
                    for (int q = 0; q < QB; q++) in[at(c, j, q, b)] = draw(flavour);
                for (int q = 0; q < QS; q++) in[at(c, NB, q, b)] = draw(flavour);
            }
        host = in;
        for (int c = 0; c < R; c++)
            for (long b = 0; b < BATCH; b++) {
                float a[NB * QB], as[QS], bb[NB * QB], bs[QS];
                for (int j = 0; j < NB; j++)
                    for (int q = 0; q < QB; q++) a[j * QB + q] = in[at(c, j, q, b)];
                for (int q = 0; q < QS; q++) as[q] = in[at(c, NB, q, b)];
                host_check(a, as, bb, bs);
                for (int j = 0; j < NB; j++)
                    for (int q = 0; q < QB; q++) host[at(c, j, q, b)] = bb[j * QB + q];
                for (int q = 0; q < QS; q++) host[at(c, NB, q, b)] = bs[q];
            }
        for (int k = 0; k < 4; k++) {
            HIPOK(hipMemcpy(d_work, in.data(), sizeof(float) * n, hipMemcpyHostToDevice));
            if (k == 0) {
                const size_t lds = (size_t)2 * (NB * QB + QS) * 4 * 64;
                hipLaunchKernelGGL(k_q_special_check, dim3(R, Bp / 64), dim3(64), lds, 0, d_row_ptr, d_work, B, BSUM, W, Bp, BATCH, NB);
            } else if (k == 1) {
                const size_t lds = ((size_t)NB * QB + QS + (size_t)(NB * QB + QS) * 64) * 4;
                hipLaunchKernelGGL((k_q_special_check_tree<QB, NB>), dim3(R, BATCH), dim3(64), lds, 0, d_row_ptr, d_work, BSUM, W, Bp);
            } else if (k == 2)
                hipLaunchKernelGGL((k_q_special_check_dp<QB, NB, false>), dim3(R, Bp / 64), dim3(64), 0, 0, d_row_ptr, d_work, BSUM, W, Bp,
                                   BATCH);
            else
                hipLaunchKernelGGL((k_q_special_check_dp<QB, NB, true>), dim3(R, Bp / 64), dim3(256), 0, 0, d_row_ptr, d_work, BSUM, W, Bp,
                                   BATCH);
            HIPOK(hipGetLastError());
            HIPOK(hipDeviceSynchronize());
            out[k].resize(n);
            HIPOK(hipMemcpy(out[k].data(), d_work, sizeof(float) * n, hipMemcpyDeviceToHost));
        }
        long cnt = 0, diff[4] = {0, 0, 0, 0}, inf_out = 0;
        for (int c = 0; c < R; c++)
            for (long b = 0; b < BATCH; b++)
                for (int j = 0; j <= NB; j++)
                    for (int q = 0; q < (j < NB ? QB : QS); q++) {
                        const size_t i = at(c, j, q, b);
                        uint32_t h;
                        memcpy(&h, &host[i], 4);
                        cnt++;
                        inf_out += std::isinf(host[i]);
                        for (int k = 0; k < 4; k++) {
                            uint32_t g;
                            memcpy(&g, &out[k][i], 4);
                            if (g != h) {
                                if (diff[k]++ < 3)
                                    fprintf(stderr, "%s kernel %d: check %d codeword %ld edge %d symbol %d: %a (host) vs %a\n",
                                            names[flavour], k, c, b, j, q, host[i], out[k][i]);
                            }
                        }
                    }
        printf("CASE %-10s %ld messages (%ld of them +inf): %ld differ in lane, %ld differ in tree, %ld differ in dp, %ld differ in split dp\n",
               names[flavour], cnt, inf_out, diff[0], diff[1], diff[2], diff[3]);
        bad_total += (int)(diff[0] + diff[1] + diff[2] + diff[3] != 0);
    }
    hipFree(d_row_ptr);
    hipFree(d_in);
    hipFree(d_work);
    return bad_total ? 1 : 0;
}

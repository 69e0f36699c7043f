// The check-node kernels of DecoderSpecial at the Kyber shape (five symbols, six coefficient edges + the row-sum edge)
// compared MESSAGE FOR MESSAGE, as bit patterns:
//   host   a plain enumeration on the CPU in the reference's own form (decoder_special.rs:531-554 restated below: every
//          assignment forms S left to right and lowers beta_j[d_j] with S - a_j[d_j], f32::min semantics)
//   lane   k_q_special_check       the product's enumeration in the same form (codeword per lane)
//   tree   k_q_special_check_tree  the product's tree walk in min-marginal form
//   dp     k_q_special_check_dp    the product's min-plus recursion (no enumeration), whole row per lane and split over two / four waves
// all four kernels included from the product's header as it stands.  Inputs: smooth random LLRs over 20 binades (every
// addition rounds), heavy ties, impossible symbols (+inf), NaN alphas (the variable update's inf - inf), zeros, sums that
// overflow to +inf.  A difference is a bug in the kernel or a hole in the monotonicity argument of the header.
// Second part, Decoder's check update (decoder.rs:585-631) at Q = 3, rows of 1 .. 7 edges (config 4's decoder): a host
// enumeration over the finite supports in the reference's form against k_q_check_unrolled<3,7> (min-marginal enumeration) and
// k_q_check_dp<3,7> (clipped min-plus recursion), messages as bit patterns and the error code of the pass.
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
#include "../../sca-ldpc_amd/csrc/scaldpc_qary_rows.h"

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

// decoder.rs:585-631 for one check of k edges over Q = 3 symbols: finite supports of the first k - 1 edges enumerated, the last
// symbol follows from sum d = 0; returns 0, QERR_NO_FINITE (an edge without a finite symbol) or QERR_NO_CONFIG
constexpr int GQ = 3, GB = 1;
int host_check_generic(int k, const float *a /* [k][GQ] */, float *bt)
{
    for (int i = 0; i < k * GQ; i++) bt[i] = INFINITY;
    int fin[8][GQ], nfin[8];
    bool bad = false;
    for (int j = 0; j < k; j++) {
        nfin[j] = 0;
        for (int q = 0; q < GQ; q++)
            if (std::isfinite(a[j * GQ + q])) fin[j][nfin[j]++] = q;
        bad |= nfin[j] == 0;
    }
    if (bad) return QERR_NO_FINITE;
    int idx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nconf = 0;
    for (;;) {
        int dsum = 0, qs[8];
        volatile float S = 0.0f;
        for (int j = 0; j < k - 1; j++) {
            qs[j] = fin[j][idx[j]];
            dsum += qs[j] - GB;
            S = S + a[j * GQ + qs[j]];
        }
        const int dl = -dsum;
        if (dl >= -GB && dl <= GB) {
            qs[k - 1] = dl + GB;
            S = S + a[(k - 1) * GQ + qs[k - 1]];
            if (std::isfinite(S)) {
                nconf++;
                for (int j = 0; j < k; j++) {
                    volatile float c = S - a[j * GQ + qs[j]];
                    bt[j * GQ + qs[j]] = fminf(c, bt[j * GQ + qs[j]]);
                }
            }
        }
        int j = 0;
        for (; j < k - 1; j++) {
            if (idx[j] + 1 < nfin[j]) {
                idx[j]++;
                break;
            }
            idx[j] = 0;
        }
        if (j >= k - 1) break;
    }
    return nconf ? 0 : QERR_NO_CONFIG;
}
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
        std::vector<float> in(n, 7.0f), host(n), out[5];
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
        for (int k = 0; k < 5; k++) {
            HIPOK(hipMemcpy(d_work, in.data(), sizeof(float) * n, hipMemcpyHostToDevice));
            if (k == 0) {
                const size_t lds = (size_t)2 * (NB * QB + QS) * 4 * 64;
                hipLaunchKernelGGL(k_q_special_check, dim3(R, Bp / 64), dim3(64), lds, 0, d_row_ptr, d_work, B, BSUM, W, Bp, BATCH, NB);
            } else if (k == 1) {
                const size_t lds = ((size_t)NB * QB + QS + (size_t)(NB * QB + QS) * 64) * 4;
                hipLaunchKernelGGL((k_q_special_check_tree<QB, NB>), dim3(R, BATCH), dim3(64), lds, 0, d_row_ptr, d_work, BSUM, W, Bp);
            } else if (k == 2)
                hipLaunchKernelGGL((k_q_special_check_dp<QB, NB, 1>), dim3(R, Bp / 64), dim3(64), 0, 0, d_row_ptr, d_work, BSUM, W, Bp,
                                   BATCH);
            else if (k == 3)
                hipLaunchKernelGGL((k_q_special_check_dp<QB, NB, 4>), dim3(R, Bp / 64), dim3(256), 0, 0, d_row_ptr, d_work, BSUM, W, Bp,
                                   BATCH);
            else
                hipLaunchKernelGGL((k_q_special_check_dp<QB, NB, 2>), dim3(R, Bp / 64), dim3(128), 0, 0, d_row_ptr, d_work, BSUM, W, Bp,
                                   BATCH);
            HIPOK(hipGetLastError());
            HIPOK(hipDeviceSynchronize());
            out[k].resize(n);
            HIPOK(hipMemcpy(out[k].data(), d_work, sizeof(float) * n, hipMemcpyDeviceToHost));
        }
        long cnt = 0, diff[5] = {0, 0, 0, 0, 0}, inf_out = 0;
        for (int c = 0; c < R; c++)
            for (long b = 0; b < BATCH; b++)
                for (int j = 0; j <= NB; j++)
                    for (int q = 0; q < (j < NB ? QB : QS); q++) {
                        const size_t i = at(c, j, q, b);
                        uint32_t h;
                        memcpy(&h, &host[i], 4);
                        cnt++;
                        inf_out += std::isinf(host[i]);
                        for (int k = 0; k < 5; k++) {
                            uint32_t g;
                            memcpy(&g, &out[k][i], 4);
                            if (g != h) {
                                if (diff[k]++ < 3)
                                    fprintf(stderr, "%s kernel %d: check %d codeword %ld edge %d symbol %d: %a (host) vs %a\n",
                                            names[flavour], k, c, b, j, q, host[i], out[k][i]);
                            }
                        }
                    }
        printf("CASE %-10s %ld messages (%ld of them +inf): %ld differ in lane, %ld differ in tree, %ld differ in dp, %ld differ in split dp, %ld differ in half-split dp\n",
               names[flavour], cnt, inf_out, diff[0], diff[1], diff[2], diff[3], diff[4]);
        bad_total += (int)(diff[0] + diff[1] + diff[2] + diff[3] + diff[4] != 0);
    }
    // ---- Decoder (Q = 3), rows of 1 .. 7 edges ----
    {
        const int degs[14] = {1, 2, 3, 4, 5, 6, 7, 7, 7, 6, 5, 4, 3, 2};
        const int GR = 14;
        std::vector<int> rp(GR + 1, 0);
        for (int c = 0; c < GR; c++) rp[c + 1] = rp[c] + degs[c];
        const int GE = rp[GR];
        const size_t gn = (size_t)GE * GQ * Bp;
        int *d_rp, *d_err;
        float *d_g;
        HIPOK(hipMalloc(&d_rp, sizeof(int) * (GR + 1)));
        HIPOK(hipMalloc(&d_err, sizeof(int)));
        HIPOK(hipMalloc(&d_g, sizeof(float) * gn));
        HIPOK(hipMemcpy(d_rp, rp.data(), sizeof(int) * (GR + 1), hipMemcpyHostToDevice));
        auto gat = [&](int e, int q, long b) { return ((size_t)e * GQ + q) * Bp + b; };
        for (int flavour = 0; flavour < 6; flavour++) {
            std::vector<float> in(gn, 7.0f), host, out[2];
            for (int e = 0; e < GE; e++)
                for (int q = 0; q < GQ; q++)
                    for (long b = 0; b < BATCH; b++) in[gat(e, q, b)] = draw(flavour);
            host = in;
            int host_err = 0;
            for (int c = 0; c < GR; c++)
                for (long b = 0; b < BATCH; b++) {
                    float a[8 * GQ], bt[8 * GQ];
                    const int k = degs[c];
                    for (int j = 0; j < k; j++)
                        for (int q = 0; q < GQ; q++) a[j * GQ + q] = in[gat(rp[c] + j, q, b)];
                    const int st = host_check_generic(k, a, bt);
                    if (st > host_err) host_err = st;
                    // (a row the reference refuses -- it asserts or would not terminate -- has no messages to compare)
                    for (int j = 0; j < k; j++)
                        for (int q = 0; q < GQ; q++) host[gat(rp[c] + j, q, b)] = st == QERR_NO_FINITE ? NAN : bt[j * GQ + q];
                }
            int errs[2] = {0, 0};
            for (int kk = 0; kk < 2; kk++) {
                HIPOK(hipMemcpy(d_g, in.data(), sizeof(float) * gn, hipMemcpyHostToDevice));
                HIPOK(hipMemset(d_err, 0, sizeof(int)));
                if (kk == 0)
                    hipLaunchKernelGGL((k_q_check_unrolled<3, 7>), dim3(GR, Bp / 64), dim3(64), 0, 0, d_rp, d_g, Bp, BATCH, d_err);
                else
                    hipLaunchKernelGGL((k_q_check_dp<3, 7>), dim3(GR, Bp / 64), dim3(64), 0, 0, d_rp, d_g, Bp, BATCH, d_err);
                HIPOK(hipGetLastError());
                HIPOK(hipDeviceSynchronize());
                out[kk].resize(gn);
                HIPOK(hipMemcpy(out[kk].data(), d_g, sizeof(float) * gn, hipMemcpyDeviceToHost));
                HIPOK(hipMemcpy(&errs[kk], d_err, sizeof(int), hipMemcpyDeviceToHost));
            }
            long cnt = 0, diff[2] = {0, 0}, skipped = 0;
            for (int e = 0; e < GE; e++)
                for (int q = 0; q < GQ; q++)
                    for (long b = 0; b < BATCH; b++) {
                        const size_t i = gat(e, q, b);
                        if (std::isnan(host[i])) {
                            skipped++;
                            continue;
                        }
                        uint32_t h;
                        memcpy(&h, &host[i], 4);
                        cnt++;
                        for (int kk = 0; kk < 2; kk++) {
                            uint32_t g;
                            memcpy(&g, &out[kk][i], 4);
                            if (g != h && diff[kk]++ < 3)
                                fprintf(stderr, "generic %s kernel %d: edge %d symbol %d codeword %ld: %a (host) vs %a\n", names[flavour], kk, e, q, b,
                                        host[i], out[kk][i]);
                        }
                    }
            printf("GENERIC %-10s %ld messages (%ld more in rows the reference refuses), error code %d: %ld differ in unrolled (code %d), "
                   "%ld differ in dp (code %d)\n",
                   names[flavour], cnt, skipped, host_err, diff[0], errs[0], diff[1], errs[1]);
            bad_total += (int)(diff[0] + diff[1] != 0 || errs[0] != host_err || errs[1] != host_err);
        }
        hipFree(d_rp);
        hipFree(d_err);
        hipFree(d_g);
    }
    hipFree(d_row_ptr);
    hipFree(d_in);
    hipFree(d_work);
    return bad_total ? 1 : 0;
}

/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file (oracle/liboracle.so).
 *
 * CPU restatement of the binary BP decoder the reference calls
 * (`ldpc.bp_decoder(...).decode(v)`, simulate/decode.py:155-161,171;
 * simulate/hqc.py:694-699,708).  See bp_oracle_impl.h for the algorithm and its
 * parity status (hard decisions pinned by the reference's doctests; posterior
 * LLRs "parity unpinned").
 *
 * Two instantiations:
 *   *_f64  float64, as the reference package computes;
 *   *_f32  the same operation order in float32 = what the HIP kernels compute
 *          (min-sum: bit-exact target; tanh rule: tolerance target).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL double
#define SFX f64
#define RLOG log
#define REXP exp
#define RFMA fma
#define RTANH tanh
#define RABS fabs
#define RBIG 1e308
#define RTINY 0.0 /* no flush in the float64 instantiation */
#include "bp_oracle_impl.h"
#undef REAL
#undef SFX
#undef RLOG
#undef REXP
#undef RFMA
#undef RTANH
#undef RABS
#undef RBIG
#undef RTINY

#define REAL float
#define SFX f32
#define RLOG logf
#define REXP expf
#define RFMA fmaf
#define RTANH tanhf
#define RABS fabsf
#define RBIG FLT_MAX
#define RTINY FLT_MIN /* method 3 mirrors the kernel: reciprocals below FLT_MIN are flushed (hardware v_rcp_f32) */
#include "bp_oracle_impl.h"
#undef REAL
#undef SFX
#undef RLOG
#undef REXP
#undef RFMA
#undef RTANH
#undef RABS
#undef RBIG
#undef RTINY

/*
 * Batch front end with the `decode(v)` input convention of the reference's
 * package: mode 0 = syndrome (len m), result = e;
 * mode 1 = received vector v (len n), s = H v mod 2, result = e XOR v.
 * in:  uint8 [batch][m or n];  out_bits: uint8 [batch][n];
 * out_llr: [batch][n] (log(p0/p1));  out_iter, out_conv: int32 [batch].
 * threads <= 0 -> 1.
 */
#define BATCH_FN(SFX, REAL)                                                                        \
    int oracle_bp_decode_batch_##SFX(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, \
                                     const int32_t *col_ptr, const int32_t *csc_edge,              \
                                     const double *channel_probs, const uint8_t *in, int mode,     \
                                     int batch, int max_iter, int method, double alpha,            \
                                     uint8_t *out_bits, REAL *out_llr, int32_t *out_iter,          \
                                     int32_t *out_conv, int threads, int early_exit)               \
    {                                                                                              \
        if (mode != 0 && mode != 1) return -2;                                                     \
        const int in_len = mode ? n : m;                                                           \
        const int nnz = row_ptr[m];                                                                \
        int rc = 0;                                                                                \
        if (threads <= 0) threads = 1;                                                             \
        _Pragma("omp parallel num_threads(threads)")                                               \
        {                                                                                          \
            REAL *work = (REAL *)malloc(sizeof(REAL) * 2 * (size_t)(nnz > 0 ? nnz : 1));           \
            uint8_t *synd = (uint8_t *)malloc((size_t)(m > 0 ? m : 1));                            \
            _Pragma("omp for schedule(dynamic, 1)")                                                \
            for (int b = 0; b < batch; b++) {                                                      \
                const uint8_t *v = in + (size_t)b * in_len;                                        \
                if (mode == 0) {                                                                   \
                    for (int i = 0; i < m; i++) synd[i] = v[i] & 1;                                \
                } else {                                                                           \
                    for (int i = 0; i < m; i++) {                                                  \
                        int par = 0;                                                               \
                        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) par ^= v[col_idx[e]] & 1;\
                        synd[i] = (uint8_t)par;                                                    \
                    }                                                                              \
                }                                                                                  \
                int r = oracle_bp_decode_##SFX(m, n, row_ptr, col_idx, col_ptr, csc_edge,          \
                                               channel_probs, synd, max_iter, method, alpha,       \
                                               out_bits + (size_t)b * n, out_llr + (size_t)b * n,  \
                                               out_iter + b, out_conv + b, work, early_exit);      \
                if (r) rc = r;                                                                     \
                if (mode == 1)                                                                     \
                    for (int j = 0; j < n; j++) out_bits[(size_t)b * n + j] ^= v[j] & 1;           \
            }                                                                                      \
            free(work);                                                                            \
            free(synd);                                                                            \
        }                                                                                          \
        return rc;                                                                                 \
    }

BATCH_FN(f64, double)
BATCH_FN(f32, float)

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

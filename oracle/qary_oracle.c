/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file (oracle/liboracle.so).
 *
 * CPU restatement, in f32 and in the reference's operation order, of the q-ary
 * min-sum decoders of the in-tree Rust crate `simulate_rs` (cannot be compiled
 * here: no rustc/cargo, and the crate depends on an empty submodule):
 *
 *   oracle_qary_into_llr      decoder.rs:668-692   (== decoder_special.rs:619-643)
 *   oracle_qary_min_sum       decoder.rs:560-666   Decoder::min_sum
 *   oracle_qary_special       decoder_special.rs:471-617   DecoderSpecial::min_sum
 *
 * Graph: CSR (check -> its variables in ascending column = decoder.rs:507-539
 * fill order) with h in {-1,+1}; CSC permutation (variable -> its edges in
 * ascending row).  Symbols q in [0,Q) stand for values q-B.
 *
 * Pinned by the reference's own known-answer tests (tests/test_oracle_pins.py):
 * decoder.rs:744-768 (into_llr exact values), :771-799 (6x3, Q=15, 10 it),
 * :819-854 (150x450 + benches/parity_check_150_450.txt), decode.py:192-209.
 * decoder_special.rs has no live test (its tests are commented out, :691-746):
 * oracle_qary_special is "parity unpinned" by the reference.  ...pinned on cycle-free
 * graphs by tests/test_exact_inference*.py: on tree-shaped [H' | +-I] (B = 2, BSUM = 12)
 * its decision, and the HIP path's, is the enumerated minimum-cost valid assignment;
 * the same holds for oracle_qary_min_sum on random +-1 trees with B = 1, 2, up to 250
 * variables against exact min-marginals ((min,+) elimination, tests/exact.tree_exact_qary).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QERR_PMF_SUM (-3)      /* decoder.rs:683-684 assert */
#define QERR_NO_MAX (-4)       /* decoder.rs:680 expect */
#define QERR_NO_FINITE (-5)    /* decoder.rs:368-375 would spin forever */
#define QERR_NO_CONFIG (-6)    /* decoder.rs:618 assert */
#define QERR_SHAPE (-7)

/* decoder.rs:668-692 */
int oracle_qary_into_llr(int nvars, int Q, const float *pmf, float *llr)
{
    const float EPSILON = 0.001f;
    for (int v = 0; v < nvars; v++) {
        const float *p = pmf + (size_t)v * Q;
        float sum = 0.0f;
        for (int q = 0; q < Q; q++) sum += p[q];
        int have = 0;
        float mx = 0.0f;
        for (int q = 0; q < Q; q++) {
            if (p[q] != p[q]) continue; /* NotNan::new filter */
            if (!have || p[q] > mx) { mx = p[q]; have = 1; }
        }
        if (!have) return QERR_NO_MAX;
        if (!(sum < 1.0f + EPSILON)) return QERR_PMF_SUM;
        if (!(sum > 1.0f - EPSILON)) return QERR_PMF_SUM;
        for (int q = 0; q < Q; q++) llr[(size_t)v * Q + q] = logf(mx / p[q]);
    }
    return 0;
}

/* decoder.rs:694-704: first strict minimum, NaN never selected, default 0 */
static int arg_min(const float *m, int Q)
{
    float mv = INFINITY;
    int ma = 0;
    for (int q = 0; q < Q; q++)
        if (m[q] < mv) { mv = m[q]; ma = q; }
    return ma;
}

/* variable-node update for one variable of alphabet size Q (decoder.rs:634-658) */
static int var_update(int Q, const float *channel, int deg, const int32_t *edges, const int8_t *val,
                      const float *c2v, float *v2c, float *sum /* Q scratch */, float *tmp)
{
    for (int q = 0; q < Q; q++) sum[q] = channel[q];
    for (int t = 0; t < deg; t++) {
        const float *in = c2v + (size_t)edges[t] * Q;
        if (val[edges[t]] > 0)
            for (int q = 0; q < Q; q++) sum[q] = sum[q] + in[q];
        else
            for (int q = 0; q < Q; q++) sum[q] = sum[q] + in[Q - q - 1];
    }
    for (int t = 0; t < deg; t++) {
        const int e = edges[t];
        const float *in = c2v + (size_t)e * Q;
        float *out = v2c + (size_t)e * Q;
        if (val[e] > 0) {
            for (int q = 0; q < Q; q++) tmp[q] = sum[q] - in[q];
        } else {
            /* (sum - rev(in)) then reversed again */
            for (int q = 0; q < Q; q++) tmp[Q - q - 1] = sum[q] - in[Q - q - 1];
        }
        int am = arg_min(tmp, Q);
        float mn = tmp[am];
        for (int q = 0; q < Q; q++) out[q] = tmp[q] - mn;
    }
    return arg_min(sum, Q);
}

/* Decoder::min_sum, decoder.rs:560-666.  llr: [N][Q]; out: int8 [N]. */
int oracle_qary_min_sum(int R, int N, int Q, const int32_t *row_ptr, const int32_t *col_idx,
                        const int8_t *val, const int32_t *col_ptr, const int32_t *csc_edge,
                        const float *llr, int max_iter, int8_t *out)
{
    if (Q < 1 || (Q & 1) == 0) return QERR_SHAPE;
    const int B = (Q - 1) / 2;
    const int nnz = row_ptr[R];
    int maxdc = 0;
    for (int c = 0; c < R; c++)
        if (row_ptr[c + 1] - row_ptr[c] > maxdc) maxdc = row_ptr[c + 1] - row_ptr[c];
    float *v2c = (float *)malloc(sizeof(float) * (size_t)(nnz + 1) * Q);
    float *c2v = (float *)malloc(sizeof(float) * (size_t)(nnz + 1) * Q);
    int *fin = (int *)malloc(sizeof(int) * (size_t)(maxdc + 1) * Q);
    int *num = (int *)malloc(sizeof(int) * (size_t)(maxdc + 1));
    int *idx = (int *)malloc(sizeof(int) * (size_t)(maxdc + 1));
    int *dv = (int *)malloc(sizeof(int) * (size_t)(maxdc + 1));
    float *sum = (float *)malloc(sizeof(float) * 2 * (size_t)Q);
    int rc = 0;
    if (max_iter < 1) max_iter = 1; /* the loop body runs at least once (it += 1 first) */

    /* 0. init, decoder.rs:567-573 */
    for (int v = 0; v < N; v++)
        for (int t = col_ptr[v]; t < col_ptr[v + 1]; t++) {
            int e = csc_edge[t];
            for (int q = 0; q < Q; q++)
                v2c[(size_t)e * Q + q] = (val[e] < 0) ? llr[(size_t)v * Q + (Q - q - 1)] : llr[(size_t)v * Q + q];
        }

    for (int it = 1; it <= max_iter && !rc; it++) {
        /* 3. check node update, decoder.rs:585-631 */
        for (int c = 0; c < R && !rc; c++) {
            const int e0 = row_ptr[c], k = row_ptr[c + 1] - row_ptr[c];
            if (k == 0) { rc = QERR_NO_CONFIG; break; }
            for (int j = 0; j < k; j++) {
                num[j] = 0;
                for (int q = 0; q < Q; q++)
                    if (isfinite(v2c[(size_t)(e0 + j) * Q + q])) fin[j * Q + num[j]++] = q - B;
                if (num[j] == 0) rc = QERR_NO_FINITE;
                for (int q = 0; q < Q; q++) c2v[(size_t)(e0 + j) * Q + q] = INFINITY;
            }
            if (rc) break;
            int nconf = 0;
            for (int j = 0; j < k; j++) idx[j] = 0;
            for (;;) {
                int dsum = 0;
                for (int j = 0; j < k - 1; j++) { dv[j] = fin[j * Q + idx[j]]; dsum += dv[j]; }
                dv[k - 1] = -dsum;
                if (dv[k - 1] >= -B && dv[k - 1] <= B) {
                    float S = 0.0f;
                    for (int j = 0; j < k; j++) S += v2c[(size_t)(e0 + j) * Q + (dv[j] + B)];
                    if (isfinite(S)) {
                        nconf++;
                        for (int j = 0; j < k; j++) {
                            float *b = &c2v[(size_t)(e0 + j) * Q + (dv[j] + B)];
                            *b = fminf(S - v2c[(size_t)(e0 + j) * Q + (dv[j] + B)], *b);
                        }
                    }
                }
                /* increment, index 0 fastest (decoder.rs:353-368) */
                int j = 0;
                for (; j < k - 1; j++) {
                    if (idx[j] + 1 < num[j]) { idx[j]++; break; }
                    idx[j] = 0;
                }
                if (j >= k - 1) break;
            }
            if (nconf == 0) rc = QERR_NO_CONFIG;
        }
        if (rc) break;
        /* 4-6. variable node update, decoder.rs:634-658 */
        for (int v = 0; v < N; v++) {
            int am = var_update(Q, llr + (size_t)v * Q, col_ptr[v + 1] - col_ptr[v], csc_edge + col_ptr[v], val,
                                c2v, v2c, sum, sum + Q);
            if (it >= max_iter) out[v] = (int8_t)(am - B);
        }
    }
    free(v2c); free(c2v); free(fin); free(num); free(idx); free(dv); free(sum);
    return rc;
}

/*
 * DecoderSpecial::min_sum, decoder_special.rs:471-617.  H = [H' | I]: the first
 * N-R columns are B-variables (alphabet 2B+1), the last R columns are the
 * row-sum variables (alphabet 2*BSUM+1, one per check, its LAST entry).
 * llr_b: [N-R][2B+1]; llr_s: [R][2BSUM+1]; out: int8 [N].
 */
int oracle_qary_special(int R, int N, int B, int BSUM, const int32_t *row_ptr, const int32_t *col_idx,
                        const int8_t *val, const int32_t *col_ptr, const int32_t *csc_edge,
                        const float *llr_b, const float *llr_s, int max_iter, int8_t *out)
{
    const int BV = N - R, QB = 2 * B + 1, QS = 2 * BSUM + 1;
    if (B < 1 || BSUM % B != 0) return QERR_SHAPE; /* decoder_special.rs:388-392 */
    const int nnz = row_ptr[R];
    /* edge storage: every edge gets QS floats of room (simple, it is an oracle) */
    const size_t W = (size_t)(QS > QB ? QS : QB);
    float *v2c = (float *)malloc(sizeof(float) * (size_t)(nnz + 1) * W);
    float *c2v = (float *)malloc(sizeof(float) * (size_t)(nnz + 1) * W);
    float *sum = (float *)malloc(sizeof(float) * 2 * W);
    int rc = 0;
    if (max_iter < 1) max_iter = 1;
    int dv[64];

    for (int c = 0; c < R; c++) {
        int k = row_ptr[c + 1] - row_ptr[c];
        if (k < 1 || k > 63) { rc = QERR_SHAPE; break; }
        for (int j = 0; j < k - 1; j++)
            if (col_idx[row_ptr[c] + j] >= BV) rc = QERR_SHAPE;
        if (col_idx[row_ptr[c] + k - 1] < BV) rc = QERR_SHAPE;
        if ((k - 1) * B > BSUM) rc = QERR_SHAPE; /* index would leave the BSUM alphabet */
    }
    for (int v = BV; v < N && !rc; v++)
        if (col_ptr[v + 1] - col_ptr[v] != 1) rc = QERR_SHAPE; /* VariableNode<1,..>, :316 */
    if (rc) goto done;

    /* 0. init, :480-493 */
    for (int v = 0; v < N; v++) {
        const int Q = v < BV ? QB : QS;
        const float *ch = v < BV ? llr_b + (size_t)v * QB : llr_s + (size_t)(v - BV) * QS;
        for (int t = col_ptr[v]; t < col_ptr[v + 1]; t++) {
            int e = csc_edge[t];
            for (int q = 0; q < Q; q++) v2c[(size_t)e * W + q] = (val[e] < 0) ? ch[Q - q - 1] : ch[q];
        }
    }

    for (int it = 1; it <= max_iter; it++) {
        /* 3. check node update, :506-563 */
        for (int c = 0; c < R; c++) {
            const int e0 = row_ptr[c], k = row_ptr[c + 1] - row_ptr[c], nb = k - 1;
            const float *as = v2c + (size_t)(e0 + nb) * W;
            float *bs = c2v + (size_t)(e0 + nb) * W;
            for (int j = 0; j < nb; j++)
                for (int q = 0; q < QB; q++) c2v[(size_t)(e0 + j) * W + q] = INFINITY;
            for (int q = 0; q < QS; q++) bs[q] = INFINITY;
            for (int j = 0; j < nb; j++) dv[j] = -B;
            for (;;) {
                int dsum = 0;
                for (int j = 0; j < nb; j++) dsum += dv[j];
                dsum = -dsum;
                float S = 0.0f;
                for (int j = 0; j < nb; j++) S += v2c[(size_t)(e0 + j) * W + (dv[j] + B)];
                S += as[dsum + BSUM];
                for (int j = 0; j < nb; j++) {
                    float *b = &c2v[(size_t)(e0 + j) * W + (dv[j] + B)];
                    *b = fminf(*b, S - v2c[(size_t)(e0 + j) * W + (dv[j] + B)]);
                }
                bs[dsum + BSUM] = fminf(bs[dsum + BSUM], S - as[dsum + BSUM]);
                int j = 0;
                for (; j < nb; j++) {
                    if (dv[j] < B) { dv[j]++; break; }
                    dv[j] = -B;
                }
                if (j >= nb) break;
            }
        }
        /* variable updates, :566-609 -- edges use stride W, so go through a gather */
        for (int v = 0; v < N; v++) {
            const int Q = v < BV ? QB : QS;
            const float *ch = v < BV ? llr_b + (size_t)v * QB : llr_s + (size_t)(v - BV) * QS;
            const int deg = col_ptr[v + 1] - col_ptr[v];
            const int32_t *edges = csc_edge + col_ptr[v];
            float *tmp = sum + W;
            for (int q = 0; q < Q; q++) sum[q] = ch[q];
            for (int t = 0; t < deg; t++) {
                const float *in = c2v + (size_t)edges[t] * W;
                if (val[edges[t]] > 0)
                    for (int q = 0; q < Q; q++) sum[q] = sum[q] + in[q];
                else
                    for (int q = 0; q < Q; q++) sum[q] = sum[q] + in[Q - q - 1];
            }
            for (int t = 0; t < deg; t++) {
                const int e = edges[t];
                const float *in = c2v + (size_t)e * W;
                float *o = v2c + (size_t)e * W;
                if (val[e] > 0)
                    for (int q = 0; q < Q; q++) tmp[q] = sum[q] - in[q];
                else
                    for (int q = 0; q < Q; q++) tmp[Q - q - 1] = sum[q] - in[Q - q - 1];
                int am = arg_min(tmp, Q);
                float mn = tmp[am];
                for (int q = 0; q < Q; q++) o[q] = tmp[q] - mn;
            }
            if (it >= max_iter) out[v] = (int8_t)(arg_min(sum, Q) - (v < BV ? B : BSUM));
        }
    }
done:
    free(v2c); free(c2v); free(sum);
    return rc;
}

/* batch front ends: pmf in, hard decisions out; threads over the batch */
int oracle_qary_min_sum_batch(int R, int N, int Q, const int32_t *row_ptr, const int32_t *col_idx,
                              const int8_t *val, const int32_t *col_ptr, const int32_t *csc_edge,
                              const float *pmf /* [batch][N][Q] */, int batch, int max_iter,
                              int8_t *out /* [batch][N] */, int threads)
{
    int rc = 0;
    if (threads <= 0) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        float *llr = (float *)malloc(sizeof(float) * (size_t)N * Q);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; b++) {
            int r = oracle_qary_into_llr(N, Q, pmf + (size_t)b * N * Q, llr);
            if (!r)
                r = oracle_qary_min_sum(R, N, Q, row_ptr, col_idx, val, col_ptr, csc_edge, llr, max_iter,
                                        out + (size_t)b * N);
            if (r) rc = r;
        }
        free(llr);
    }
    return rc;
}

int oracle_qary_special_batch(int R, int N, int B, int BSUM, const int32_t *row_ptr, const int32_t *col_idx,
                              const int8_t *val, const int32_t *col_ptr, const int32_t *csc_edge,
                              const float *pmf_b /* [batch][N-R][2B+1] */,
                              const float *pmf_s /* [batch][R][2BSUM+1] */, int batch, int max_iter,
                              int8_t *out, int threads)
{
    int rc = 0;
    const int BV = N - R, QB = 2 * B + 1, QS = 2 * BSUM + 1;
    if (threads <= 0) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        float *lb = (float *)malloc(sizeof(float) * (size_t)BV * QB);
        float *ls = (float *)malloc(sizeof(float) * (size_t)R * QS);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; b++) {
            int r = oracle_qary_into_llr(BV, QB, pmf_b + (size_t)b * BV * QB, lb);
            if (!r) r = oracle_qary_into_llr(R, QS, pmf_s + (size_t)b * R * QS, ls);
            if (!r)
                r = oracle_qary_special(R, N, B, BSUM, row_ptr, col_idx, val, col_ptr, csc_edge, lb, ls,
                                        max_iter, out + (size_t)b * N);
            if (r) rc = r;
        }
        free(lb);
        free(ls);
    }
    return rc;
}

/* Plain-C client of libscaldpc (no Python, no torch): builds the repetition-code decoder of
 * the reference's "official example" (main.py:265-276), decodes a batch of syndromes with
 * early exit, and a tiny q-ary instance (decoder.rs:771-799).  Used by tests/test_c_abi_gpu.py.
 * build: gcc -O2 -I include tests/c_abi/demo.c -o demo -L sca-ldpc_amd -lscaldpc -Wl,-rpath,... */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "scaldpc.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, scaldpc_last_error()); return 1; } } while (0)

int main(void)
{
    /* rep_code(13): row i = {i, i+1} */
    enum { N = 13, M = 12, BATCH = 70 };
    int32_t row_ptr[M + 1], col_idx[2 * M];
    for (int i = 0; i <= M; i++) row_ptr[i] = 2 * i;
    for (int i = 0; i < M; i++) { col_idx[2 * i] = i; col_idx[2 * i + 1] = i + 1; }
    scaldpc_bp *h = NULL;
    CHECK(scaldpc_bp_create(M, N, 2 * M, row_ptr, col_idx, &h));
    double probs[N];
    for (int j = 0; j < N; j++) probs[j] = 0.05;
    CHECK(scaldpc_bp_set_channel_probs(h, probs));
    /* codeword b carries a single error at position b % 13 */
    uint8_t synd[BATCH][M], bits[BATCH][N], conv[BATCH];
    int32_t iters[BATCH];
    memset(synd, 0, sizeof synd);
    for (int b = 0; b < BATCH; b++) {
        int e = b % N;
        if (e > 0) synd[b][e - 1] ^= 1;
        if (e < M) synd[b][e] ^= 1;
    }
    CHECK(scaldpc_bp_decode_batch(h, &synd[0][0], SCALDPC_IN_SYNDROME, BATCH, N, SCALDPC_BP_PRODUCT_SUM, 1.0f,
                                  SCALDPC_F_EARLY_EXIT, NULL, &bits[0][0], NULL, iters, conv));
    int ok = 0;
    for (int b = 0; b < BATCH; b++) {
        int good = conv[b] == 1;
        for (int j = 0; j < N; j++) good &= bits[b][j] == (j == b % N);
        ok += good;
    }
    printf("bp: %d/%d single errors corrected, iters[0]=%d\n", ok, BATCH, iters[0]);
    /* bad arguments come back as codes, not crashes */
    int rc = scaldpc_bp_decode_batch(h, &synd[0][0], 7, BATCH, N, SCALDPC_BP_MIN_SUM, 1.0f, 0, NULL, &bits[0][0], NULL, NULL, NULL);
    printf("bad input kind -> %d (%s)\n", rc, scaldpc_last_error());
    scaldpc_bp_destroy(h);

    /* a graph that grows (hqc.py:885-908): rep_code(7) built from its first 3 checks, the other 3 appended
     * -- bringing columns 4..6 -- then decoded with the tile kernels (whose tables are rebuilt on demand) */
    {
        enum { N2 = 7, M2 = 6, HALF = 3 };
        int32_t rp[M2 + 1], ci[2 * M2];
        for (int i = 0; i <= M2; i++) rp[i] = 2 * i;
        for (int i = 0; i < M2; i++) { ci[2 * i] = i; ci[2 * i + 1] = i + 1; }
        scaldpc_bp *g = NULL;
        CHECK(scaldpc_bp_create(HALF, HALF + 1, 2 * HALF, rp, ci, &g));
        double p4[HALF + 1] = {0.05, 0.05, 0.05, 0.05}, ptail[N2 - HALF - 1] = {0.05, 0.05, 0.05};
        CHECK(scaldpc_bp_set_channel_probs(g, p4));
        int32_t rp_new[HALF + 1] = {0, 2, 4, 6};
        CHECK(scaldpc_bp_append_rows(g, HALF, rp_new, ci + 2 * HALF, N2));
        CHECK(scaldpc_bp_set_channel_probs_tail(g, HALF + 1, N2 - HALF - 1, ptail));
        CHECK(scaldpc_bp_configure(g, "path", "stream"));
        uint8_t s2[M2] = {0, 0, 0, 1, 1, 0}, b2[N2]; /* one error at position 4 */
        CHECK(scaldpc_bp_decode_batch(g, s2, SCALDPC_IN_SYNDROME, 1, N2, SCALDPC_BP_MIN_SUM, 1.0f, SCALDPC_F_EARLY_EXIT, NULL,
                                      b2, NULL, NULL, NULL));
        int good = 1;
        for (int j = 0; j < N2; j++) good &= b2[j] == (j == 4);
        int32_t where[4];
        CHECK(scaldpc_bp_device_of(g, where));
        printf("append: error at 4 %s, handle on device %d, graph on %d\n", good ? "found" : "MISSED", where[0], where[1]);
        scaldpc_bp_destroy(g);
        if (!good || where[0] != where[1]) return 3;
    }

    /* q-ary: decoder.rs:771-799, one bad symbol, Q = 15 */
    const int8_t H[3][6] = {{1, 1, 1, 1, 0, 0}, {0, 0, 1, 1, 0, 1}, {1, 0, 0, 1, 1, 0}};
    scaldpc_qary *q = NULL;
    CHECK(scaldpc_qary_create(3, 6, 7, &H[0][0], 10, &q));
    float pmf[6][15];
    memset(pmf, 0, sizeof pmf);
    for (int v = 0; v < 6; v++) pmf[v][7] = 1.0f;
    pmf[1][7] = 0.1f;
    pmf[1][14] = 0.9f;
    int8_t out[6];
    CHECK(scaldpc_qary_min_sum_batch(q, &pmf[0][0], 1, 0, NULL, out));
    int zeros = 1;
    for (int v = 0; v < 6; v++) zeros &= out[v] == 0;
    printf("qary: all-zero decoding %s\n", zeros ? "yes" : "no");
    scaldpc_qary_destroy(q);
    /* Decoder::into_llr alone (decoder.rs:744-768): ln(0.14 / 0.02) = 1.9459101 in f32, ln(max / 0) = +inf */
    const float row[15] = {0, 0, 0, 0, .14f, .14f, .14f, .14f, .14f, .14f, .14f, .02f, 0, 0, 0};
    float llr[15];
    CHECK(scaldpc_qary_into_llr(row, 1, 15, 0, NULL, llr));
    printf("into_llr: %.7f %s\n", (double)llr[11], llr[0] > 3e38f ? "inf" : "finite");
    if (llr[11] != 1.9459101f || !(llr[0] > 3e38f) || llr[4] != 0.0f) return 4;
    return (ok == BATCH && rc == SCALDPC_EINVAL && zeros) ? 0 : 2;
}

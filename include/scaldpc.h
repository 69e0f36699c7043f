/*
 * scaldpc.h -- C ABI of libscaldpc: MI355X (gfx950) LDPC belief-propagation
 * decoders behind the decoder boundary of atneit/SCA-LDPC's Monte-Carlo drivers.
 *
 * Plain C types only (pointers, sizes, ints); opaque handles; every entry point
 * returns an int status (0 = ok) and never throws across the boundary;
 * scaldpc_last_error() gives the message of the calling thread's last failure.
 *
 * What each entry point replaces in the reference (paths relative to
 * simulate-with-python/):
 *
 *   scaldpc_bp_create            ldpc.bp_decoder.__init__ dense->sparse graph build,
 *                                call sites simulate/decode.py:155-161, simulate/hqc.py:694-699
 *   scaldpc_bp_set_channel_probs the `error_rate=` / `channel_probs=` constructor kwargs (same sites)
 *   scaldpc_bp_append_rows /     the attack loop's growing H: add_check's np.vstack (simulate/hqc.py:885-908) and the
 *     _set_channel_probs_tail    decoder rebuild of every decode (hqc.py:680,694) -> rows appended to a live decoder
 *   scaldpc_bp_decode_batch      ldpc.bp_decoder.decode(v), simulate/decode.py:171, simulate/hqc.py:708
 *                                (batched: one call = `batch` independent decode() calls)
 *   scaldpc_bp_destroy           object lifetime
 *   scaldpc_mc_fer_run           the per-trial body of simulate_frame_error_rate,
 *                                simulate/decode.py:36-40,165-175 (noise, syndrome, decode, compare)
 *   scaldpc_mc_hqc_run           hqc.decode() input assembly + success test for synthetic trials,
 *                                simulate/hqc.py:684-705,742-749
 *   scaldpc_qary_create          simulate_rs Decoder::new via PyO3 `#[new]`,
 *                                simulate_rs/src/pydecoder.rs:24-45 -> simulate_rs/src/decoder.rs:494-553
 *   scaldpc_qary_min_sum_batch   PyO3 `min_sum`, simulate_rs/src/pydecoder.rs:53-65
 *                                -> decoder.rs:668-692 (into_llr) + decoder.rs:560-666 (min_sum)
 *   scaldpc_qary_into_llr        Decoder::into_llr, simulate_rs/src/decoder.rs:668-692 (== decoder_special.rs:619-643),
 *                                the conversion alone (its known-answer test: decoder.rs:744-768)
 *   scaldpc_qary_special_create / _min_sum_batch
 *                                simulate_rs/src/pydecoder.rs:96-117,125-145
 *                                -> simulate_rs/src/decoder_special.rs:387-464, 471-617
 *
 * Threading: calls on distinct handles are independent; calls on one handle are
 * serialised internally, so one decoder object may be shared by many host
 * threads as the reference's thread pool does (simulate/decode.py:247-262).
 *
 * Device / host buffers: unless SCALDPC_F_DEVICE_IO is set, `in`/`out_*` are host
 * pointers and are staged through device memory by the call.  With
 * SCALDPC_F_DEVICE_IO they are device pointers (e.g. torch tensors' data_ptr())
 * and nothing crosses PCIe.  `stream` is a hipStream_t passed as void* (NULL =
 * the handle's own non-blocking stream); the call returns after the stream work is
 * complete unless SCALDPC_F_ASYNC is set (device I/O only).  With device pointers the
 * caller owns the ordering: pass the stream the inputs were produced on (or
 * synchronise first), and do not touch the handle's priors or buffers while an
 * SCALDPC_F_ASYNC call is still in flight.
 */
#ifndef SCALDPC_H
#define SCALDPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCALDPC_VERSION 103

/* status codes */
#define SCALDPC_OK 0
#define SCALDPC_EINVAL 1   /* bad argument (shape, mode, method, NULL) -> Python ValueError */
#define SCALDPC_EHIP 2     /* HIP runtime failure */
#define SCALDPC_ENOMEM 3
#define SCALDPC_EPMF 4     /* pmf row does not sum to 1 +- 1e-3 (decoder.rs:683-684 assert) */
#define SCALDPC_ENOCONF 5  /* a check admits no finite configuration (decoder.rs:618 assert) */
#define SCALDPC_EDEGREE 6  /* degree/alphabet outside what the kernels are built for */

/* bp methods (ldpc.bp_decoder `bp_method`) */
#define SCALDPC_BP_PRODUCT_SUM 0 /* "product_sum"/"ps"(+_log): tanh rule, LLR domain, fp32 */
#define SCALDPC_BP_MIN_SUM 1     /* "min_sum"/"ms"(+_log): scaling `alpha` (0 => 1 - 2^-iter) */

/* decode input convention (ldpc.bp_decoder.decode) */
#define SCALDPC_IN_SYNDROME 0 /* in: uint8 [batch][m]; result = error estimate e */
#define SCALDPC_IN_RECEIVED 1 /* in: uint8 [batch][n]; s = H v mod 2; result = e XOR v */

/* flags */
#define SCALDPC_F_EARLY_EXIT 1u /* per-codeword stop at H e == s (reference behaviour) */
#define SCALDPC_F_DEVICE_IO 2u  /* in/out pointers are device pointers */
#define SCALDPC_F_ASYNC 4u      /* with DEVICE_IO: enqueue only, caller synchronises `stream` */

const char *scaldpc_last_error(void);
int scaldpc_version(void);
int scaldpc_device_count(int *count);
int scaldpc_set_device(int device);
/* Device and pinned-host blocks (up to 64 MiB each, 512 MiB in total) released by destroyed
 * handles are parked for the next handle instead of going through hipFree: the reference builds
 * a new decoder per decode (simulate/hqc.py:694), and allocation would otherwise cost several
 * times the decode.  scaldpc_trim() returns the parked blocks to the driver;
 * SCALDPC_NO_CACHE=1 in the environment disables parking.  The host-side graph mirrors and tables of a handle
 * (a few multi-megabyte vectors) are recycled the same way (blocks >= 32 KiB, at most 256 MiB parked, also returned
 * by scaldpc_trim()): left to malloc they are unmapped and page-faulted in again per decoder, at a cost that depends
 * on the state of the host process's heap (0.45 ms or 2-8 ms per construction, measured). */
int scaldpc_trim(void);
/* Leak accounting (tests, profiles/microbench/leak_check.py): blocks the library's allocator has handed
 * out and that a live handle still owns.  out[6] = { device blocks, device bytes, pinned-host blocks,
 * pinned-host bytes, parked (idle) blocks, parked bytes }.  A create / decode / destroy cycle must leave
 * out[0..3] where it found them -- the reference builds a decoder per decode (simulate/hqc.py:694-708). */
int scaldpc_debug_live_blocks(int64_t *out);
/* Fault injection (tests): the `countdown`-th allocation the library makes from now on (device or pinned host,
 * any handle, any thread) fails with SCALDPC_ENOMEM; 0 disarms.  Every entry point must then return an error
 * code, leave the handle either intact or refusing further calls (never decoding on half-updated state), and
 * scaldpc_*_destroy must still release everything (scaldpc_debug_live_blocks back at its starting value).
 * Opt-in: the injector exists only in a process started with SCALDPC_DEBUG=1 (read once); otherwise arming it
 * returns SCALDPC_EINVAL and the allocator never consults the countdown. */
int scaldpc_debug_fail_alloc(int32_t countdown);
/* Measurement aid for bench.py (`roofline.cache_ceiling_GBps`): an in-place read-all / write-all stream over
 * `bytes` of scratch device memory -- every wave reads `rows_per_wave` consecutive 256-B rows and writes them back,
 * the access shape of an in-place BP pass -- `reps` launches between two HIP events on the NULL stream of the
 * current device.  *gbps = read + written bytes per second.  At the size of a cache-resident tile group (209 MB on
 * the HQC-128 graph) this is what the Infinity Cache sustains for this shape; far beyond 256 MiB, what HBM does.
 * No reference counterpart (measurement only). */
int scaldpc_measure_rmw_stream(int64_t bytes, int32_t rows_per_wave, int32_t reps, double *gbps);

/* ------------------------------------------------------------------ binary BP */
typedef struct scaldpc_bp scaldpc_bp;

/* Graph in CSR: row_ptr[m+1], col_idx[nnz] ascending inside each row (host pointers). */
int scaldpc_bp_create(int32_t m, int32_t n, int64_t nnz, const int32_t *row_ptr,
                      const int32_t *col_idx, scaldpc_bp **out);
/* Per-bit prior error probabilities, float64 [n] (host). p = 0 / p = 1 are legal
 * (LLR = +-inf), as the reference's certainty-1.0 checks produce (hqc.py:689). */
int scaldpc_bp_set_channel_probs(scaldpc_bp *h, const double *probs);
/*
 * A graph that GROWS: the attack loop adds one parity check per oracle answer (hqc.py:885-908,
 * `H = np.vstack([H, row])`) and decodes every DECODE_EVERY answers on [H | I] (hqc.py:972-980, 680),
 * where the reference rebuilds the decoder from the dense matrix each time.  Here the decoder lives on:
 *   scaldpc_bp_append_rows(h, nrows, row_ptr, col_idx, new_n)
 *       appends `nrows` checks (CSR of the new rows only: row_ptr[0] = 0, columns strictly ascending,
 *       < new_n) and grows the block length to new_n >= n (the new columns come last -- for [Hin | I]
 *       each appended row brings its identity column).  Only the new rows are validated; device CSR
 *       and priors have spare capacity and are appended to; the tables of the row-parallel kernels
 *       (the single decode() of the attack loop) are updated in place -- one word per new edge while
 *       the edge's column has a free lane, a moved segment otherwise; what only the 64-codeword-tile
 *       and LDS kernels need (CSC, degree buckets, their tables) is rebuilt when one of them is next
 *       used.  Results are those of a decoder freshly built on the grown graph, bit for bit.
 *   scaldpc_bp_set_channel_probs_tail(h, first, count, probs)
 *       priors of columns [first, first + count) (float64, as scaldpc_bp_set_channel_probs); the
 *       columns an append added must be given theirs before the next decode.
 */
int scaldpc_bp_append_rows(scaldpc_bp *h, int32_t nrows, const int32_t *row_ptr, const int32_t *col_idx, int32_t new_n);
int scaldpc_bp_set_channel_probs_tail(scaldpc_bp *h, int32_t first, int32_t count, const double *probs);
/*
 * Decode `batch` independent inputs, flooding schedule, fp32 messages.
 *   max_iter <= 0 -> n (ldpc convention)
 *   out_bits  uint8 [batch][n]   required
 *   out_llr   float [batch][n]   optional (NULL): posterior log(p0/p1) of the error estimate
 *   out_iters int32 [batch]      optional: iteration at which H e == s first held (else max_iter)
 *   out_conv  uint8 [batch]      optional: 1 iff the returned decision satisfies H e == s
 */
int scaldpc_bp_decode_batch(scaldpc_bp *h, const uint8_t *in, int32_t input_kind, int32_t batch,
                            int32_t max_iter, int32_t method, float alpha, uint32_t flags,
                            void *stream, uint8_t *out_bits, float *out_llr, int32_t *out_iters,
                            uint8_t *out_conv);
/*
 * Measurement aid for bench.py: on the message state left by the last (fixed-iteration)
 * decode, run `iters` back-to-back launches of the check kernel and of the variable kernel of
 * `method`, each series bracketed by HIP events on its launch stream.
 *   ms[0] / ms[1]   total ms of the check / variable series          (float[2])
 *   launches[0..1]  launches in each series                           (int32[6])
 *   launches[2]     codewords swept per check launch
 *   launches[3]     lanes: 1 = the series ran one after the other over the whole tile group;
 *                   2 = the decode's two-stream schedule: the check series over the first
 *                   lane's tiles and the variable series over the second lane's tiles ran
 *                   concurrently, as they do in a decode's steady state.  There every launch is followed by an
 *                   event (its duration = the distance to the previous event on its lane); because those events
 *                   cost the schedule its back-to-back dispatch, the same launch pattern is run once more WITHOUT
 *                   them, bracketed per lane, and ms[0] / ms[1] are scaled so that check + variable = the true pair
 *   launches[4]     codewords swept per variable launch
 *   launches[5]     bit 0: min-sum ran in its record form (k_check_minsum_rec / k_var_rec; knob "minsum_rec");
 *                   bit 1: the variable pass was timed WITH its decision output -- the form every pass of an
 *                   early-exit decode launches; the timing follows the last decode (fixed iterations: no output);
 *                   bit 2: the timed variable launch left out the columns of degree <= 1 (record form, no output)
 */
int scaldpc_bp_time_kernels(scaldpc_bp *h, int32_t iters, int32_t method, float alpha, void *stream,
                            float *ms, int32_t *launches);
/* Tuning knob: codeword tiles (64 codewords each) per cache-resident group; 0 = auto
 * (largest group whose in-place message array stays within ~200 MB of Infinity Cache). */
int scaldpc_bp_set_tile_group(scaldpc_bp *h, int32_t tiles);
/* Number of codewords the last early-exit call re-decoded in its compact second pass
 * (stragglers of mostly-converged tile groups; results are identical either way;
 * environment SCALDPC_COMPACT_AFTER=0 disables the pass, =k moves the decision point). */
int scaldpc_bp_last_compacted(scaldpc_bp *h, int64_t *count);
/* Path statistics of the last call, out[4]:
 *   out[0] = scaldpc_bp_last_compacted
 *   out[1] = codewords decoded with the row-parallel kernels (wave = one row of one codeword,
 *            lane = edge; taken for calls -- or compact passes -- of at most 6 (min-sum) / 4
 *            (tanh rule) codewords on graphs too large for LDS: the single decode() of
 *            hqc.py:708).  Results are identical to the 64-codeword-tile kernels;
 *            SCALDPC_PATH=stream disables the path, =edge extends it to 64 codewords
 *   out[2] = deepest compaction level reached (0 = none; stragglers of a compact pass are
 *            compacted again, up to 3 levels)
 *   out[3] = reserved (0) */
int scaldpc_bp_last_stats(scaldpc_bp *h, int64_t *out);
/* Tuning / test knobs of one handle.  A new handle takes its defaults from the environment ONCE, at
 * creation (SCALDPC_PATH, SCALDPC_SPLIT, SCALDPC_GROUP_MB, SCALDPC_EL_MAX, SCALDPC_EL_FUSE,
 * SCALDPC_COMPACT_AFTER, SCALDPC_VAR_ORDER, SCALDPC_FIRST_FUSED, SCALDPC_FUSE_TEST, SCALDPC_MINSUM_REC,
 * SCALDPC_REC_SKIP1); the decode entry points never read the environment.  Every knob selects a path, a size or a
 * form that some graph or call still falls back to; the switches of variants that were measured and rejected
 * (round 3's test_overlap, var_form, rec_maskpos, rec_xmap, rec_sc1, fuse_finalize, speculate, minsum_loop) are gone
 * with their code (round 4; numbers in profiles/HISTORY.md) and are refused as unknown keys.  key / value (text):
 *   "path"          "auto" | "stream" (64-codeword tiles) | "edge" (row-parallel up to 64) | "lds"
 *   "split"         stream lanes per tile group (default 2)
 *   "group_mb"      budget of a cache-resident tile group in MB (default 215; large = stream from HBM)
 *   "el_max"        largest call the row-parallel kernels take (default 6 min-sum / 4 tanh)
 *   "el_fuse"       1 = two-launch early-exit loop of the row-parallel path (default), 0 = four-launch
 *   "compact_after" iteration from which stragglers may be handed to a compact pass (default 4, 0 = never)
 *   "var_order"     launch order of the columns in a variable-node pass: bit 0 = inside a degree by first
 *                   edge id (else by column id), bit 1 = heaviest columns first; -1 (default) = auto:
 *                   2 when a tile group runs as one stream lane, 1 otherwise
 *   "fuse_test"     1 (default) = in the early-exit tile loop the convergence test of an iteration rides on the check
 *                   pass of the next one (except where the host polls or stops), with sharded accumulators;
 *                   0 = a stand-alone launch after every variable pass (what poll iterations use anyway).
 *   "minsum_rec"    1 (default) = min-sum on the 64-codeword-tile kernels in its RECORD form: the check pass writes, per
 *                   row and codeword, the two magnitudes a min-sum check sends (8 B) and, per edge and tile, two lane
 *                   masks (sign, arg-min: 0.25 B per codeword) instead of 4 B per edge and codeword; the variable pass
 *                   rebuilds every message from them, bit for bit.  0 = messages both ways (what graphs with a row
 *                   wider than 64 or a column wider than 32 use anyway).
 *   "rec_skip1"     1 (default) = in the record form a variable pass WITHOUT output (fixed-iteration runs, every pass but
 *                   the first and the last) leaves out the columns of degree <= 1: such a column always sends its prior,
 *                   iteration 1 has written it into the message array and the record check pass never overwrites it
 *                   (the identity block of an HQC graph: 4000 of 21669 column waves per tile).  0 = all columns (what
 *                   passes with output launch anyway).
 *   "first_fused"   1 (default) = iteration 1 of the tile kernels runs without its check pass: the first variable
 *                   pass takes the first check-to-variable messages from a per-edge table (the message of a
 *                   zero-syndrome codeword) and the row's syndrome bit; 0 = check pass (reading the priors) + plain
 *                   variable pass, both in the message form (what graphs with a row or column wider than 64 use anyway).
 *   Results never depend on any of these. */
int scaldpc_bp_configure(scaldpc_bp *h, const char *key, const char *value);
/* Where a handle lives, out[4]: the device it was created on; the device (hipPointerGetAttributes)
 * of its graph allocation, of its message workspace and of its state planes (-1 = not allocated yet).
 * Every entry point switches to the handle's device for the call and restores the caller's. */
int scaldpc_bp_device_of(scaldpc_bp *h, int32_t *out);
void scaldpc_bp_destroy(scaldpc_bp *h);

/* ------------------------------------------- Monte-Carlo helpers on the device (K6) */
/*
 * The per-trial helpers of the reference's drivers -- draw the noise, form the syndrome,
 * decode, compare -- without a host round trip per trial.  Random numbers: Philox4x32-10,
 * key = seed, counter = (block, stream, GLOBAL trial index): a trial's inputs depend only
 * on (seed, first_trial + i), not on batch size or GPU count.
 *
 * scaldpc_mc_fer_run  = the body of simulate_frame_error_rate (simulate/decode.py:165-175):
 *   error_i ~ Bernoulli(channel_probs[i]) (flip iff stream-0 word i < floor(p_i 2^32)),
 *   syndrome = H error, decode, success = (decoding == error everywhere).
 *   out_error (optional) uint8 [batch][n]: the sampled error vectors.
 * scaldpc_mc_hqc_run  = synthetic hqc.decode() trials (simulate/hqc.py:684-705,742-749) on
 *   H = [Hin | I_R]: y = omega distinct positions in [0, N) (stream-1 candidates
 *   mulhi(word, N), duplicates skipped), checks = Hin y with each bit flipped with
 *   probability eps (stream 0), msg = [0]*N ++ checks, decode in received-vector mode,
 *   success = (decoded[:N] == indicator(y)).
 *   out_msg (optional) uint8 [batch][n]; out_y (optional) int32 [batch][omega], in draw order.
 * out_success uint8 [batch] required; out_iters int32 [batch] optional.
 * flags: SCALDPC_F_EARLY_EXIT, SCALDPC_F_DEVICE_IO (outputs are device pointers).
 */
int scaldpc_mc_fer_run(scaldpc_bp *h, int64_t first_trial, int32_t batch, uint64_t seed, int32_t max_iter,
                       int32_t method, float alpha, uint32_t flags, void *stream, uint8_t *out_success,
                       int32_t *out_iters, uint8_t *out_error);
int scaldpc_mc_hqc_run(scaldpc_bp *h, int32_t omega, double eps, int64_t first_trial, int32_t batch, uint64_t seed,
                       int32_t max_iter, int32_t method, float alpha, uint32_t flags, void *stream,
                       uint8_t *out_success, int32_t *out_iters, uint8_t *out_msg, int32_t *out_y);

/* ------------------------------------------------------------ q-ary min-sum */
typedef struct scaldpc_qary scaldpc_qary;

/* H: int8 [R][N] dense row-major with entries in {-1,0,+1} (what pydecoder.rs:24
 * receives), Q = 2B+1 symbols, `iterations` fixed (no early exit, decoder.rs:660). */
int scaldpc_qary_create(int32_t R, int32_t N, int32_t B, const int8_t *H, int32_t iterations,
                        scaldpc_qary **out);
/* pmf: float [batch][N][Q] probabilities (LLR conversion inside, as pydecoder.rs:60);
 * out: int8 [batch][N] hard decisions in [-B, B]. */
int scaldpc_qary_min_sum_batch(scaldpc_qary *h, const float *pmf, int32_t batch, uint32_t flags,
                               void *stream, int8_t *out);
void scaldpc_qary_destroy(scaldpc_qary *h);
/* The probability -> LLR conversion both decoders apply to their inputs, on its own:
 * llr[r][q] = ln(max_q' pmf[r][q'] / pmf[r][q]) in f32 (+inf where pmf = 0); pmf, llr: float [rows][Q].
 * Runs on the device with glibc's logf algorithm and the correctly rounded f32 division, so the values
 * are bit for bit those of the host's logf (what the reference's f32::ln calls).  A row that does not
 * sum to 1 +- 1e-3 returns SCALDPC_EPMF (the reference asserts).  flags: SCALDPC_F_DEVICE_IO. */
int scaldpc_qary_into_llr(const float *pmf, int64_t rows, int32_t Q, uint32_t flags, void *stream, float *llr);
/* Test / tuning knobs of one q-ary handle (defaults from SCALDPC_QARY_WAVE / SCALDPC_QARY_NO_UNROLL, read
 * once at creation): "wave" = -1 auto | 0 codeword per lane | 1 wave per (check, codeword);
 * "unroll" = 1 register-resident unrolled enumeration for small alphabets | 0 off;
 * "tree" = 1 tree-walk check kernel for the Kyber shape (B = 2, six coefficient edges per check) | 0 off
 * (SCALDPC_QARY_NO_TREE); "dp" = 1 (default) the same shape's check update as a min-plus recursion over the edges in the
 * reference's order of additions -- no enumeration, the reference's messages bit for bit (scaldpc_qary_special.h) --
 * for calls of at least "dp_min" codewords (default 5; below, the tree walk), the row's edges split over four waves up
 * to "dp_split" codewords (default 64) and over two up to "dp_split2" (default 192) | 0 off; "llr_tiled", "var_small": forms of the conversion / variable kernels;
 * "timing" = 1: bracket the launches of a call with HIP events (scaldpc_qary_last_timing). */
int scaldpc_qary_configure(scaldpc_qary *h, const char *key, const char *value);
/* Measurement aid for bench.py (the q-ary counterpart of scaldpc_bp_time_kernels): after
 * scaldpc_qary_configure(h, "timing", "1"), every check-node and variable-node launch of a call is bracketed by
 * HIP events on the launch stream (off by default -- the product path records nothing).  For the last call:
 *   ms[0] / ms[1]  total ms of its check / variable launches;  ms[2]  from the first check launch to the last
 *                  variable launch (the iteration loop, without the probability -> LLR conversion and the copies)
 *   info[0] iterations run;  info[1] check kernel: 0 k_q_check_unrolled<3,7>, 1 k_q_check_unrolled<5,5>,
 *           2 k_q_special_check_tree<5,6> (+ wave kernel for other row degrees), 3 k_q_special_check_wave,
 *           4 k_q_check_wave, 5 k_q_special_check, 6 k_q_check, 7 k_q_special_check_dp<5,6> (either form), 8 k_q_check_dp<3,7>;  info[2] batch;  info[3] largest check degree */
int scaldpc_qary_last_timing(scaldpc_qary *h, float *ms, int32_t *info);

/* DecoderSpecial: H = [H' | I_R]; first N-R variables over [-B,B], last R over [-BSUM,BSUM]. */
int scaldpc_qary_special_create(int32_t R, int32_t N, int32_t B, int32_t BSUM, const int8_t *H,
                                int32_t iterations, scaldpc_qary **out);
/* pmf_b: float [batch][N-R][2B+1]; pmf_sum: float [batch][R][2BSUM+1]; out int8 [batch][N]. */
int scaldpc_qary_special_min_sum_batch(scaldpc_qary *h, const float *pmf_b, const float *pmf_sum,
                                       int32_t batch, uint32_t flags, void *stream, int8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* SCALDPC_H */

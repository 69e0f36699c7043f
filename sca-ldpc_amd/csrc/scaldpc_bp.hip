// libscaldpc -- binary LDPC belief propagation on MI355X (gfx950), flooding schedule.
//
// Replaces the decode loop behind `ldpc.bp_decoder(H, ...).decode(v)` as the
// reference calls it (simulate-with-python/simulate/decode.py:155-161,171;
// simulate/hqc.py:694-699,708).  The arithmetic follows that package's sweep
// order: per check an exclusive forward/backward sweep
// over the row in ascending column, per variable an exclusive prefix/suffix
// SUM over the column in ascending row, hard decision `L <= 0 -> 1`, then
// H e == s.  Nothing is ever "total minus self", so +-inf priors (certainty-1.0
// checks, hqc.py:689) never meet an inf - inf.
//
// Memory design (measured on MI355X, profiles/microbench/rmw_stream.hip):
//   * ONE fp32 message array, updated IN PLACE: a check node reads all its v2c
//     and then overwrites them with c2v at the same addresses; a variable node
//     does the converse.  Every node owns its edges exclusively, so no other
//     thread ever sees a half-updated edge.
//         msg : float [tile][edge][64]      edge id = CSR position, tile = 64 codewords
//     A check node's edges are one contiguous range: a wave working on
//     (row, tile) streams deg x 256 B of consecutive memory; a variable node
//     gathers its 256 B edge rows through the CSC permutation.
//   * Codeword tiles are decoded in GROUPS whose message array fits the 256 MiB
//     Infinity Cache (~<= 200 MB): all max_iter iterations of a group run
//     back-to-back, so after the first touch the messages never go to HBM again
//     (an in-place read-all/write-all stream runs at ~6.5-7.3 TB/s from the
//     Infinity Cache against ~4.5 TB/s from HBM).
//   * Bit planes (syndrome, received word, hard decision, done mask):
//         u64 [tile][node]      bit c = codeword c of the tile
//     posterior : float [tile][var][64]
//
// No MFMA anywhere: this is a sparse gather/scatter stream, not a contraction.
#include "scaldpc_common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

typedef unsigned long long u64;

using namespace scaldpc;

#include "scaldpc_bp_kernels.h"

namespace {

struct HostBuckets {
    Buckets bk;
    pvec<int> list;
    bool has_generic = false;
};

}  // namespace

// ===========================================================================
// handle
// ===========================================================================
// Tuning / test knobs of a handle.  The environment is read ONCE, when the handle is created
// (SCALDPC_PATH, _SPLIT, _GROUP_MB, _EL_MAX, _EL_FUSE, _COMPACT_AFTER, ...); afterwards
// scaldpc_bp_configure() changes them -- the decode entry points never call getenv().
// Round 4 pruned the knobs that only switched a measured-and-rejected variant back on (test_overlap, var_form = 0,
// rec_maskpos = 0, rec_xmap = 0, rec_sc1 = 0, fuse_finalize = 0, speculate = 0, minsum_loop; numbers in
// profiles/HISTORY.md): what is left selects a path, a size, or a form that something still falls back to.
struct Knobs {
    enum Path { AUTO = 0, STREAM = 1, EDGE = 2, LDS = 3 };
    int path = AUTO;
    int split = 2;            // stream lanes per tile group
    double group_mb = 215.0;  // cache-resident group budget (auto_group)
    int el_max = -1;          // row-parallel limit; -1 = per-method default (6 / 4)
    int el_fuse = 1;          // early-exit row-parallel loop: fused two-launch form
    int compact_after = -1;   // -1 = default (4); 0 = no compact pass
    int var_order = -1;       // k_var launch order: bit 0 = inside a degree by first edge id, bit 1 = heaviest columns first; -1 = auto
    int fuse_test = 1;        // early-exit tile loop: the convergence test of iteration it rides on the check pass of it + 1 wherever the host does not need the verdict in between
    int first_fused = 1;      // iteration 1 of the tile kernels without its check pass (k_var_first from the first-message table); 0 = check pass + plain variable pass
    int rec_skip1 = 1;        // record form: passes without output leave out the columns of degree <= 1 (their message is the prior, written once by iteration 1; the record check pass never overwrites it); 0 = all columns every pass (A/B knob)
    int minsum_rec = 1;       // min-sum on the tile kernels: check pass writes per-row records + lane masks instead of messages (k_check_minsum_rec / k_var_rec); 0 = message form
};

struct scaldpc_bp {
    Knobs kn;
    int m = 0, n = 0;
    long E = 0;
    int max_row_deg = 0, max_col_deg = 0, min_row_deg = 0;
    // device graph
    int *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr;
    int *d_var_list = nullptr, *d_row_list = nullptr;
    int *d_var_meta = nullptr, *d_csc_list = nullptr;  // k_var: packed column descriptors + edge lists in launch order
    Buckets var_bk{}, row_bk{};
    bool need_scratch = false;
    // priors
    float *d_prior = nullptr;
    bool have_prior = false;
    int prior_n = 0;  // columns whose prior is set (a grown graph needs scaldpc_bp_set_channel_probs_tail for the new ones)
    // workspace (state arrays sized for cap_tiles, messages for cap_group tiles)
    int cap_tiles = 0, cap_group = 0, tile_group = 0;
    float *d_msg = nullptr, *d_scratch = nullptr, *d_post = nullptr;
    u64 *d_synd = nullptr, *d_recv = nullptr, *d_hard = nullptr, *d_done = nullptr, *d_conv = nullptr,
        *d_unsat = nullptr;
    int *d_iters = nullptr, *d_remaining = nullptr;
    int cap_remaining = 0;
    // d_remaining holds rem_rows rows of cap_remaining "codewords still running after iteration it" counters: every tile
    // group of a call takes the next row (zeroed once per call, not once per group: a memset is a 5 us launch in the
    // group's dependency chain); rem_slot = next free row, rem_rows = all used (the next taker zeroes the array).
    // rem_rows follows the groups a call can run (at most REM_SLOTS): max_iter defaults to n, and 512 rows of n + 2
    // counters were 44 MB -- allocated and cleared -- on an HQC-sized decoder that decodes ONE codeword.
    int rem_slot = 0, rem_rows = 0;
    bool post_alloc = false;
    // host-I/O staging
    uint8_t *d_in = nullptr, *d_out_bits = nullptr, *d_out_conv = nullptr;
    float *d_out_llr = nullptr;
    int *d_out_iters = nullptr;
    size_t cap_in = 0, cap_out_bits = 0, cap_out_llr = 0, cap_out_b = 0;
    int *h_remaining = nullptr;  // pinned
    // small host calls (one tile, a handful of codewords): fused reshaping kernels, one pinned staging buffer
    uint8_t *h_io = nullptr, *d_out_all = nullptr;
    size_t cap_h_io = 0, cap_out_all = 0;
    bool small_prepared = false;  // k_small_prepare already reset the state this call (run_core / the early-exit loop skip theirs)
    // Monte-Carlo helpers
    pvec<double> h_probs;
    u64 *d_thr = nullptr, *d_mc = nullptr, *d_diff = nullptr;
    int *d_ylist = nullptr;
    uint8_t *d_succ = nullptr;
    size_t cap_mc = 0, cap_ylist = 0, cap_succ = 0, cap_diff = 0;
    bool thr_valid = false;
    // compact second pass over stragglers (early-exit runs)
    // (level k re-decodes the stragglers of level k-1 in dense tiles of their own; level 0 = the call's arrays)
    struct Level {
        int cap_tiles = 0;
        u64 *synd = nullptr, *hard = nullptr, *done = nullptr, *conv = nullptr, *unsat = nullptr;
        int *iters = nullptr, *ids = nullptr, *slot_of = nullptr;
        float *post = nullptr;
        size_t cap_post = 0, cap_slot_of = 0;
    };
    static constexpr int MAX_LEVELS = 3;
    Level lv[MAX_LEVELS + 1];  // [0] unused
    long stat_levels = 0;      // deepest compact level the last call reached
    long stat_deferred = 0;  // codewords re-decoded by the compact pass in the last call
    // row-parallel path (a handful of codewords): per-codeword message / prefix arrays [codeword][edge]
    float *d_emsg = nullptr;
    size_t cap_el = 0;
    int *d_el_unsat = nullptr;  // [iteration][64] "some row unsatisfied" flags of the fused early-exit loop
    size_t cap_el_unsat = 0;
    // k_el_var: columns packed into waves ("bins") of 64 lane slots, each column a segment of `cap` lanes
    int *d_el_slots = nullptr, *d_el_slot_col = nullptr;  // views into d_el_tab: [2 * 64 * el_cap_bins] and [64 * el_cap_bins] ints
    int el_waves = 0;     // bins in use
    int el_cap_bins = 0;  // bins d_el_tab has room for
    bool el_ok = false;   // the row-parallel path can take this graph (non-empty, no row / column wider than a wave)
    pvec<int> h_el_slots, h_el_col;  // host mirrors of the two parts of d_el_tab
    pvec<int> seg_slot;            // per column: bin * 64 + first lane of its segment (-1: none)
    pvec<unsigned char> seg_cap;   // per column: lanes of its segment
    int el_open_used = 0;                 // lanes handed out in the last bin
    // A handle whose graph grows (scaldpc_bp_append_rows): CSR and priors move to allocations of their
    // own with spare capacity, the row-parallel tables keep a few free lanes per column and are
    // updated in place; everything only the tile / LDS kernels need (CSC, degree buckets, their
    // tables) is marked stale and rebuilt from the host mirror when one of those paths is next taken.
    bool incremental = false, full_stale = false;
    pvec<int> hg_col_idx;  // host CSR mirror (incremental handles)
    int *d_csr_rp = nullptr, *d_csr_ci = nullptr;
    float *d_prior_buf = nullptr;
    size_t cap_rows = 0, cap_edges = 0, cap_cols = 0;
    int ws_m = 0, ws_n = 0;  // what the workspace planes are sized for
    pvec<int> el_dirty_slot, el_dirty_col;  // words an append call changed (kept for their capacity)
    int *d_pairs = nullptr;  // staging of table updates
    int *h_pairs = nullptr;  // pinned
    size_t cap_pairs = 0;
    int *d_graph = nullptr;  // ONE allocation behind the arrays every path needs (CSR, CSC, var list, d_prior: views into it)
    // The tables only one kernel family reads are built on that family's first use -- a decoder
    // that lives for one single decode (hqc.py:694) never pays for the tile kernels' tables, a
    // benchmark never for the row-parallel ones.  What the builders need stays on the host:
    int *d_tile_tab = nullptr;  // row descriptors, k_var records, re-laid edge list (d_row_list, d_var_meta, d_csc_list)
    // iteration 1 without its check pass: {first check-to-variable message of a zero-syndrome codeword, row} per position
    // of the re-laid edge list; valid for one (method, alpha of iteration 1) and the current priors / graph
    int2 *d_first_tab = nullptr;
    size_t cap_first = 0;
    // record form of min-sum on the tile kernels: per (tile, row) the two magnitudes, per (tile, edge) two lane masks;
    // d_csc_row = row of every position of the re-laid edge list (inside d_tile_tab)
    float *d_rec = nullptr;
    ulonglong2 *d_mask = nullptr;
    int cap_rec_group = 0;
    int *d_csc_row = nullptr, *d_var_rows = nullptr, *d_csr_pos = nullptr;
    bool var_reversed = false;  // the column records are laid out heaviest first (var_order bit 1): the degree-1 bucket is at the END
    bool first_valid = false;
    int first_method = -1;
    float first_alpha = 0.0f;
    int *d_el_tab = nullptr;    // k_el_var slots and wave info (d_el_slots, d_el_winfo)
    pvec<int> hg_row_ptr, hg_cdeg, hg_col_ptr, hg_csc_edge;
    HostBuckets hg_var, hg_row;
    int stat_el = 0;  // codewords the row-parallel kernels decoded in the last call
    int identity_from = -1;  // n - m if the last m columns of H are I_m (H = [Hin | I]), else -1
    hipStream_t own_stream = nullptr;
    int device = 0;  // the device the handle (and its stream) was created on
    hipStream_t aux_stream[4] = {};  // further lanes of a tile group (iterate_tiles)
    hipEvent_t ev_join[4] = {}, ev_phase[4] = {};
    // Set by the first SCALDPC_F_ASYNC call and never cleared: work may be in flight when a later call
    // (or destroy) releases a buffer, so this handle's blocks go back through hipFree (which
    // waits for the device) instead of being parked for immediate reuse.
    bool async_used = false;
    // Set when scaldpc_bp_append_rows failed part-way (an allocation, a copy): host mirrors and device arrays may
    // disagree, so every later call on the handle returns an error instead of decoding on it; destroy still frees all.
    bool broken = false;
    int last_group = 0;  // tiles of the last decoded group (for scaldpc_bp_time_kernels)
    bool last_early = false;  // ... and whether that decode ran with early exit (its variable passes then all write decisions)
    std::mutex mu;
};

namespace {

// one knob from its textual value; false = unknown key or bad value
bool set_knob(Knobs &k, const char *key, const char *val)
{
    if (!key || !val) return false;
    if (!strcmp(key, "path")) {
        if (!strcmp(val, "auto") || !*val) k.path = Knobs::AUTO;
        else if (!strcmp(val, "stream")) k.path = Knobs::STREAM;
        else if (!strcmp(val, "edge")) k.path = Knobs::EDGE;
        else if (!strcmp(val, "lds")) k.path = Knobs::LDS;
        else return false;
        return true;
    }
    char *end = nullptr;
    const double x = strtod(val, &end);
    if (end == val) return false;
    if (!strcmp(key, "split")) k.split = (int)x;
    else if (!strcmp(key, "group_mb")) k.group_mb = x;
    else if (!strcmp(key, "el_max")) k.el_max = (int)x;
    else if (!strcmp(key, "el_fuse")) k.el_fuse = (int)x != 0;
    else if (!strcmp(key, "compact_after")) k.compact_after = (int)x;
    else if (!strcmp(key, "var_order")) k.var_order = (int)x;
    else if (!strcmp(key, "first_fused")) k.first_fused = (int)x != 0;
    else if (!strcmp(key, "minsum_rec")) k.minsum_rec = (int)x != 0;
    else if (!strcmp(key, "rec_skip1")) k.rec_skip1 = (int)x != 0;
    else if (!strcmp(key, "fuse_test")) k.fuse_test = (int)x != 0;
    else return false;
    return true;
}

void knobs_from_env(Knobs &k)
{
    static const char *const names[][2] = {{"SCALDPC_PATH", "path"}, {"SCALDPC_SPLIT", "split"},
                                           {"SCALDPC_GROUP_MB", "group_mb"}, {"SCALDPC_EL_MAX", "el_max"},
                                           {"SCALDPC_EL_FUSE", "el_fuse"}, {"SCALDPC_COMPACT_AFTER", "compact_after"},
                                           {"SCALDPC_VAR_ORDER", "var_order"}, {"SCALDPC_FIRST_FUSED", "first_fused"},
                                           {"SCALDPC_FUSE_TEST", "fuse_test"}, {"SCALDPC_MINSUM_REC", "minsum_rec"},
                                           {"SCALDPC_REC_SKIP1", "rec_skip1"}};
    for (auto &nm : names)
        if (const char *e = getenv(nm[0])) (void)set_knob(k, nm[1], e);
}

// bucket b holds nodes with bounds[b-1] < deg <= bounds[b]; beyond the last bound -> generic (maxd 0)
void build_buckets(const pvec<int> &deg, const int *bounds, int nb, bool keep_isolated, HostBuckets &out)
{
    // Stable counting sort by degree (ascending node id inside a degree): bucket lists are
    // degree ranges of it, so neighbouring waves of a launch run the same exact-degree code
    // path.  (Decoder construction is on the attack loop's critical path: no per-bucket
    // vectors, no comparison sort.)
    const int n = (int)deg.size();
    int maxdeg = 0;
    for (int d : deg) maxdeg = std::max(maxdeg, d);
    pvec<int> start(maxdeg + 2, 0);
    for (int d : deg) start[d + 1]++;
    for (int d = 0; d <= maxdeg; d++) start[d + 1] += start[d];
    pvec<int> order(n), cursor(start.begin(), start.end() - 1);
    for (int i = 0; i < n; i++) order[cursor[deg[i]]++] = i;
    out.list.clear();
    out.list.reserve(n);
    out.has_generic = false;
    Buckets &bk = out.bk;
    memset(&bk, 0, sizeof bk);
    int lo = keep_isolated ? 0 : 1;  // smallest degree of the bucket being formed
    for (int b = 0; b <= nb && lo <= maxdeg; b++) {
        const int hi = b < nb ? std::min(bounds[b], maxdeg) : maxdeg;  // beyond the last bound: any-degree fallback
        if (hi < lo) continue;
        const int first = start[lo], last = start[hi + 1];
        if (last > first) {
            const int i = bk.nb++;
            bk.maxd[i] = b < nb ? bounds[b] : 0;
            if (b == nb) out.has_generic = true;
            bk.off[i] = (int)out.list.size();
            bk.cnt[i] = last - first;
            bk.blk[i + 1] = bk.blk[i] + (bk.cnt[i] + 3) / 4;
            out.list.insert(out.list.end(), order.begin() + first, order.begin() + last);
        }
        lo = hi + 1;
    }
}

template <typename T>
int grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return 0;
    dev_free(*p);
    *cap = 0;
    const size_t want = need + need / 8;  // a little headroom: the buffers of a decoder whose graph grows would otherwise be re-made at every step
    SC_TRY(dev_alloc(p, want));
    *cap = want;
    return 0;
}

// tiles per group: the in-place message array of a group should stay resident in the
// 256 MiB Infinity Cache (measured knee between 209 and 261 MB, profiles/microbench)
int auto_group(const scaldpc_bp *h, int T)
{
    const double budget = h->kn.group_mb * 1e6;  // default 215: 4 tiles of the HQC-128 bench graph = 208.9 MB, fastest measured; 261 MB falls off
    const double per_tile = (double)h->E * TW * sizeof(float);
    const int g = per_tile > 0 ? (int)(budget / per_tile) : T;
    return std::max(1, std::min(g, T));
}

// waves per tile of a k_parity launch = words per tile of the unsat arrays
int parity_waves(const scaldpc_bp *h) { return (h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE) * 4; }

constexpr int REM_SLOTS = 512;

int ensure_workspace(scaldpc_bp *h, int T, int G, bool want_post, int max_iter)
{
    if (T > h->cap_tiles || h->m > h->ws_m || h->n > h->ws_n) {
        dev_free(h->d_synd); dev_free(h->d_recv); dev_free(h->d_hard); dev_free(h->d_done);
        dev_free(h->d_conv); dev_free(h->d_unsat); dev_free(h->d_iters); dev_free(h->d_post);
        T = std::max(T, h->cap_tiles);
        h->cap_tiles = 0;
        h->post_alloc = false;
        // a growing graph gets planes with room for the rows / columns still to come
        const int wm = h->incremental ? h->m + h->m / 2 + 64 : h->m, wn = h->incremental ? h->n + h->n / 2 + 64 : h->n;
        const size_t pw = (size_t)(wm + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE) * 4;
        SC_TRY(dev_alloc(&h->d_synd, (size_t)T * wm));
        SC_TRY(dev_alloc(&h->d_recv, (size_t)T * wn));
        SC_TRY(dev_alloc(&h->d_hard, (size_t)T * wn));
        SC_TRY(dev_alloc(&h->d_done, (size_t)T));
        SC_TRY(dev_alloc(&h->d_conv, (size_t)T));
        SC_TRY(dev_alloc(&h->d_unsat, (size_t)T * pw));
        SC_TRY(dev_alloc(&h->d_iters, (size_t)T * TW));
        h->cap_tiles = T;
        h->ws_m = wm;
        h->ws_n = wn;
    }
    if (want_post && !h->post_alloc) {
        SC_TRY(dev_alloc(&h->d_post, (size_t)h->cap_tiles * h->ws_n * TW));
        h->post_alloc = true;
    }
    // (the message arrays are allocated by the path that uses them: ensure_msg / ensure_el)
    // rows of counters: one per tile group of the call, the compact levels' groups (fewer tiles each) included, + row 0
    const int groups = (T + std::max(1, G) - 1) / std::max(1, G);
    const int want_rows = std::min(REM_SLOTS, 2 * groups + 8);
    if (max_iter + 2 > h->cap_remaining || want_rows > h->rem_rows) {
        dev_free(h->d_remaining);
        cached_free(h->h_remaining);
        h->h_remaining = nullptr;
        h->cap_remaining = 0;
        h->rem_rows = 0;
        const int len = std::max(max_iter + 2, h->cap_remaining), rows = std::max(want_rows, h->rem_rows);
        SC_TRY(dev_alloc(&h->d_remaining, (size_t)rows * (size_t)len));
        SC_TRY(cached_alloc((void **)&h->h_remaining, sizeof(int) * (size_t)len, true));
        h->cap_remaining = len;
        h->rem_rows = rows;
    }
    h->rem_slot = h->rem_rows;  // a new call: the first tile group that needs counters zeroes the array
    return 0;
}

// the next zeroed row of "still running" counters (the row-parallel and small-call paths keep using row 0 with a
// memset of their own; they run between tile groups on the same stream)
int next_remaining_row(scaldpc_bp *h, hipStream_t s, int **row)
{
    if (h->rem_slot >= h->rem_rows) {
        SC_HIP(hipMemsetAsync(h->d_remaining, 0, sizeof(int) * (size_t)h->rem_rows * h->cap_remaining, s));
        h->rem_slot = 1;  // (row 0 is the other paths')
    }
    *row = h->d_remaining + (size_t)h->rem_slot++ * h->cap_remaining;
    return 0;
}

// Host staging buffer for table uploads, reused by the thread's later constructions (fresh pages
// for a multi-megabyte vector cost more page-fault time than filling it).  Not cleared.
int *stage_buffer(size_t ints)
{
    struct Staging {
        int *p = nullptr;
        size_t cap = 0;
        ~Staging() { free(p); }
    };
    static thread_local Staging stage;
    if (ints > stage.cap || stage.cap > ((size_t)64 << 20)) {  // (and do not sit on more than 256 MB)
        free(stage.p);
        stage.cap = 0;
        stage.p = (int *)malloc(std::max<size_t>(ints, 1) * sizeof(int));
        if (!stage.p) return nullptr;
        stage.cap = ints;
    }
    return stage.p;
}

int upload_table(int **dst, const int *host, size_t ints)
{
    SC_TRY(dev_alloc(dst, ints));
    SC_HIP(hipMemcpy(*dst, host, ints * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// Tables of the 64-codeword-tile kernels: one descriptor per wave of a check launch, one record per
// wave of a k_var launch (descriptor + first edges inline), the edge lists in launch order.
int ensure_tile_tables(scaldpc_bp *h)
{
    if (h->d_tile_tab) return 0;
    const HostBuckets &hv = h->hg_var, &hr = h->hg_row;
    const int *row_ptr = h->hg_row_ptr.data(), *cdeg = h->hg_cdeg.data(), *col_ptr = h->hg_col_ptr.data(),
              *csc_edge = h->hg_csc_edge.data();
    size_t total = 0;
    auto reserve = [&](size_t cnt) {
        const size_t off = total;
        total += (cnt + 63) / 64 * 64;
        return off;
    };
    const size_t o_var_meta = reserve((size_t)4 * VAR_REC * hv.bk.blk[hv.bk.nb] + 4), o_csc_list = reserve((size_t)h->E + 1 + 64);
    const size_t o_row_list = reserve((size_t)16 * hr.bk.blk[hr.bk.nb] + 4);
    const size_t o_csc_row = reserve((size_t)h->E + 1 + 64);
    const size_t o_csr_pos = reserve((size_t)h->E + 64);  // position of every CSR edge in the re-laid list
    const size_t o_var_rows = reserve((size_t)4 * VAR_INLINE * hv.bk.blk[hv.bk.nb] + 4);  // rows of the records' inline edges
    int *host = stage_buffer(total);
    if (!host) return fail(SCALDPC_ENOMEM, "out of host memory");
    for (int b = 0; b < hr.bk.nb; b++) {
        const int blocks = hr.bk.blk[b + 1] - hr.bk.blk[b];
        for (int sl = 0; sl < blocks * 4; sl++) {
            int *md = host + o_row_list + 4 * ((size_t)hr.bk.blk[b] * 4 + sl);
            const bool pad = sl >= hr.bk.cnt[b];
            const int r = pad ? -1 : hr.list[hr.bk.off[b] + sl];
            md[0] = r;
            md[1] = pad ? 0 : row_ptr[r];
            md[2] = pad ? 0 : row_ptr[r + 1] - row_ptr[r];
            md[3] = hr.bk.maxd[b];
        }
    }
    int *meta = host + o_var_meta, *relaid = host + o_csc_list, *relaid_row = host + o_csc_row;
    int pos = 0;
    pvec<int> edge_row((size_t)h->E);
    for (int r = 0; r < h->m; r++)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) edge_row[e] = r;
    // launch order of the columns: the bucket lists are sorted by degree, ascending column id inside a
    // degree.  var_order = 1 re-sorts each run of equal degree by the column's FIRST edge id, so that
    // neighbouring waves of a launch start their gathers in neighbouring rows of the message array
    // (the records carry the column id, so the order is free; results cannot depend on it).
    pvec<int> order_buf;
    const int *vlist = hv.list.data();
    // auto: a tile group that runs as ONE stream lane (a tile too large to share the cache with a second one:
    // the HQC-256 graph) has nothing to fill the tail of its launches with, so its heaviest columns go first
    // (measured 321.5 -> 315.5 ms per step, k_var 51.5 -> 49.9 us); with two lanes the other lane's kernel
    // lives in that tail and the same order costs 0.8 % (HQC-128: 95.2 -> 96.0 ms): there the columns of a
    // degree are ordered by their first edge instead (profiles/r02/ab_*order*.json)
    const bool single_lane = auto_group(h, 1 << 20) < 2 || h->kn.split < 2;
    const int var_order = h->kn.var_order >= 0 ? h->kn.var_order : (single_lane ? 2 : 1);
    if ((var_order & 1) && !hv.list.empty()) {
        order_buf = hv.list;
        size_t i = 0;
        while (i < order_buf.size()) {
            size_t j = i;
            while (j < order_buf.size() && cdeg[order_buf[j]] == cdeg[order_buf[i]]) j++;
            if (cdeg[order_buf[i]] > 0)
                std::sort(order_buf.begin() + i, order_buf.begin() + j,
                          [&](int a, int b2) { return csc_edge[col_ptr[a]] < csc_edge[col_ptr[b2]]; });
            i = j;
        }
        vlist = order_buf.data();
    }
    for (int b = 0; b < hv.bk.nb; b++) {
        const int blocks = hv.bk.blk[b + 1] - hv.bk.blk[b];
        for (int sl = 0; sl < blocks * 4; sl++) {
            int *md = meta + (size_t)VAR_REC * ((size_t)hv.bk.blk[b] * 4 + sl);
            int *mr = host + o_var_rows + (size_t)VAR_INLINE * ((size_t)hv.bk.blk[b] * 4 + sl);
            if (sl >= hv.bk.cnt[b]) {
                md[0] = -1;
                for (int k = 1; k < VAR_REC; k++) md[k] = 0;
                for (int k = 0; k < VAR_INLINE; k++) mr[k] = 0;
                continue;
            }
            const int v = vlist[hv.bk.off[b] + sl], d = cdeg[v];
            md[0] = v;
            md[1] = pos;
            md[2] = d;
            md[3] = hv.bk.maxd[b];
            for (int k = 0; k < d; k++) {
                relaid[pos + k] = csc_edge[(size_t)col_ptr[v] + k];
                relaid_row[pos + k] = edge_row[relaid[pos + k]];
                host[o_csr_pos + relaid[pos + k]] = pos + k;
            }
            for (int k = 0; k < VAR_INLINE; k++) md[4 + k] = k < d ? relaid[pos + k] : 0;
            for (int k = 0; k < VAR_INLINE; k++) mr[k] = k < d ? relaid_row[pos + k] : 0;
            pos += d;
        }
    }
    h->var_reversed = (var_order & 2) != 0;
    if (var_order & 2) {
        // heaviest columns FIRST: the waves that start last are then the cheapest ones (degree-1 identity
        // columns), which shortens the tail of the launch (longest-processing-time-first)
        const size_t nrec = (size_t)4 * hv.bk.blk[hv.bk.nb];
        int *rows = host + o_var_rows;
        for (size_t i = 0, j = nrec ? nrec - 1 : 0; i < j; i++, j--) {
            std::swap_ranges(meta + i * VAR_REC, meta + (i + 1) * VAR_REC, meta + j * VAR_REC);
            std::swap_ranges(rows + i * VAR_INLINE, rows + (i + 1) * VAR_INLINE, rows + j * VAR_INLINE);
        }
    }
    SC_TRY(upload_table(&h->d_tile_tab, host, total));
    h->first_valid = false;  // (first_tab follows the re-laid edge list)
    h->d_var_meta = h->d_tile_tab + o_var_meta;
    h->d_csc_list = h->d_tile_tab + o_csc_list;
    h->d_row_list = h->d_tile_tab + o_row_list;
    h->d_csc_row = h->d_tile_tab + o_csc_row;
    h->d_var_rows = h->d_tile_tab + o_var_rows;
    h->d_csr_pos = h->d_tile_tab + o_csr_pos;
    return 0;
}

// ---- tables of the row-parallel path (k_el_var) ------------------------------------------------
// d_el_tab = [slots: 2 ints per lane slot][slot_col: 1 int per lane slot] for el_cap_bins bins of 64
// lane slots; h_el mirrors it on the host.
inline int el_tag(int start, int pos, int cap, bool live) { return start | (pos << 6) | ((cap - 1) << 12) | ((live ? 1 : 0) << 18); }
inline size_t el_col_off(const scaldpc_bp *h) { return (size_t)128 * h->el_cap_bins; }

// free lanes a column's segment gets on top of its degree: none on a decoder whose graph is fixed
// (and for the degree-1 columns of an identity block, which never grow); a few on one that grows
int el_slack(const scaldpc_bp *h, int col, int deg)
{
    if (!h->incremental) return 0;
    if (h->identity_from >= 0 && col >= h->identity_from) return 0;
    return 2 + deg / 4;
}

// hands out `cap` neighbouring lanes (a segment never wraps around a bin); returns bin * 64 + start.
// The host mirrors grow with the bins (dead lanes: no edge, not live); the device table is
// re-uploaded by the caller when the bins outgrow el_cap_bins.
int el_alloc_segment(scaldpc_bp *h, int cap)
{
    if (h->el_waves == 0 || h->el_open_used + cap > 64) {
        h->el_waves++;
        h->el_open_used = 0;
        if (h->h_el_col.size() < (size_t)64 * h->el_waves) {
            const size_t lanes = (size_t)64 * (h->el_waves + h->el_waves / 2 + 16), old = h->h_el_col.size();
            h->h_el_slots.resize(2 * lanes, 0);
            h->h_el_col.resize(lanes, 0);
            for (size_t i = old; i < lanes; i++) h->h_el_slots[2 * i] = -1;
        }
    }
    const int slot = (h->el_waves - 1) * 64 + h->el_open_used;
    h->el_open_used += cap;
    return slot;
}

// device copy of the mirrors: [slots: 2 ints per lane][slot_col: 1 int per lane] for el_cap_bins bins.
// EVERY lane of the table is initialised (bins not yet in use hold dead lanes): an append that opens a
// new bin only sends the words it changes, and a launch covers whole bins.
int el_upload(scaldpc_bp *h)
{
    dev_free(h->d_el_tab);
    h->el_cap_bins = h->incremental ? h->el_waves + h->el_waves / 2 + 64 : h->el_waves;
    const size_t lanes = (size_t)64 * h->el_cap_bins, old = h->h_el_col.size();
    h->h_el_slots.resize(2 * lanes, 0);
    h->h_el_col.resize(lanes, 0);
    for (size_t i = old; i < lanes; i++) h->h_el_slots[2 * i] = -1;
    SC_TRY(dev_alloc(&h->d_el_tab, 3 * lanes));
    h->d_el_slots = h->d_el_tab;
    h->d_el_slot_col = h->d_el_tab + el_col_off(h);
    if (lanes) {
        SC_HIP(hipMemcpy(h->d_el_slots, h->h_el_slots.data(), 2 * lanes * sizeof(int), hipMemcpyHostToDevice));
        SC_HIP(hipMemcpy(h->d_el_slot_col, h->h_el_col.data(), lanes * sizeof(int), hipMemcpyHostToDevice));
    }
    return 0;
}

// (re)builds host mirrors + device table from the host graph mirror: columns in degree order (hg_var.list)
int refresh_full(scaldpc_bp *h);
int ensure_el_tables(scaldpc_bp *h)
{
    if (h->d_el_tab || !h->el_ok) return 0;
    SC_TRY(refresh_full(h));  // built from the host's CSC mirror: bring it up to date with appended rows first
    const HostBuckets &hv = h->hg_var;
    const int *cdeg = h->hg_cdeg.data(), *col_ptr = h->hg_col_ptr.data(), *csc_edge = h->hg_csc_edge.data();
    h->seg_slot.assign(h->n, -1);
    h->seg_cap.assign(h->n, 0);
    h->el_waves = 0;
    h->el_open_used = 0;
    h->h_el_slots.clear();
    h->h_el_col.clear();
    for (size_t i = 0; i < hv.list.size(); i++) {
        const int v = hv.list[i], d = cdeg[v];
        const int cap = std::min(64, std::max(d, 1) + el_slack(h, v, d));
        const int s0 = el_alloc_segment(h, cap), start = s0 & 63;
        h->seg_slot[v] = s0;
        h->seg_cap[v] = (unsigned char)cap;
        const int *ce = csc_edge + col_ptr[v];
        for (int k = 0; k < cap; k++) {
            h->h_el_slots[2 * (size_t)(s0 + k)] = k < d ? ce[k] : -1;
            h->h_el_slots[2 * (size_t)(s0 + k) + 1] = el_tag(start, k, cap, true);
        }
        h->h_el_col[s0] = v;
    }
    return el_upload(h);
}

// Min-sum in its record form (k_check_minsum_rec / k_var_rec)?  Rows and columns that live in registers, the
// exact-degree check kernels, the scalar-id variable kernel.
bool rec_form(const scaldpc_bp *h, int method)
{
    // (columns of 33-64 edges: message form -- the exact-degree variable pass is compiled up to 32)
    return method == SCALDPC_BP_MIN_SUM && h->kn.minsum_rec && h->E > 0 && h->max_row_deg <= 64 && h->max_col_deg <= 32 &&
           !h->hg_var.has_generic;
}

// message array of the tile path, G tiles (and the records of the min-sum record form)
int ensure_msg(scaldpc_bp *h, int G, int method)
{
    SC_TRY(ensure_tile_tables(h));
    if (rec_form(h, method) && G > h->cap_rec_group) {  // (append_rows resets cap_rec_group too)
        dev_free(h->d_rec); dev_free(h->d_mask);
        h->cap_rec_group = 0;
        SC_TRY(dev_alloc(&h->d_rec, (size_t)G * h->m * 2 * TW));
        SC_TRY(dev_alloc(&h->d_mask, (size_t)G * h->E));
        h->cap_rec_group = G;
    }
    if (G > h->cap_group) {  // (append_rows resets cap_group: the arrays are sized by E)
        dev_free(h->d_msg); dev_free(h->d_scratch);
        h->cap_group = 0;
        SC_TRY(dev_alloc(&h->d_msg, (size_t)G * h->E * TW));
        if (h->need_scratch) SC_TRY(dev_alloc(&h->d_scratch, (size_t)G * h->E * TW));
        h->cap_group = G;
    }
    return 0;
}

// message + prefix arrays of the row-parallel path, nb codewords
int ensure_el(scaldpc_bp *h, int nb)
{
    SC_TRY(ensure_el_tables(h));
    const size_t need = (size_t)nb * h->E;
    if (need > h->cap_el || !h->d_emsg) {
        dev_free(h->d_emsg);
        h->cap_el = 0;
        const size_t want = h->incremental ? need + need / 4 : need;  // room for the edges still to come
        SC_TRY(dev_alloc(&h->d_emsg, want));
        h->cap_el = want;
    }
    return 0;
}

// How many codewords the row-parallel kernels take (0 = none: use 64-codeword tiles).
// Their cost grows with the codeword count (12 us per iteration for one codeword, +3.7 us per
// further one with min-sum, 16 / +6.5 us with the tanh rule, HQC-128 bench graph), a tile's
// does not (40 / 43 us): the default limit is where the two meet
// (profiles/microbench/small_batch_latency.py).  SCALDPC_PATH=edge lifts the limit to a whole
// tile, SCALDPC_PATH=stream disables the path (tests pin either).
int el_limit(const scaldpc_bp *h, int method)
{
    if (!h->el_ok) return 0;  // empty graph, or a row / column wider than a wave
    int lim = method == SCALDPC_BP_MIN_SUM ? 6 : 4;
    if (h->kn.el_max >= 0) lim = h->kn.el_max;
    if (h->kn.path == Knobs::STREAM) lim = 0;
    if (h->kn.path == Knobs::EDGE) lim = TW;
    return std::max(0, std::min(lim, TW));
}

#define LAUNCH_CHECK() SC_HIP(hipGetLastError())

// No message initialisation pass: the first check update reads the priors (v2c = prior of the
// edge's column by definition).  The tanh rule's any-degree fallback (rows wider than 64) keeps it.
bool fused_init(const scaldpc_bp *h, int method)
{
    return h->E > 0 && (method == SCALDPC_BP_MIN_SUM || h->max_row_deg <= ROW_CAP);
}

// Which check kernels can carry the convergence test of the previous iteration (fused_test)?  The register-resident
// ones: the tanh kernel always, min-sum in its exact-degree form.
bool check_can_test(const scaldpc_bp *h, int method)
{
    if (h->E == 0) return false;
    if (method == SCALDPC_BP_MIN_SUM) return h->max_row_deg <= ROW_CAP;
    return true;
}

// ft: non-null = this pass also runs the convergence test of the previous iteration (check_can_test(h, method), never `first`)
int launch_check(scaldpc_bp *h, int method, float alpha, int G, const u64 *synd_g, const u64 *done_g, int skip_done,
                 hipStream_t s, bool first = false, int tile0 = 0, const FusedTest *ft = nullptr)
{
    if (h->E == 0) return 0;
    float *const msg0 = h->d_msg + (size_t)tile0 * h->E * TW;  // tile0: first tile of a sub-group inside the group's array
    float *const scr0 = h->d_scratch ? h->d_scratch + (size_t)tile0 * h->E * TW : nullptr;
    if (method == SCALDPC_BP_MIN_SUM) {
        dim3 grid((h->m + 3) / 4, G);
#define MS_LAUNCH(F)                                                                                                 \
    hipLaunchKernelGGL((k_check_minsum<F>), grid, dim3(256), 0, s, h->d_row_ptr, msg0, synd_g, done_g, skip_done, \
                       h->m, h->E, alpha, h->d_col_idx, h->d_prior)
        // (a first check pass -- iteration 1 with first_fused off -- reads the priors and writes MESSAGES: the message
        // form's kernel, followed by the message form's variable pass; the records start with iteration 2)
        if (rec_form(h, method) && !first) {
            dim3 gridx(h->row_bk.blk[h->row_bk.nb], G);
            float *const rec0 = h->d_rec + (size_t)tile0 * h->m * 2 * TW;
            ulonglong2 *const mask0 = h->d_mask + (size_t)tile0 * h->E;
#define MSR_LAUNCH(CAP)                                                                                             \
    hipLaunchKernelGGL((k_check_minsum_rec<CAP>), gridx, dim3(256), 0, s, h->d_row_list, msg0, synd_g, done_g, skip_done, \
                       h->m, h->E, alpha, h->d_col_idx, rec0, mask0, h->d_csr_pos)
#define MSR_PAR(CAP)                                                                                                \
    hipLaunchKernelGGL((k_check_minsum_rec<CAP, true>), gridx, dim3(256), 0, s, h->d_row_list, msg0, synd_g, done_g, \
                       skip_done, h->m, h->E, alpha, h->d_col_idx, rec0, mask0, h->d_csr_pos, *ft)
            if (ft) {
                if (h->max_row_deg <= 16) MSR_PAR(16); else if (h->max_row_deg <= 32) MSR_PAR(32); else MSR_PAR(64);
            } else if (h->max_row_deg <= 16) {
                MSR_LAUNCH(16);
            } else if (h->max_row_deg <= 32) {
                MSR_LAUNCH(32);
            } else {
                MSR_LAUNCH(64);
            }
#undef MSR_PAR
#undef MSR_LAUNCH
        } else if (h->max_row_deg <= ROW_CAP) {
            dim3 gridx(h->row_bk.blk[h->row_bk.nb], G);
#define MSX_LAUNCH(CAP, F)                                                                                          \
    hipLaunchKernelGGL((k_check_minsum_x<CAP, F>), gridx, dim3(256), 0, s, h->d_row_list, msg0, synd_g, done_g, skip_done, \
                       h->m, h->E, alpha, h->d_col_idx, h->d_prior)
#define MSX_PAR(CAP)                                                                                                \
    hipLaunchKernelGGL((k_check_minsum_x<CAP, false, true>), gridx, dim3(256), 0, s, h->d_row_list, msg0, synd_g, done_g,  \
                       skip_done, h->m, h->E, alpha, h->d_col_idx, h->d_prior, *ft)
            if (ft && !first) {
                if (h->max_row_deg <= 16) MSX_PAR(16); else if (h->max_row_deg <= 32) MSX_PAR(32); else MSX_PAR(64);
            } else if (h->max_row_deg <= 16) {
                if (first) MSX_LAUNCH(16, true); else MSX_LAUNCH(16, false);
            } else if (h->max_row_deg <= 32) {
                if (first) MSX_LAUNCH(32, true); else MSX_LAUNCH(32, false);
            } else {
                if (first) MSX_LAUNCH(64, true); else MSX_LAUNCH(64, false);
            }
#undef MSX_PAR
#undef MSX_LAUNCH
        } else {  // a row wider than 64: the loop form
            if (first) MS_LAUNCH(true); else MS_LAUNCH(false);
        }
#undef MS_LAUNCH
    } else {
        dim3 grid(h->row_bk.blk[h->row_bk.nb], G);
#define TANH_LAUNCH(CAP, F)                                                                                         \
    hipLaunchKernelGGL((k_check_tanh<CAP, F>), grid, dim3(256), 0, s, h->row_bk, h->d_row_list, h->d_row_ptr, msg0, \
                       scr0, synd_g, done_g, skip_done, h->m, h->E, h->d_col_idx, h->d_prior)
#define TANH_PAR(CAP)                                                                                               \
    hipLaunchKernelGGL((k_check_tanh<CAP, false, true>), grid, dim3(256), 0, s, h->row_bk, h->d_row_list, h->d_row_ptr, msg0, \
                       scr0, synd_g, done_g, skip_done, h->m, h->E, h->d_col_idx, h->d_prior, *ft)
        if (ft && !first) {
            if (h->max_row_deg <= 16) TANH_PAR(16); else if (h->max_row_deg <= 32) TANH_PAR(32); else TANH_PAR(64);
        } else if (h->max_row_deg <= 16) {
            if (first) TANH_LAUNCH(16, true); else TANH_LAUNCH(16, false);
        } else if (h->max_row_deg <= 32) {
            if (first) TANH_LAUNCH(32, true); else TANH_LAUNCH(32, false);
        } else {
            if (first) TANH_LAUNCH(64, true); else TANH_LAUNCH(64, false);
        }
#undef TANH_PAR
#undef TANH_LAUNCH
    }
    LAUNCH_CHECK();
    return 0;
}

// Can iteration 1 run without its check pass (k_var_first)?  The first messages come from the row-parallel check
// kernel (rows of at most 64 edges), the first variable pass keeps every column in registers (columns of at most 64).
bool first_fusable(const scaldpc_bp *h, int method)
{
    return h->kn.first_fused && fused_init(h, method) && h->E > 0 && h->max_row_deg <= 64 && h->max_col_deg <= 64 &&
           !h->hg_var.has_generic;
}

// first_tab for (method, alpha of iteration 1) and the current priors: the row-parallel check kernel decodes the first
// pass of ONE codeword with an all-zero syndrome into a scratch array (its values are the tile kernels' own, bit for
// bit), k_first_tab lays them out beside each edge's row in the order of the re-laid edge list.
int ensure_first_table(scaldpc_bp *h, int method, float alpha1, hipStream_t s)
{
    if (h->first_valid && h->first_method == method && h->first_alpha == alpha1) return 0;
    const size_t need = (size_t)h->E + 128;  // (a column's scalar loads may run past its last edge, as in the edge list)
    if (need > h->cap_first) {
        dev_free(h->d_first_tab);
        h->cap_first = 0;
        SC_TRY(dev_alloc(&h->d_first_tab, need + need / 8));
        h->cap_first = need + need / 8;
    }
    float *tmp = nullptr;
    u64 *zero = nullptr;
    SC_TRY(dev_alloc(&tmp, (size_t)h->E));
    int rc = dev_alloc(&zero, (size_t)h->m + 1);
    if (rc) {
        dev_free(tmp);
        return rc;
    }
    hipError_t e = hipMemsetAsync(zero, 0, sizeof(u64) * ((size_t)h->m + 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_first_tab, 0, sizeof(int2) * h->cap_first, s);
    if (e == hipSuccess) {
        dim3 grid((h->m + 3) / 4, 1);
        if (method == SCALDPC_BP_MIN_SUM)
            hipLaunchKernelGGL((k_el_check<SCALDPC_BP_MIN_SUM, true>), grid, dim3(256), 0, s, h->d_row_ptr, h->d_col_idx, h->d_prior,
                               tmp, zero, zero + h->m, 0, h->m, h->E, alpha1, (const u64 *)nullptr, (int *)nullptr);
        else
            hipLaunchKernelGGL((k_el_check<SCALDPC_BP_PRODUCT_SUM, true>), grid, dim3(256), 0, s, h->d_row_ptr, h->d_col_idx,
                               h->d_prior, tmp, zero, zero + h->m, 0, h->m, h->E, alpha1, (const u64 *)nullptr, (int *)nullptr);
        hipLaunchKernelGGL(k_first_tab, dim3((unsigned)((h->E + 255) / 256)), dim3(256), 0, s, h->d_csc_list, tmp, h->d_row_ptr, h->m,
                           h->E, h->d_first_tab);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // tmp / zero go back to the allocator below
    dev_free(tmp);
    dev_free(zero);
    if (e != hipSuccess) return fail(SCALDPC_EHIP, "first-message table: %s", hipGetErrorString(e));
    h->first_valid = true;
    h->first_method = method;
    h->first_alpha = alpha1;
    return 0;
}

// first_synd: non-null = iteration 1 without its check pass (the group's syndrome planes; ensure_first_table first)
int launch_var(scaldpc_bp *h, int G, float *post_g, u64 *hard_g, const u64 *done_g, int skip_done, int write_out,
               hipStream_t s, int tile0 = 0, const u64 *first_synd = nullptr, bool rec = false, bool light = false)
{
    dim3 grid(h->var_bk.blk[h->var_bk.nb], G);
    float *const msg0 = h->d_msg + (size_t)tile0 * h->E * TW;
    float *const scr0 = h->d_scratch ? h->d_scratch + (size_t)tile0 * h->E * TW : nullptr;
    if (first_synd) {
        const int nrec = 4 * h->var_bk.blk[h->var_bk.nb];  // one record per wave of a plain k_var launch; two per wave here
        const dim3 grid2((unsigned)((nrec + 7) / 8), G);
#define VAR_FIRST(CAP)                                                                                              \
    hipLaunchKernelGGL((k_var_first<CAP>), grid2, dim3(256), 0, s, h->d_var_meta, h->d_csc_list, h->d_prior, msg0, post_g,  \
                       hard_g, done_g, skip_done, h->n, h->E, write_out, (const int2 *)h->d_first_tab, first_synd, h->m, nrec)
        if (h->max_col_deg <= 16)
            VAR_FIRST(16);
        else if (h->max_col_deg <= 32)
            VAR_FIRST(32);
        else
            VAR_FIRST(64);
#undef VAR_FIRST
        LAUNCH_CHECK();
        return 0;
    }
    if (rec) {  // the check pass left records, not messages (rec_form)
        // light: a pass without output after iteration 1.  A column of degree <= 1 always sends its prior; iteration 1
        // has written that into the message array and the record check pass never overwrites it, so such a pass has
        // nothing to do for the first bucket (the identity block of an HQC graph: 4000 of 21669 column waves per tile).
        const int nblk = h->var_bk.blk[h->var_bk.nb];
        const int n1 = (h->var_bk.nb > 0 && h->var_bk.maxd[0] == 1) ? h->var_bk.blk[1] : 0;
        const bool slim = light && !write_out && h->kn.rec_skip1 && n1 > 0 && n1 < nblk;
        const int nb_launch = slim ? nblk - n1 : nblk;
        // XCD-aware tile placement where the launch's tiles divide the 8 XCDs (fetch 38.0 -> 29.5 MB per launch, profiles/r03/ab_rec_xmap.log)
        const bool xm = G == 2 || G == 4 || G == 8;
        const dim3 gridr((unsigned)(xm ? (nb_launch + 7) / 8 * 8 : nb_launch), G);
        const int blk0 = slim && !h->var_reversed ? n1 : 0, xmap = xm ? nb_launch : 0;
#define VAR_REC_LAUNCH(CAP)                                                                                         \
    hipLaunchKernelGGL((k_var_rec<CAP>), gridr, dim3(256), 0, s, h->d_var_meta, h->d_var_rows, h->d_csc_list, h->d_csc_row, \
                       h->d_prior, msg0, h->d_rec + (size_t)tile0 * h->m * 2 * TW, h->d_mask + (size_t)tile0 * h->E, post_g,  \
                       hard_g, done_g, skip_done, h->n, h->m, h->E, write_out, blk0, xmap)
        if (h->max_col_deg <= 16) VAR_REC_LAUNCH(16); else VAR_REC_LAUNCH(32);
#undef VAR_REC_LAUNCH
        LAUNCH_CHECK();
        return 0;
    }
#define VAR_LAUNCH(CAP)                                                                                             \
    hipLaunchKernelGGL((k_var<CAP>), grid, dim3(256), 0, s, h->var_bk, h->d_var_meta, h->d_col_ptr, h->d_csc_list, \
                       h->d_prior, msg0, scr0, post_g, hard_g, done_g, skip_done, h->n, h->E, write_out)
    if (h->max_col_deg <= 16)
        VAR_LAUNCH(16);
    else if (h->max_col_deg <= 32)
        VAR_LAUNCH(32);
    else
        VAR_LAUNCH(64);
#undef VAR_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

int launch_el_check(scaldpc_bp *h, int method, float alpha, int nb, const u64 *synd_g, const u64 *done_g,
                    int skip_done, hipStream_t s, bool first, const u64 *hard_g = nullptr, int *unsat_prev = nullptr)
{
    dim3 grid((h->m + 3) / 4, nb);
#define EL_LAUNCH(M, F)                                                                                             \
    hipLaunchKernelGGL((k_el_check<M, F>), grid, dim3(256), 0, s, h->d_row_ptr, h->d_col_idx, h->d_prior, h->d_emsg,  \
                       synd_g, done_g, skip_done, h->m, h->E, alpha, hard_g, unsat_prev)
    if (method == SCALDPC_BP_MIN_SUM) {
        if (first) EL_LAUNCH(SCALDPC_BP_MIN_SUM, true); else EL_LAUNCH(SCALDPC_BP_MIN_SUM, false);
    } else {
        if (first) EL_LAUNCH(SCALDPC_BP_PRODUCT_SUM, true); else EL_LAUNCH(SCALDPC_BP_PRODUCT_SUM, false);
    }
#undef EL_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

int launch_el_var(scaldpc_bp *h, int nb, float *post_g, u64 *hard_g, u64 *done_g, int skip_done, int write_out,
                  hipStream_t s, const int *unsat_prev = nullptr, int it_prev = 0, u64 *conv_g = nullptr,
                  int *iters_g = nullptr, int *remaining_prev = nullptr)
{
    // grid.x a multiple of 8: block x lands on the same XCD for every codeword row, so an XCD's L2
    // keeps its share of the slot table
    const unsigned gx = (unsigned)(((h->el_waves + 3) / 4 + 7) / 8 * 8);
    hipLaunchKernelGGL(k_el_var, dim3(gx, nb), dim3(256), 0, s, (const int2 *)h->d_el_slots, h->d_el_slot_col,
                       h->el_waves, h->d_prior, h->d_emsg, post_g, hard_g, done_g, skip_done, h->E,
                       write_out, unsat_prev, it_prev, conv_g, iters_g, remaining_prev);
    LAUNCH_CHECK();
    return 0;
}

float alpha_for(float alpha, int it)
{
    // ms_scaling_factor == 0 -> 1 - 2^-iter
    return alpha == 0.0f ? (float)(1.0 - std::pow(2.0, -1.0 * it)) : alpha;
}

struct TileState {
    const u64 *synd;
    u64 *hard, *done, *conv, *unsat;
    int *iters;
    float *post;
};

// All iterations of the tile group [g0, g0+g) of `st`.  With defer_after > 0 the group
// stops at the first poll point from that iteration on at which at most half of its
// codewords are still running, and reports *deferred = true: those go to the compact pass.
// Early-exit loop of the row-parallel path: two launches per iteration.  check(it) also tests
// H e == s on the decisions of iteration it-1, var(it) latches the codewords that passed
// (frozen at it-1) before it updates the others; the last iteration is followed by the
// ordinary k_parity / k_finalize pair.  The host looks at "still running after it-1" one
// iteration late, so a finished call enqueues one iteration of (skipped) launches more than
// the four-launch form -- and half as many overall.
int ensure_el_unsat(scaldpc_bp *h, int max_iter)
{
    const size_t flags = ((size_t)max_iter + 2) * TW;
    if (flags > h->cap_el_unsat || !h->d_el_unsat) {
        dev_free(h->d_el_unsat);
        h->cap_el_unsat = 0;
        SC_TRY(dev_alloc(&h->d_el_unsat, flags));
        h->cap_el_unsat = flags;
    }
    return 0;
}

int iterate_el_early(scaldpc_bp *h, int nb, int max_iter, int method, float alpha, const u64 *synd_g, u64 *hard_g,
                     u64 *done_g, u64 *conv_g, u64 *unsat_g, int *iters_g, float *post_g, hipStream_t s)
{
    const size_t flags = ((size_t)max_iter + 2) * TW;
    SC_TRY(ensure_el_unsat(h, max_iter));
    if (!h->small_prepared) {
        SC_HIP(hipMemsetAsync(h->d_el_unsat, 0, sizeof(int) * flags, s));
        SC_HIP(hipMemsetAsync(h->d_remaining, 0, sizeof(int) * ((size_t)max_iter + 2), s));
    }
    const bool fused = fused_init(h, method);
    if (!fused) {
        hipLaunchKernelGGL(k_el_init, dim3((unsigned)((h->E + 255) / 256), nb), dim3(256), 0, s, h->d_col_idx, h->d_prior,
                           h->d_emsg, h->E);
        LAUNCH_CHECK();
    }
    for (int it = 1; it <= max_iter; it++) {
        int *up = it > 1 ? h->d_el_unsat + (size_t)(it - 1) * TW : nullptr;
        SC_TRY(launch_el_check(h, method, alpha_for(alpha, it), nb, synd_g, done_g, 1, s, fused && it == 1, hard_g, up));
        SC_TRY(launch_el_var(h, nb, post_g, hard_g, done_g, 1, 1, s, up, it - 1, conv_g, iters_g, h->d_remaining + it - 1));
        if (it == max_iter) {
            hipLaunchKernelGGL(k_parity<true>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), 1), dim3(256), 0,
                               s, h->d_row_ptr, h->d_col_idx, hard_g, h->m, h->n, const_cast<u64 *>(synd_g), unsat_g,
                               (const u64 *)done_g);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, it, 1, done_g, conv_g, unsat_g, parity_waves(h), iters_g,
                               h->d_remaining + it);
            LAUNCH_CHECK();
            break;
        }
        // Poll sparsely: a poll drains the queue and costs about as much as an iteration here,
        // while iterations enqueued for codewords that turn out to be finished return at once.
        const int ip = it - 1;  // the iteration whose verdict var(it) just latched
        // Graphs that come here are too large for LDS: a decode that converges does so in 3-5 iterations (the
        // attack loop once enough checks are in; measured 3-4 on the HQC-128 graph), one that does not runs to
        // max_iter.  Poll densely where convergence is likely, sparsely afterwards.
        if ((ip >= 3 && ip <= 6) || (ip > 6 && ip <= 16 && ip % 2 == 0) || (ip > 16 && ip % 16 == 0)) {
            SC_HIP(hipMemcpyAsync(h->h_remaining + ip, h->d_remaining + ip, sizeof(int), hipMemcpyDeviceToHost, s));
            SC_HIP(hipStreamSynchronize(s));
            if (h->h_remaining[ip] == 0) break;
        }
    }
    return 0;
}

// All iterations of one tile group on the 64-codeword-tile kernels, on SEVERAL streams
// ("lanes"): the group's tiles are dealt to lanes whose launch sequences (independent: a pass
// depends only on the previous pass over the same tiles) run one kernel out of phase, so that
// while one lane drains the tail of its kernel another lane's kernel fills the machine.  Same
// kernels, same cache footprint, same results.  Early-exit runs join the lanes at every poll.
constexpr int MAX_LANES = 4;
// measured on the HQC-128 bench (4-tile groups, ms per 4096-codeword step): 1 lane 104.0,
// 2 lanes 98.5, 3 lanes 101.3, 4 lanes 106.8 -- two kernels in flight fill each other's tails,
// more only shrink the launches.  SCALDPC_SPLIT=n overrides (1 = single stream).
int fixed_lanes(const scaldpc_bp *h, int g)
{
    const int nl = h->kn.split;
    if (h->E == 0) return 1;
    return std::max(1, std::min(std::min(nl, g), MAX_LANES));
}
int ensure_lanes(scaldpc_bp *h, int nl)
{
    for (int k = 1; k < nl; k++)
        if (!h->aux_stream[k]) {
            int dev = 0;
            SC_TRY(stream_acquire(&h->aux_stream[k], &dev));
            SC_HIP(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
            SC_HIP(hipEventCreateWithFlags(&h->ev_phase[k], hipEventDisableTiming));
        }
    if (nl > 1 && !h->ev_join[0]) SC_HIP(hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming));  // [0]: the fork event
    return 0;
}
// What the groups of one call learn from each other about polling (a poll drains the queue: the GPU idles
// while the host turns around, ~4 % of a group's time on the config-5 sweep):
//   hint   poll points before this iteration saw no codeword finish in earlier groups: skip them
//   streak consecutive groups that handed a small remainder (<= 1/8) to the compact pass at `defer_after`:
//          after two of them the next groups stop there WITHOUT polling (their stragglers are found from the
//          done masks afterwards, as always), with a real poll every 16th group to keep the assumption honest.
//          A group stopped on a wrong guess only sends more codewords to the compact pass, which decodes
//          them from their inputs: results cannot change.
constexpr int SPEC_PERIOD = 16;  // (64 and 4096 measured the same to 0.5 % on the config-5 sweep: the polls are not where its time goes)
struct PollState {
    int hint = 1, streak = 0, since_poll = 0;
};

// chain: bit 0 = this group CONTINUES the lanes of the group before it (same geometry, fixed iterations): no fork -- a lane's
//        work on the new group is ordered behind its own work on the previous one by its stream, and nothing else feeds it;
//        bit 1 = the group after this one will continue them: no join at the end.  Without the per-group join / fork the lane
//        that runs a kernel ahead flows straight into the next group instead of idling for the other lane's last pass (and
//        the other lane for its first): ~75 us per group of 3.7 ms on the HQC-128 bench.
int iterate_tiles(scaldpc_bp *h, const TileState &st, int g0, int g, int max_iter, int method, float alpha, bool early,
                  int defer_after, hipStream_t s, bool *deferred, int real_codewords, PollState *ps, int chain = 0)
{
    int *const poll_hint = ps ? &ps->hint : nullptr;
    const int poll_every = 4;
    const int skip = early ? 1 : 0;
    const int nl = fixed_lanes(h, g);
    SC_TRY(ensure_lanes(h, nl));
    hipStream_t lane[MAX_LANES];
    int gs[MAX_LANES], t0[MAX_LANES];
    for (int k = 0, t = 0; k < nl; k++) {
        lane[k] = k ? h->aux_stream[k] : s;
        gs[k] = g / nl + (k < g % nl ? 1 : 0);
        t0[k] = t;
        t += gs[k];
    }
    const int pw = parity_waves(h);
    const bool fused = fused_init(h, method);
    if (h->E && !fused) {
        hipLaunchKernelGGL(k_init_msg, dim3((unsigned)((h->E + 3) / 4), g), dim3(256), 0, s, h->d_col_idx, h->d_prior,
                           h->d_msg, h->E);
        LAUNCH_CHECK();
    }
    int *rem = h->d_remaining;  // this group's row of "still running" counters
    if (early) SC_TRY(next_remaining_row(h, s, &rem));
    // (the accumulators / block counters of k_parity_fin and fused_commit are zeroed once per level, in decode_level,
    // and every launch leaves them zero)
    if (nl > 1 && !(chain & 1)) {  // fork: the other lanes start after everything enqueued on `s` so far
        SC_HIP(hipEventRecord(h->ev_join[0], s));
        for (int k = 1; k < nl; k++) SC_HIP(hipStreamWaitEvent(lane[k], h->ev_join[0], 0));
    }
    auto join = [&]() -> int {
        for (int k = 1; k < nl; k++) {
            SC_HIP(hipEventRecord(h->ev_join[k], lane[k]));
            SC_HIP(hipStreamWaitEvent(s, h->ev_join[k], 0));
        }
        return 0;
    };
    // iteration 1 without its check pass: the first variable pass reads the first-message table and the syndrome planes
    const bool first_fuse = first_fusable(h, method);
    if (first_fuse) SC_TRY(ensure_first_table(h, method, alpha_for(alpha, 1), s));
    // The convergence test of iteration it rides on the check pass of it + 1 (fused_test) wherever the host does not need
    // its verdict in between: at the iterations it polls at, stops a group at, or ends with, the stand-alone launch stays.
    // (Running the test on a side stream beside the check pass instead was measured and rejected: -3.4 % on the config-5
    // sweep, profiles/r03/ab_test_overlap.log; so was the two-launch k_parity + k_finalize form in this loop.)
    const bool ride = early && h->kn.fuse_test && check_can_test(h, method) && pw >= FT_WORDS;
    bool verdict_pending[MAX_LANES] = {};  // lane k's last variable pass has not been tested yet: its next check pass will
    // (re-)establish the one-kernel offset between neighbouring lanes -- in a chained group too: carrying the offset over
    // instead measured +0.5 % on the config-5 sweep and -2 % on the fixed-iteration bench, profiles/r04/ab_chain_unseen_groups.log
    bool set_phase = true;
    for (int it = 1; it <= max_iter; it++) {
        const bool last = it == max_iter;
        const bool no_check = first_fuse && it == 1;
        // (decided before anything of this iteration is enqueued: the host's own sync points)
        const bool poll = (it % poll_every == 0 || it == 1 || it == defer_after) && (!poll_hint || it >= *poll_hint);
        for (int k = 0; k < nl && !no_check; k++) {
            const int ta = g0 + t0[k];
            FusedTest ft{};
            if (verdict_pending[k]) {
                ft = FusedTest{st.hard + (size_t)ta * h->n, st.unsat + (size_t)ta * pw, st.done + ta, st.conv + ta,
                               st.iters + (size_t)ta * TW, rem + (it - 1), h->n, pw, it - 1, 1};
                verdict_pending[k] = false;
            }
            SC_TRY(launch_check(h, method, alpha_for(alpha, it), gs[k], st.synd + (size_t)ta * h->m, st.done + ta, skip, lane[k],
                                fused && it == 1, t0[k], ft.hard ? &ft : nullptr));
            if (set_phase && k + 1 < nl) {
                SC_HIP(hipEventRecord(h->ev_phase[k + 1], lane[k]));
                SC_HIP(hipStreamWaitEvent(lane[k + 1], h->ev_phase[k + 1], 0));
            }
        }
        if (!no_check) set_phase = false;  // (a first iteration without check passes leaves the offset to the second)
        for (int k = 0; k < nl; k++) {
            const int ta = g0 + t0[k];
            // (the record form starts with iteration 2: an iteration 1 that has a check pass ran it in the message form)
            SC_TRY(launch_var(h, gs[k], st.post ? st.post + (size_t)ta * h->n * TW : nullptr, st.hard + (size_t)ta * h->n,
                              st.done + ta, skip, (early || last) ? 1 : 0, lane[k], t0[k],
                              no_check ? st.synd + (size_t)ta * h->m : nullptr, it > 1 && rec_form(h, method), it > 1));
            if (ride && !last && !poll) {
                verdict_pending[k] = true;  // the next check pass of this lane carries the test
            } else if (early || last) {  // convergence test + latch, one launch
                hipStream_t ts = lane[k];
                if (pw >= FT_WORDS && h->kn.fuse_test) {  // sharded accumulators / counters (fused_commit)
                    const FusedTest pf{st.hard + (size_t)ta * h->n, st.unsat + (size_t)ta * pw, st.done + ta, st.conv + ta,
                                       st.iters + (size_t)ta * TW, rem + it, h->n, pw, it, early ? 1 : 0};
                    hipLaunchKernelGGL(k_parity_fin_sharded, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), gs[k]),
                                       dim3(256), 0, ts, h->d_row_ptr, h->d_col_idx, h->m, st.synd + (size_t)ta * h->m, pf);
                } else
                    hipLaunchKernelGGL(k_parity_fin, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), gs[k]), dim3(256), 0,
                                       ts, h->d_row_ptr, h->d_col_idx, st.hard + (size_t)ta * h->n, h->m, h->n,
                                       st.synd + (size_t)ta * h->m, st.unsat + (size_t)ta * pw, pw, it, early ? 1 : 0, st.done + ta,
                                       st.conv + ta, st.iters + (size_t)ta * TW, rem + it);
                LAUNCH_CHECK();
            }
        }
        // a poll drains the queue (the GPU idles while the host turns around): skip the poll
        // points at which the call's earlier groups saw no codeword finish yet (`poll`, above)
        if (early && !last && poll && ps && it == defer_after && defer_after > 0 && 2 * it < max_iter && ps->streak >= 2 &&
            ps->since_poll < SPEC_PERIOD - 1) {
            ps->since_poll++;  // stop here unseen, as the last groups did
            *deferred = true;
            if (chain & 2) return 0;  // (the next group stops unseen too and continues these lanes: no host in between)
            return join();
        }
        if (early && !last && poll) {
            SC_TRY(join());
            SC_HIP(hipMemcpyAsync(h->h_remaining + it, rem + it, sizeof(int), hipMemcpyDeviceToHost, s));
            SC_HIP(hipStreamSynchronize(s));
            set_phase = true;
            const int rem = h->h_remaining[it];
            if (rem == 0) return 0;
            if (poll_hint && rem >= real_codewords) *poll_hint = std::max(*poll_hint, it + 1);
            // from `defer_after` on, any poll point may hand the stragglers over, provided the
            // restart (it iterations redone) is cheap next to what is still ahead
            // ... and the stragglers really get cheaper: fewer tiles, or few enough for the
            // row-parallel kernels
            const bool shrinks = (rem + TW - 1) / TW < g || rem <= el_limit(h, method);
            if (defer_after > 0 && it >= defer_after && 2 * it < max_iter && 2 * rem <= real_codewords && shrinks) {
                if (ps) {
                    ps->streak = (it == defer_after && 8 * rem <= real_codewords) ? ps->streak + 1 : 0;
                    ps->since_poll = 0;
                }
                *deferred = true;
                return 0;
            }
            if (ps && it >= defer_after) ps->streak = 0;
        }
    }
    if (chain & 2) return 0;  // (the next group continues these lanes; the last one of the chain joins)
    return join();
}

// el > 0: the group is ONE tile holding `el` codewords, decoded by the row-parallel kernels.
int iterate_group(scaldpc_bp *h, const TileState &st, int g0, int g, int max_iter, int method, float alpha, bool early,
                  int defer_after, hipStream_t s, bool *deferred, int el = 0, int real_codewords = 0,
                  PollState *poll_hint = nullptr, int chain = 0)
{
    *deferred = false;
    if (!el)
        return iterate_tiles(h, st, g0, g, max_iter, method, alpha, early, defer_after, s, deferred, real_codewords, poll_hint, chain);
    const int skip = early ? 1 : 0;
    const u64 *synd_g = st.synd + (size_t)g0 * h->m;
    u64 *hard_g = st.hard + (size_t)g0 * h->n;
    u64 *done_g = st.done + g0, *conv_g = st.conv + g0, *unsat_g = st.unsat + (size_t)g0 * parity_waves(h);
    int *iters_g = st.iters + (size_t)g0 * TW;
    float *post_g = st.post ? st.post + (size_t)g0 * h->n * TW : nullptr;
    if (early && h->kn.el_fuse)
        return iterate_el_early(h, el, max_iter, method, alpha, synd_g, hard_g, done_g, conv_g, unsat_g, iters_g, post_g, s);
    // row-parallel kernels, fixed iterations (or the four-launch early-exit form, SCALDPC_EL_FUSE=0)
    const bool fused = fused_init(h, method);
    if (h->E && !fused) {
        hipLaunchKernelGGL(k_el_init, dim3((unsigned)((h->E + 255) / 256), el), dim3(256), 0, s, h->d_col_idx, h->d_prior,
                           h->d_emsg, h->E);
        LAUNCH_CHECK();
    }
    if (early) SC_HIP(hipMemsetAsync(h->d_remaining, 0, sizeof(int) * ((size_t)max_iter + 2), s));
    for (int it = 1; it <= max_iter; it++) {
        const bool last = it == max_iter;
        SC_TRY(launch_el_check(h, method, alpha_for(alpha, it), el, synd_g, done_g, skip, s, fused && it == 1));
        SC_TRY(launch_el_var(h, el, post_g, hard_g, done_g, skip, (early || last) ? 1 : 0, s));
        if (early || last) {
            hipLaunchKernelGGL(k_parity<true>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), 1), dim3(256), 0, s,
                               h->d_row_ptr, h->d_col_idx, hard_g, h->m, h->n, const_cast<u64 *>(synd_g), unsat_g,
                               (const u64 *)done_g);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, it, early ? 1 : 0, done_g, conv_g, unsat_g, parity_waves(h),
                               iters_g, h->d_remaining + it);
            LAUNCH_CHECK();
        }
        if (early && !last && (it % 4 == 0 || it == 1)) {
            SC_HIP(hipMemcpyAsync(h->h_remaining + it, h->d_remaining + it, sizeof(int), hipMemcpyDeviceToHost, s));
            SC_HIP(hipStreamSynchronize(s));
            if (h->h_remaining[it] == 0) break;
        }
    }
    return 0;
}

// One compaction level: the `batch` codewords of `st` (T tiles), all iterations of one
// cache-resident tile group after the other.  In early-exit runs the stragglers of
// mostly-converged groups are gathered into dense tiles of their own and re-decoded from their
// inputs one level down (codewords are independent and BP is deterministic: identical
// results), and that level may shed its own stragglers again -- on the config-5 sweep the
// second level drops the codewords that needed 5-7 iterations and leaves the ~0.5 % that never
// converge to run their 100 iterations in 3 tiles instead of 11.
int decode_level(scaldpc_bp *h, int lvl, const TileState &st, int batch, int T, int G, int max_iter, int method,
                 float alpha, bool early, bool want_post, hipStream_t s)
{
    const int lim = el_limit(h, method);
    const int el = (T == 1 && batch <= lim) ? batch : 0;  // a handful of codewords: row-parallel kernels
    const int Gl = std::min(G, T);
    if (el)
        SC_TRY(ensure_el(h, el));
    else
        SC_TRY(ensure_msg(h, Gl, method));
    if (lvl == 0) {
        h->last_group = el ? 0 : Gl;
        h->last_early = early;
    }
    if (el) h->stat_el = el;
    h->stat_levels = lvl;

    int defer_after = 0;
    if (early && !el && lvl < scaldpc_bp::MAX_LEVELS) {
        defer_after = 4;  // measured on the config-5 sweep: 4-5 best (177k trials/s), 8: 156k, 12: 136k
        if (h->kn.compact_after >= 0) defer_after = h->kn.compact_after;
        if (max_iter <= 2 * defer_after) defer_after = 0;  // nothing to gain
    }
    if (!el)  // accumulators and block counters of the convergence tests: zero once, every launch leaves them zero
        SC_HIP(hipMemsetAsync(st.unsat, 0, sizeof(u64) * (size_t)T * parity_waves(h), s));
    std::vector<char> deferred_tile(T, 0);
    bool any = false;
    PollState poll_hint;
    // fixed-iteration runs chain consecutive groups of the same geometry lane by lane (iterate_tiles): no host interaction,
    // every pass of a lane depends only on that lane's previous pass over the same slice of the message array
    // (A/B on the HQC-128 bench, best of five runs each: 60.43 against 61.12 ms per step; tanh rule 94.5 against 95.0-95.4:
    // profiles/r04/ab_chain_groups.log)
    const bool lanes2 = !el && fused_init(h, method) && fixed_lanes(h, Gl) > 1;
    const bool chainable = !early && lanes2;
    // Early-exit runs chain too where the host stays out: a group that will stop at the hand-over point UNSEEN (two groups in
    // a row handed a small remainder over there, the real poll of every 16th group is not due, and no earlier poll point is
    // live) enqueues its launches without ever synchronising -- iterate_tiles decides exactly this from the PollState it is
    // handed, so it can be foretold here.
    auto unseen = [&](const PollState &p) {
        return early && lanes2 && defer_after > 0 && 2 * defer_after < max_iter && p.streak >= 2 && p.since_poll < SPEC_PERIOD - 1 && p.hint > 1 &&
               p.hint <= defer_after;
    };
    bool open = false;  // the previous group left its lanes un-joined
    for (int g0 = 0; g0 < T; g0 += Gl) {
        const int g = std::min(Gl, T - g0);
        const int real = std::min(batch - g0 * TW, g * TW);
        bool d = false;
        int chain = 0;
        const bool next_full = g0 + Gl < T && std::min(Gl, T - (g0 + Gl)) == Gl;
        if (chainable && g == Gl) {
            if (g0 > 0) chain |= 1;       // (the group before it was a full one too)
            if (next_full) chain |= 2;    // ... and so is the next
        } else if (g == Gl && unseen(poll_hint)) {
            PollState nxt = poll_hint;
            nxt.since_poll++;
            if (open) chain |= 1;
            // (the next group takes the next row of "still running" counters: no chaining across the wrap, which clears them all)
            if (next_full && unseen(nxt) && h->rem_slot + 2 < h->rem_rows) chain |= 2;
        }
        open = (chain & 2) != 0;
        SC_TRY(iterate_group(h, st, g0, g, max_iter, method, alpha, early, defer_after, s, &d, el, real, &poll_hint, chain));
        if (d) {
            any = true;
            for (int t = g0; t < g0 + g; t++) deferred_tile[t] = 1;
        }
    }
    if (!any) return 0;

    // ---- the stragglers, one level down --------------------------------------------------
    std::vector<u64> done_h(T);
    SC_HIP(hipMemcpyAsync(done_h.data(), st.done, sizeof(u64) * T, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    pvec<int> ids, slot_of((size_t)T * TW, -1);
    for (int t = 0; t < T; t++) {
        if (!deferred_tile[t]) continue;
        for (int c = 0; c < TW; c++) {
            const long b = (long)t * TW + c;
            if (b < batch && !((done_h[t] >> c) & 1)) {
                slot_of[b] = (int)ids.size();
                ids.push_back((int)b);
            }
        }
    }
    const int batch2 = (int)ids.size();
    if (batch2 == 0) return 0;
    if (lvl == 0) h->stat_deferred = batch2;
    const int T2 = (batch2 + TW - 1) / TW;
    ids.resize((size_t)T2 * TW, -1);
    scaldpc_bp::Level &L = h->lv[lvl + 1];
    if (T2 > L.cap_tiles) {
        dev_free(L.synd); dev_free(L.hard); dev_free(L.done); dev_free(L.conv); dev_free(L.unsat);
        dev_free(L.iters); dev_free(L.ids);
        L.cap_tiles = 0;
        SC_TRY(dev_alloc(&L.synd, (size_t)T2 * h->m));
        SC_TRY(dev_alloc(&L.hard, (size_t)T2 * h->n));
        SC_TRY(dev_alloc(&L.done, (size_t)T2));
        SC_TRY(dev_alloc(&L.conv, (size_t)T2));
        SC_TRY(dev_alloc(&L.unsat, (size_t)T2 * parity_waves(h)));
        SC_TRY(dev_alloc(&L.iters, (size_t)T2 * TW));
        SC_TRY(dev_alloc(&L.ids, (size_t)T2 * TW));
        L.cap_tiles = T2;
    }
    SC_TRY(grow(&L.slot_of, &L.cap_slot_of, (size_t)T * TW));
    if (want_post) SC_TRY(grow(&L.post, &L.cap_post, (size_t)T2 * h->n * TW));
    SC_HIP(hipMemcpyAsync(L.ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(L.slot_of, slot_of.data(), sizeof(int) * slot_of.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_planes, dim3((h->m + 63) / 64, T2), dim3(256), 0, s, st.synd, h->m, L.ids, L.synd);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_init_state, dim3(T2), dim3(64), 0, s, batch2, max_iter, L.done, L.conv, L.iters);
    LAUNCH_CHECK();
    SC_HIP(hipMemsetAsync(L.hard, 0, sizeof(u64) * (size_t)T2 * h->n, s));
    SC_HIP(hipStreamSynchronize(s));  // ids / slot_of are locals, and the level below reuses the stream
    const TileState st2{L.synd, L.hard, L.done, L.conv, L.unsat, L.iters, want_post ? L.post : nullptr};
    SC_TRY(decode_level(h, lvl + 1, st2, batch2, T2, G, max_iter, method, alpha, early, want_post, s));
    hipLaunchKernelGGL(k_scatter_planes, dim3((h->n + 63) / 64, T), dim3(256), 0, s, st.hard, h->n, L.slot_of, L.hard);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scatter_state, dim3(T), dim3(64), 0, s, st.iters, st.conv, L.slot_of, L.iters, L.conv);
    LAUNCH_CHECK();
    if (want_post) {
        hipLaunchKernelGGL(k_scatter_post, dim3(h->n, T), dim3(64), 0, s, st.post, h->n, L.slot_of, L.post);
        LAUNCH_CHECK();
    }
    return 0;
}

// The decode proper, on inputs already staged as planes (h->d_synd; h->d_recv when the
// caller wants e XOR v): state reset, then decode_level.
// Results stay on the device (h->d_hard / d_post / d_conv / d_iters).
int run_core(scaldpc_bp *h, int batch, int T, int G, int max_iter, int method, float alpha, bool early,
             bool want_post, hipStream_t s)
{
    if (!h->small_prepared) {
        hipLaunchKernelGGL(k_init_state, dim3(T), dim3(64), 0, s, batch, max_iter, h->d_done, h->d_conv, h->d_iters);
        LAUNCH_CHECK();
        SC_HIP(hipMemsetAsync(h->d_hard, 0, sizeof(u64) * (size_t)T * h->n, s));
    }
    h->last_group = 0;
    h->stat_deferred = 0;
    h->stat_el = 0;
    h->stat_levels = 0;

    // small graph: the LDS-resident single-launch decoder, plane I/O (Monte-Carlo entry points)
    {
        const size_t small_lds = (size_t)2 * h->E * sizeof(float) + (size_t)2 * h->n + h->m + 64;
        if (small_lds <= 60 * 1024 && h->E > 0 && h->kn.path != Knobs::STREAM && h->kn.path != Knobs::EDGE) {
#define SMALL_PLANES(M)                                                                                              \
    hipLaunchKernelGGL((k_bp_small<M, true>), dim3(batch), dim3(256), small_lds, s, h->d_row_ptr, h->d_col_idx,         \
                       h->d_col_ptr, h->d_csc_edge, h->d_prior, h->m, h->n, (int)h->E, (const void *)h->d_synd,         \
                       SCALDPC_IN_SYNDROME, max_iter, alpha, early ? 1 : 0, (void *)h->d_hard,                          \
                       want_post ? h->d_post : (float *)nullptr, h->d_iters, (void *)h->d_conv)
            if (method == SCALDPC_BP_MIN_SUM)
                SMALL_PLANES(SCALDPC_BP_MIN_SUM);
            else
                SMALL_PLANES(SCALDPC_BP_PRODUCT_SUM);
#undef SMALL_PLANES
            LAUNCH_CHECK();
            return 0;
        }
    }
    const TileState st{h->d_synd, h->d_hard, h->d_done, h->d_conv, h->d_unsat, h->d_iters,
                       want_post ? h->d_post : nullptr};
    return decode_level(h, 0, st, batch, T, G, max_iter, method, alpha, early, want_post, s);
}

}  // namespace

#include <chrono>
#define TMARK(name)                                                                                   \
    do {                                                                                              \
        static const bool on__ = getenv("SCALDPC_TIMING") != nullptr;                                 \
        if (on__) {                                                                                   \
            auto now__ = std::chrono::steady_clock::now();                                            \
            fprintf(stderr, "[create] %-10s %8.1f us\n", name,                                        \
                    std::chrono::duration<double, std::micro>(now__ - tmark_last()).count());         \
            tmark_last() = now__;                                                                     \
        }                                                                                             \
    } while (0)
static std::chrono::steady_clock::time_point &tmark_last()
{
    static thread_local std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    return t;
}

namespace {

// Everything that depends on the WHOLE graph, from the host CSR and the node degrees: degree
// buckets, CSC permutation, identity block, and the device copy of the arrays every path needs --
// views into ONE allocation filled by ONE copy: the attack loop builds a new decoder per decode
// (hqc.py:694), so construction is on its critical path (a dozen hipMalloc + synchronous hipMemcpy
// pairs cost more than the decode).  The per-kernel-family tables follow on first use
// (ensure_tile_tables / ensure_el_tables).  Used by scaldpc_bp_create and, on a handle whose graph
// has grown, by refresh_full (there CSR and priors already live in buffers of their own).
// Consumes cdeg.
int finish_graph(scaldpc_bp *h, const int *row_ptr, const int *col_idx, const pvec<int> &rdeg, pvec<int> &cdeg)
{
    const int m = h->m, n = h->n;
    const long nnz = h->E;
    h->max_row_deg = m ? *std::max_element(rdeg.begin(), rdeg.end()) : 0;
    h->min_row_deg = m ? *std::min_element(rdeg.begin(), rdeg.end()) : 0;
    h->max_col_deg = *std::max_element(cdeg.begin(), cdeg.end());
    static const int vb[] = {1, 2, 4, 8, 16, 32, 64};
    static const int rb[] = {2, 4, 8, 16, 32, 64};
    HostBuckets hv, hr;
    build_buckets(cdeg, vb, 7, true, hv);
    build_buckets(rdeg, rb, 6, false, hr);
    TMARK("buckets");
    h->var_bk = hv.bk;
    h->row_bk = hr.bk;
    h->need_scratch = hv.has_generic || hr.has_generic;
    h->identity_from = -1;
    if (n > m) {  // H = [Hin | I_m]?  (what hqc.decode builds, hqc.py:680)
        bool ident = true;
        for (int r = 0; r < m && ident; r++) {
            const int j = n - m + r;
            ident = cdeg[j] == 1 && row_ptr[r + 1] > row_ptr[r] && col_idx[row_ptr[r + 1] - 1] == j;
        }
        h->identity_from = ident ? n - m : -1;
    }
    h->el_ok = nnz > 0 && h->max_row_deg <= 64 && h->max_col_deg <= 64;  // the row-parallel path can take this graph
    const bool own_csr = h->incremental;  // CSR and priors already live in growable buffers of their own
    size_t total = 0;
    auto reserve = [&](size_t cnt) {  // 256-byte aligned sections
        const size_t off = total;
        total += (cnt + 63) / 64 * 64;
        return off;
    };
    const size_t o_row_ptr = own_csr ? 0 : reserve((size_t)m + 1), o_col_idx = own_csr ? 0 : reserve((size_t)nnz);
    const size_t o_col_ptr = reserve((size_t)n + 1), o_csc_edge = reserve((size_t)nnz), o_var_list = reserve(hv.list.size());
    const size_t o_prior = own_csr ? 0 : reserve((size_t)n);
    int *const host = stage_buffer(total);  // not cleared: every word a kernel reads is written below
    if (!host) return fail(SCALDPC_ENOMEM, "out of host memory");
    TMARK("reserve");
    if (!own_csr) {
        std::copy(row_ptr, row_ptr + m + 1, host + o_row_ptr);
        std::copy(col_idx, col_idx + nnz, host + o_col_idx);
    }
    // CSC permutation: edges grouped by column, ascending row (row-major scan keeps rows ascending)
    int *col_ptr = host + o_col_ptr, *csc_edge = host + o_csc_edge;
    {
        col_ptr[0] = 0;
        for (int j = 0; j < n; j++) col_ptr[j + 1] = col_ptr[j] + cdeg[j];
        pvec<int> cursor(col_ptr, col_ptr + n);
        for (int e = 0; e < (int)nnz; e++) csc_edge[cursor[col_idx[e]]++] = e;
    }
    std::copy(hv.list.begin(), hv.list.end(), host + o_var_list);
    TMARK("csc");
    SC_TRY(dev_alloc(&h->d_graph, total));
    if (hipMemcpy(h->d_graph, host, total * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(SCALDPC_EHIP, "graph upload failed");
    TMARK("upload");
    // what the lazy table builders need
    if (!own_csr) h->hg_row_ptr.assign(row_ptr, row_ptr + m + 1);
    h->hg_col_ptr.assign(col_ptr, col_ptr + n + 1);
    h->hg_csc_edge.assign(csc_edge, csc_edge + nnz);
    h->hg_cdeg.swap(cdeg);
    h->hg_var = std::move(hv);
    h->hg_row = std::move(hr);
    TMARK("keep");
    if (!own_csr) {
        h->d_row_ptr = h->d_graph + o_row_ptr;
        h->d_col_idx = h->d_graph + o_col_idx;
        h->d_prior = (float *)(h->d_graph + o_prior);
    }
    h->d_col_ptr = h->d_graph + o_col_ptr;
    h->d_csc_edge = h->d_graph + o_csc_edge;
    h->d_var_list = h->d_graph + o_var_list;
    return 0;
}

// ---- a graph that grows (scaldpc_bp_append_rows) -------------------------------------------------
// First append: CSR and priors leave the construction-time allocation for buffers with spare
// capacity; the host keeps the CSR (col_idx is recovered from the CSC mirror); the row-parallel
// tables are dropped and come back with free lanes on the next decode.
int make_incremental(scaldpc_bp *h)
{
    if (h->incremental) return 0;
    const size_t rows = (size_t)h->m + 1, edges = (size_t)h->E, cols = (size_t)h->n;
    h->cap_rows = rows + rows / 2 + 256;
    h->cap_edges = edges + edges / 2 + 4096;
    h->cap_cols = cols + cols / 2 + 256;
    SC_TRY(dev_alloc(&h->d_csr_rp, h->cap_rows));
    SC_TRY(dev_alloc(&h->d_csr_ci, h->cap_edges));
    SC_TRY(dev_alloc(&h->d_prior_buf, h->cap_cols));
    SC_HIP(hipMemcpy(h->d_csr_rp, h->d_row_ptr, rows * sizeof(int), hipMemcpyDeviceToDevice));
    if (edges) SC_HIP(hipMemcpy(h->d_csr_ci, h->d_col_idx, edges * sizeof(int), hipMemcpyDeviceToDevice));
    SC_HIP(hipMemcpy(h->d_prior_buf, h->d_prior, cols * sizeof(float), hipMemcpyDeviceToDevice));
    h->d_row_ptr = h->d_csr_rp;
    h->d_col_idx = h->d_csr_ci;
    h->d_prior = h->d_prior_buf;
    h->hg_col_idx.resize(edges);
    for (int j = 0; j < h->n; j++)
        for (int k = h->hg_col_ptr[j]; k < h->hg_col_ptr[j + 1]; k++) h->hg_col_idx[h->hg_csc_edge[k]] = j;
    h->incremental = true;
    dev_free(h->d_el_tab);  // rebuilt with free lanes per column on the next use
    h->d_el_slots = h->d_el_slot_col = nullptr;
    return 0;
}

template <typename T>
int grow_keep(T **p, size_t *cap, size_t used, size_t need)
{
    if (need <= *cap) return 0;
    const size_t ncap = need + need / 2 + 256;
    T *q = nullptr;
    SC_TRY(dev_alloc(&q, ncap));
    if (used) {
        const hipError_t e = hipMemcpy(q, *p, used * sizeof(T), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            dev_free(q);
            return fail(SCALDPC_EHIP, "copy into the grown buffer failed: %s", hipGetErrorString(e));
        }
    }
    dev_free(*p);
    *p = q;
    *cap = ncap;
    return 0;
}

// The tile / LDS kernels' view of a grown graph: CSC, degree buckets, identity block, device CSC
// arrays, and (lazily, on their first use) the tile tables.  O(E) on the host, as a construction.
int refresh_full(scaldpc_bp *h)
{
    if (!h->full_stale) return 0;
    pvec<int> rdeg(h->m), cdeg(h->hg_cdeg);
    for (int r = 0; r < h->m; r++) rdeg[r] = h->hg_row_ptr[r + 1] - h->hg_row_ptr[r];
    dev_free(h->d_graph);
    dev_free(h->d_tile_tab);
    h->d_var_meta = h->d_csc_list = h->d_row_list = nullptr;
    SC_TRY(finish_graph(h, h->hg_row_ptr.data(), h->hg_col_idx.data(), rdeg, cdeg));
    h->full_stale = false;
    return 0;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================

extern "C" {

int scaldpc_bp_create(int32_t m, int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx,
                      scaldpc_bp **out)
{
    if (!out) return fail(SCALDPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (m <= 0 || n <= 0 || nnz < 0 || !row_ptr || (nnz && !col_idx))
        return fail(SCALDPC_EINVAL, "bad graph arguments (m=%d n=%d nnz=%lld)", m, n, (long long)nnz);
    if (nnz > 0x7fffffffLL) return fail(SCALDPC_EINVAL, "nnz too large");
    if (row_ptr[0] != 0 || row_ptr[m] != nnz) return fail(SCALDPC_EINVAL, "row_ptr does not span [0, nnz]");
    TMARK("start");
    pvec<int> rdeg(m), cdeg(n, 0);
    for (int r = 0; r < m; r++) {
        if (row_ptr[r + 1] < row_ptr[r]) return fail(SCALDPC_EINVAL, "row_ptr not monotone at row %d", r);
        rdeg[r] = row_ptr[r + 1] - row_ptr[r];
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return fail(SCALDPC_EINVAL, "col_idx out of range at edge %d", e);
            if (e > row_ptr[r] && col_idx[e] <= col_idx[e - 1])
                return fail(SCALDPC_EINVAL, "col_idx not strictly ascending in row %d", r);
            cdeg[col_idx[e]]++;
        }
    }
    scaldpc_bp *h = new (std::nothrow) scaldpc_bp();
    if (!h) return fail(SCALDPC_ENOMEM, "out of host memory");
    knobs_from_env(h->kn);
    h->m = m;
    h->n = n;
    h->E = nnz;
    TMARK("validate");
    int rc = finish_graph(h, row_ptr, col_idx, rdeg, cdeg);
    if (!rc) rc = stream_acquire(&h->own_stream, &h->device);
    TMARK("stream");
    if (rc) {
        scaldpc_bp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int scaldpc_bp_set_channel_probs(scaldpc_bp *h, const double *probs)
{
    if (!h || !probs) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    pvec<float> llr(h->n);
    float last_p = 0.0f, last_llr = 0.0f;
    for (int j = 0; j < h->n; j++) {
        if (!(probs[j] >= 0.0 && probs[j] <= 1.0))
            return fail(SCALDPC_EINVAL, "channel_probs[%d] = %g is not a probability", j, probs[j]);
        // same expression, in fp32, as the oracle's f32 instantiation: log((1-p)/p)
        // (priors come in long runs of one value -- [w/N]*N ++ [1-certainty]*R, hqc.py:686-691 --
        // so the previous result is reused while p repeats)
        const float p = (float)probs[j];
        if (j == 0 || p != last_p) {
            last_p = p;
            last_llr = logf((1.0f - p) / p);
        }
        llr[j] = last_llr;
    }
    SC_HIP(hipMemcpy(h->d_prior, llr.data(), sizeof(float) * h->n, hipMemcpyHostToDevice));
    h->h_probs.assign(probs, probs + h->n);
    h->thr_valid = false;
    h->first_valid = false;
    h->have_prior = true;
    h->prior_n = h->n;
    return 0;
}

int scaldpc_bp_set_channel_probs_tail(scaldpc_bp *h, int32_t first, int32_t count, const double *probs)
{
    if (!h || (count > 0 && !probs)) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    if (first < 0 || count < 0 || (long)first + count > h->n)
        return fail(SCALDPC_EINVAL, "columns [%d, %d) are outside the graph (n = %d)", first, first + count, h->n);
    if (first > h->prior_n) return fail(SCALDPC_EINVAL, "priors of columns [%d, %d) are still unset", h->prior_n, first);
    if (count == 0) return 0;
    if (h->async_used) SC_HIP(hipDeviceSynchronize());
    pvec<float> llr(count);
    float last_p = 0.0f, last_llr = 0.0f;
    for (int j = 0; j < count; j++) {
        if (!(probs[j] >= 0.0 && probs[j] <= 1.0))
            return fail(SCALDPC_EINVAL, "channel_probs[%d] = %g is not a probability", first + j, probs[j]);
        const float p = (float)probs[j];  // same expression as scaldpc_bp_set_channel_probs
        if (j == 0 || p != last_p) {
            last_p = p;
            last_llr = logf((1.0f - p) / p);
        }
        llr[j] = last_llr;
    }
    SC_HIP(hipMemcpy(h->d_prior + first, llr.data(), sizeof(float) * count, hipMemcpyHostToDevice));
    h->h_probs.resize(h->n, 0.0);
    std::copy(probs, probs + count, h->h_probs.begin() + first);
    h->thr_valid = false;
    h->first_valid = false;
    h->prior_n = std::max(h->prior_n, first + count);
    h->have_prior = true;
    return 0;
}

int scaldpc_bp_append_rows(scaldpc_bp *h, int32_t nrows, const int32_t *row_ptr, const int32_t *col_idx, int32_t new_n)
{
    if (!h || nrows < 0 || (nrows > 0 && !row_ptr)) return fail(SCALDPC_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    CacheBypass guard(h->async_used);
    if (new_n < h->n) return fail(SCALDPC_EINVAL, "new_n = %d is smaller than the current block length %d", new_n, h->n);
    const long add = nrows ? row_ptr[nrows] : 0;
    if (nrows && (row_ptr[0] != 0 || add < 0 || (add && !col_idx))) return fail(SCALDPC_EINVAL, "row_ptr does not start at 0");
    if ((long)h->E + add > 0x7fffffffL) return fail(SCALDPC_EINVAL, "nnz too large");
    // only the NEW rows are validated; the old ones were when they came in
    for (int r = 0; r < nrows; r++) {
        if (row_ptr[r + 1] < row_ptr[r]) return fail(SCALDPC_EINVAL, "row_ptr not monotone at appended row %d", r);
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            if (col_idx[e] < 0 || col_idx[e] >= new_n) return fail(SCALDPC_EINVAL, "col_idx out of range at appended edge %d", e);
            if (e > row_ptr[r] && col_idx[e] <= col_idx[e - 1])
                return fail(SCALDPC_EINVAL, "col_idx not strictly ascending in appended row %d", r);
        }
    }
    if (nrows == 0 && new_n == h->n) return 0;
    if (h->async_used) SC_HIP(hipDeviceSynchronize());  // the tables below may still be read by work in flight
    TMARK("app:start");
    // From here on the handle is being rebuilt in place.  A failure below (an allocation, a copy) leaves host mirrors
    // and device arrays in disagreement: the handle is then marked broken and refuses every later call.
    struct Unfinished {
        scaldpc_bp *h;
        bool done = false;
        ~Unfinished() { if (!done) h->broken = true; }
    } unfinished{h};
    SC_TRY(make_incremental(h));
    hipStream_t s = h->own_stream;
    const int m0 = h->m, n0 = h->n;
    const long E0 = h->E;
    // (each view is re-pointed right after its own buffer moved: a later failure must not leave it dangling)
    SC_TRY(grow_keep(&h->d_csr_rp, &h->cap_rows, (size_t)m0 + 1, (size_t)m0 + 1 + nrows));
    h->d_row_ptr = h->d_csr_rp;
    SC_TRY(grow_keep(&h->d_csr_ci, &h->cap_edges, (size_t)E0, (size_t)E0 + add));
    h->d_col_idx = h->d_csr_ci;
    SC_TRY(grow_keep(&h->d_prior_buf, &h->cap_cols, (size_t)n0, (size_t)new_n));
    h->d_prior = h->d_prior_buf;

    // ---- host mirror: CSR, column degrees ------------------------------------------------------
    h->hg_row_ptr.reserve((size_t)m0 + 1 + nrows);
    for (int r = 0; r < nrows; r++) h->hg_row_ptr.push_back((int)(E0 + row_ptr[r + 1]));
    h->hg_col_idx.insert(h->hg_col_idx.end(), col_idx, col_idx + add);
    h->hg_cdeg.resize(new_n, 0);
    for (int r = 0; r < nrows; r++) h->max_row_deg = std::max(h->max_row_deg, row_ptr[r + 1] - row_ptr[r]);

    TMARK("app:mirror");
    // ---- row-parallel tables, in place: one slot word per new edge while its column has a free lane
    const bool had_tables = h->d_el_tab != nullptr;
    pvec<int> &dirty_slot = h->el_dirty_slot, &dirty_col = h->el_dirty_col;  // words of h_el_slots / h_el_col that changed
    dirty_slot.clear();
    dirty_col.clear();
    bool el_alive = had_tables;
    if (had_tables) {
        h->seg_slot.resize(new_n, -1);
        h->seg_cap.resize(new_n, 0);
    }
    // A word may be listed more than once (filled in place, then moved with its column in the same call):
    // the pairs sent to the device carry the mirror's FINAL value of the word, so duplicates are harmless
    // in whatever order the scatter kernel applies them.
    auto set_slot = [&](size_t i, int v) {
        h->h_el_slots[i] = v;
        dirty_slot.push_back((int)i);
    };
    auto new_segment = [&](int c, int cap, const int *edges, int d) {  // writes a whole segment
        const int s0 = el_alloc_segment(h, cap), start = s0 & 63;
        for (int k = 0; k < cap; k++) {
            set_slot(2 * (size_t)(s0 + k), k < d ? edges[k] : -1);
            set_slot(2 * (size_t)(s0 + k) + 1, el_tag(start, k, cap, true));
        }
        h->h_el_col[s0] = c;
        dirty_col.push_back(s0);  // (a segment start is handed out once: no duplicates)
        h->seg_slot[c] = s0;
        h->seg_cap[c] = (unsigned char)cap;
    };
    int edges_tmp[64];
    for (long i = 0; i < add; i++) {
        const int c = col_idx[i], e = (int)(E0 + i);
        const int d_old = h->hg_cdeg[c]++;
        h->max_col_deg = std::max(h->max_col_deg, d_old + 1);
        if (!el_alive) continue;
        if (d_old + 1 > 64) {  // a column wider than a wave: the row-parallel path is gone for this graph
            el_alive = false;
            continue;
        }
        if (h->seg_slot[c] < 0) {  // a column that came with these rows
            edges_tmp[0] = e;
            new_segment(c, std::min(64, 1 + el_slack(h, c, 1)), edges_tmp, 1);
        } else if (d_old < h->seg_cap[c]) {
            set_slot(2 * (size_t)(h->seg_slot[c] + d_old), e);
        } else {  // the segment is full: the column moves to a larger one, its old lanes go dead
            const int s_old = h->seg_slot[c];
            for (int k = 0; k < d_old; k++) edges_tmp[k] = h->h_el_slots[2 * (size_t)(s_old + k)];
            edges_tmp[d_old] = e;
            for (int k = 0; k < d_old; k++) set_slot(2 * (size_t)(s_old + k), -1);  // (the free lanes hold -1 already)
            set_slot(2 * (size_t)s_old + 1, 0);  // no head, no column: the other lanes of the old segment just idle
            new_segment(c, std::min(64, d_old + 1 + std::max(el_slack(h, c, d_old + 1), (d_old + 1) / 2)), edges_tmp, d_old + 1);
        }
    }
    if (el_alive)
        for (int c = n0; c < new_n; c++)  // new columns no appended row touches: isolated variables still have a posterior
            if (h->seg_slot[c] < 0) new_segment(c, 1, edges_tmp, 0);

    h->m = m0 + nrows;
    h->n = new_n;
    h->E = E0 + add;
    h->el_ok = h->E > 0 && h->max_row_deg <= 64 && h->max_col_deg <= 64;
    if (had_tables && (!el_alive || !h->el_ok)) {
        dev_free(h->d_el_tab);
        h->d_el_slots = h->d_el_slot_col = nullptr;
        el_alive = false;
    }

    TMARK("app:tables");
    // ---- device: CSR tails, table words -----------------------------------------------------------
    const bool reupload = el_alive && h->el_waves > h->el_cap_bins;  // the bins outgrew the device table
    const size_t npairs = (el_alive && !reupload) ? dirty_slot.size() + dirty_col.size() : 0;
    const size_t stage_ints = (size_t)nrows + add + 2 * npairs;
    if (stage_ints > h->cap_pairs) {
        cached_free(h->h_pairs);
        dev_free(h->d_pairs);
        h->h_pairs = nullptr;
        h->cap_pairs = 0;
        const size_t want = stage_ints + stage_ints / 2 + 1024;
        SC_TRY(cached_alloc((void **)&h->h_pairs, want * sizeof(int), true));
        SC_TRY(dev_alloc(&h->d_pairs, want));
        h->cap_pairs = want;
    }
    int *st = h->h_pairs;
    for (int r = 0; r < nrows; r++) st[r] = (int)(E0 + row_ptr[r + 1]);
    std::copy(col_idx, col_idx + add, st + nrows);
    if (nrows) SC_HIP(hipMemcpyAsync(h->d_row_ptr + m0 + 1, st, sizeof(int) * nrows, hipMemcpyHostToDevice, s));
    if (add) SC_HIP(hipMemcpyAsync(h->d_col_idx + E0, st + nrows, sizeof(int) * add, hipMemcpyHostToDevice, s));
    if (reupload) {
        SC_TRY(el_upload(h));
    } else if (npairs) {
        int *pp = st + nrows + add;
        const int col_base = (int)el_col_off(h);
        size_t q = 0;
        for (int i : dirty_slot) {
            pp[2 * q] = i;
            pp[2 * q + 1] = h->h_el_slots[i];
            q++;
        }
        for (int i : dirty_col) {
            pp[2 * q] = col_base + i;
            pp[2 * q + 1] = h->h_el_col[i];
            q++;
        }
        SC_HIP(hipMemcpyAsync(h->d_pairs, pp, sizeof(int) * 2 * npairs, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_apply_pairs, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, h->d_el_tab,
                           (const int2 *)h->d_pairs, (int)npairs);
        LAUNCH_CHECK();
    }
    SC_HIP(hipStreamSynchronize(s));
    TMARK("app:device");

    // ---- what was sized or derived for the old graph ------------------------------------------------
    h->full_stale = true;  // CSC, degree buckets, identity block, tile tables: rebuilt when the tile / LDS kernels are next needed
    h->h_probs.resize(new_n, 0.0);
    h->thr_valid = false;
    h->first_valid = false;
    dev_free(h->d_thr);
    dev_free(h->d_msg);
    dev_free(h->d_scratch);
    h->cap_group = 0;
    dev_free(h->d_rec);
    dev_free(h->d_mask);
    h->cap_rec_group = 0;
    for (auto &L : h->lv) {
        dev_free(L.synd); dev_free(L.hard); dev_free(L.done); dev_free(L.conv); dev_free(L.unsat);
        dev_free(L.iters); dev_free(L.ids); dev_free(L.post);
        L.cap_tiles = 0;
        L.cap_post = 0;
    }
    h->last_group = 0;
    TMARK("app:tail");
    unfinished.done = true;
    return 0;
}

int scaldpc_bp_last_compacted(scaldpc_bp *h, int64_t *count)
{
    if (!h || !count) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *count = h->stat_deferred;
    return 0;
}

int scaldpc_bp_last_stats(scaldpc_bp *h, int64_t *out)
{
    if (!h || !out) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    out[0] = h->stat_deferred;
    out[1] = h->stat_el;
    out[2] = h->stat_levels;
    out[3] = 0;
    return 0;
}

int scaldpc_bp_set_tile_group(scaldpc_bp *h, int32_t tiles)
{
    if (!h || tiles < 0) return fail(SCALDPC_EINVAL, "bad tile group");
    std::lock_guard<std::mutex> lk(h->mu);
    h->tile_group = tiles;
    return 0;
}

int scaldpc_bp_decode_batch(scaldpc_bp *h, const uint8_t *in, int32_t input_kind, int32_t batch, int32_t max_iter,
                            int32_t method, float alpha, uint32_t flags, void *stream, uint8_t *out_bits,
                            float *out_llr, int32_t *out_iters, uint8_t *out_conv)
{
    if (!h || !in || !out_bits) return fail(SCALDPC_EINVAL, "NULL argument");
    if (batch <= 0) return fail(SCALDPC_EINVAL, "batch must be positive (got %d)", batch);
    if (input_kind != SCALDPC_IN_SYNDROME && input_kind != SCALDPC_IN_RECEIVED)
        return fail(SCALDPC_EINVAL, "unknown input kind %d", input_kind);
    if (method != SCALDPC_BP_PRODUCT_SUM && method != SCALDPC_BP_MIN_SUM)
        return fail(SCALDPC_EINVAL, "unknown bp method %d", method);
    if (!(alpha >= 0.0f)) return fail(SCALDPC_EINVAL, "ms_scaling_factor must be >= 0");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    if (!h->have_prior || h->prior_n < h->n)
        return fail(SCALDPC_EINVAL, "channel probabilities not set (columns [%d, %d))", h->have_prior ? h->prior_n : 0, h->n);
    if (max_iter <= 0) max_iter = h->n;
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO;
    const bool early = flags & SCALDPC_F_EARLY_EXIT;
    if ((flags & SCALDPC_F_ASYNC) && (!dev_io || early))
        return fail(SCALDPC_EINVAL, "SCALDPC_F_ASYNC needs DEVICE_IO and no EARLY_EXIT (early exit polls the device)");
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    if (flags & SCALDPC_F_ASYNC) h->async_used = true;
    CacheBypass guard(h->async_used);

    const int T = (batch + TW - 1) / TW;
    if (T > 65535) return fail(SCALDPC_EINVAL, "batch %d too large for one call (max %d)", batch, 65535 * TW);
    const int G = (h->tile_group > 0) ? std::min(h->tile_group, T) : auto_group(h, T);
    const int len = input_kind == SCALDPC_IN_SYNDROME ? h->m : h->n;

    // Small graph: the LDS-resident single-launch decoder (k_bp_small).
    const size_t small_lds = (size_t)2 * h->E * sizeof(float) + (size_t)2 * h->n + h->m + 64;
    const bool small_fits = small_lds <= 60 * 1024 && h->E > 0;
    if (h->full_stale) {  // rows were appended: only the row-parallel path is up to date
        const bool small_path = small_fits && h->kn.path != Knobs::STREAM && h->kn.path != Knobs::EDGE;
        if (small_path || !(T == 1 && batch <= el_limit(h, method))) SC_TRY(refresh_full(h));
    }
    if (h->kn.path == Knobs::LDS && !small_fits)  // "stream" / "edge" / "lds" pin a path (tests)
        return fail(SCALDPC_EINVAL, "SCALDPC_PATH=lds but the graph needs %zu B of LDS", small_lds);
    if (small_fits && h->kn.path != Knobs::STREAM && h->kn.path != Knobs::EDGE) {
        const uint8_t *din = in;
        uint8_t *dbits = out_bits, *dconv = out_conv;
        float *dllr = out_llr;
        int *diters = out_iters;
        if (!dev_io) {
            SC_TRY(grow(&h->d_in, &h->cap_in, (size_t)batch * len));
            SC_HIP(hipMemcpyAsync(h->d_in, in, (size_t)batch * len, hipMemcpyHostToDevice, s));
            din = h->d_in;
            SC_TRY(grow(&h->d_out_bits, &h->cap_out_bits, (size_t)batch * h->n));
            dbits = h->d_out_bits;
            if (out_llr) {
                SC_TRY(grow(&h->d_out_llr, &h->cap_out_llr, (size_t)batch * h->n));
                dllr = h->d_out_llr;
            }
            if ((size_t)batch > h->cap_out_b) {
                dev_free(h->d_out_iters);
                dev_free(h->d_out_conv);
                h->cap_out_b = 0;
                SC_TRY(dev_alloc(&h->d_out_iters, (size_t)batch));
                SC_TRY(dev_alloc(&h->d_out_conv, (size_t)batch));
                h->cap_out_b = batch;
            }
            diters = out_iters ? h->d_out_iters : nullptr;
            dconv = out_conv ? h->d_out_conv : nullptr;
        }
#define SMALL_LAUNCH(M)                                                                                            \
    hipLaunchKernelGGL((k_bp_small<M, false>), dim3(batch), dim3(256), small_lds, s, h->d_row_ptr, h->d_col_idx, h->d_col_ptr, \
                       h->d_csc_edge, h->d_prior, h->m, h->n, (int)h->E, din, input_kind, max_iter, alpha,          \
                       early ? 1 : 0, dbits, dllr, diters, dconv)
        if (method == SCALDPC_BP_MIN_SUM)
            SMALL_LAUNCH(SCALDPC_BP_MIN_SUM);
        else
            SMALL_LAUNCH(SCALDPC_BP_PRODUCT_SUM);
#undef SMALL_LAUNCH
        LAUNCH_CHECK();
        if (!dev_io) {
            SC_HIP(hipMemcpyAsync(out_bits, dbits, (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
            if (out_llr) SC_HIP(hipMemcpyAsync(out_llr, dllr, sizeof(float) * (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
            if (out_iters) SC_HIP(hipMemcpyAsync(out_iters, diters, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, s));
            if (out_conv) SC_HIP(hipMemcpyAsync(out_conv, dconv, (size_t)batch, hipMemcpyDeviceToHost, s));
        }
        h->stat_deferred = 0;
        if (!(flags & SCALDPC_F_ASYNC)) SC_HIP(hipStreamSynchronize(s));
        return 0;
    }
    SC_TRY(ensure_workspace(h, T, G, out_llr != nullptr, max_iter));

    // ---- a handful of codewords from host buffers: fused reshaping, one copy each way ------------------
    if (!dev_io && T == 1 && batch <= 8 && max_iter <= 1024) {
        const size_t bits_bytes = ((size_t)batch * h->n + 3) / 4 * 4;
        const size_t out_bytes = bits_bytes + (out_llr ? sizeof(float) * (size_t)batch * h->n : 0) + sizeof(int) * batch + batch;
        const size_t in_bytes = (size_t)batch * len, io_bytes = std::max(in_bytes, out_bytes);
        if (io_bytes > h->cap_h_io) {
            cached_free(h->h_io);
            h->h_io = nullptr;
            h->cap_h_io = 0;
            SC_TRY(cached_alloc((void **)&h->h_io, io_bytes + io_bytes / 2, true));
            h->cap_h_io = io_bytes + io_bytes / 2;
        }
        SC_TRY(grow(&h->d_in, &h->cap_in, in_bytes));
        SC_TRY(grow(&h->d_out_all, &h->cap_out_all, out_bytes));
        SC_TRY(ensure_el_unsat(h, max_iter));
        memcpy(h->h_io, in, in_bytes);
        SC_HIP(hipMemcpyAsync(h->d_in, h->h_io, in_bytes, hipMemcpyHostToDevice, s));
        const bool recvd = input_kind == SCALDPC_IN_RECEIVED;
        hipLaunchKernelGGL(k_small_prepare, dim3((std::max(len, h->n) + 63) / 64), dim3(256), 0, s, h->d_in, len, batch,
                           recvd ? h->d_recv : h->d_synd, h->n, h->d_hard, max_iter, h->d_done, h->d_conv, h->d_iters,
                           h->d_remaining, max_iter + 2, h->d_el_unsat, (max_iter + 2) * TW);
        LAUNCH_CHECK();
        if (recvd) {
            hipLaunchKernelGGL(k_parity<false>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), 1), dim3(256), 0, s,
                               h->d_row_ptr, h->d_col_idx, h->d_recv, h->m, h->n, h->d_synd, (u64 *)nullptr, (const u64 *)nullptr);
            LAUNCH_CHECK();
        }
        h->small_prepared = true;
        const int rc = run_core(h, batch, T, G, max_iter, method, alpha, early, out_llr != nullptr, s);
        h->small_prepared = false;
        SC_TRY(rc);
        hipLaunchKernelGGL(k_small_unpack, dim3((h->n + 255) / 256), dim3(256), 0, s, h->d_hard, recvd ? h->d_recv : (const u64 *)nullptr,
                           out_llr ? h->d_post : (const float *)nullptr, h->d_conv, h->d_iters, h->n, batch, h->d_out_all);
        LAUNCH_CHECK();
        SC_HIP(hipMemcpyAsync(h->h_io, h->d_out_all, out_bytes, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));
        const uint8_t *o = h->h_io;
        memcpy(out_bits, o, (size_t)batch * h->n);
        o += bits_bytes;
        if (out_llr) {
            memcpy(out_llr, o, sizeof(float) * (size_t)batch * h->n);
            o += sizeof(float) * (size_t)batch * h->n;
        }
        if (out_iters) memcpy(out_iters, o, sizeof(int) * batch);
        if (out_conv) memcpy(out_conv, o + sizeof(int) * batch, batch);
        return 0;
    }

    // ---- stage input --------------------------------------------------------
    const uint8_t *din = in;
    if (!dev_io) {
        SC_TRY(grow(&h->d_in, &h->cap_in, (size_t)batch * len));
        SC_HIP(hipMemcpyAsync(h->d_in, in, (size_t)batch * len, hipMemcpyHostToDevice, s));
        din = h->d_in;
    }
    if (input_kind == SCALDPC_IN_SYNDROME) {
        hipLaunchKernelGGL(k_pack_bits, dim3((h->m + 63) / 64, T), dim3(256), 0, s, din, h->m, batch, h->d_synd);
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_pack_bits, dim3((h->n + 63) / 64, T), dim3(256), 0, s, din, h->n, batch, h->d_recv);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_parity<false>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), T), dim3(256), 0, s, h->d_row_ptr, h->d_col_idx,
                           h->d_recv, h->m, h->n, h->d_synd, (u64 *)nullptr, (const u64 *)nullptr);
        LAUNCH_CHECK();
    }
    SC_TRY(run_core(h, batch, T, G, max_iter, method, alpha, early, out_llr != nullptr, s));

    // ---- outputs --------------------------------------------------------------
    uint8_t *dbits = out_bits;
    float *dllr = out_llr;
    int *diters = out_iters;
    uint8_t *dconv = out_conv;
    if (!dev_io) {
        SC_TRY(grow(&h->d_out_bits, &h->cap_out_bits, (size_t)batch * h->n));
        dbits = h->d_out_bits;
        if (out_llr) {
            SC_TRY(grow(&h->d_out_llr, &h->cap_out_llr, (size_t)batch * h->n));
            dllr = h->d_out_llr;
        }
        if ((size_t)batch > h->cap_out_b) {
            dev_free(h->d_out_iters);
            dev_free(h->d_out_conv);
            h->cap_out_b = 0;
            SC_TRY(dev_alloc(&h->d_out_iters, (size_t)batch));
            SC_TRY(dev_alloc(&h->d_out_conv, (size_t)batch));
            h->cap_out_b = batch;
        }
        diters = out_iters ? h->d_out_iters : nullptr;
        dconv = out_conv ? h->d_out_conv : nullptr;
    }
    hipLaunchKernelGGL(k_unpack_bits, dim3((h->n + 255) / 256, T), dim3(256), 0, s, h->d_hard,
                       input_kind == SCALDPC_IN_RECEIVED ? h->d_recv : (const u64 *)nullptr, h->n, batch, dbits);
    LAUNCH_CHECK();
    if (out_llr) {
        hipLaunchKernelGGL(k_unpack_llr, dim3((h->n + 63) / 64, T), dim3(256), 0, s, h->d_post, h->n, batch, dllr);
        LAUNCH_CHECK();
    }
    if (diters || dconv) {
        hipLaunchKernelGGL(k_unpack_state, dim3(T), dim3(64), 0, s, h->d_conv, h->d_iters, batch, diters, dconv);
        LAUNCH_CHECK();
    }
    if (!dev_io) {
        SC_HIP(hipMemcpyAsync(out_bits, dbits, (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
        if (out_llr) SC_HIP(hipMemcpyAsync(out_llr, dllr, sizeof(float) * (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
        if (out_iters) SC_HIP(hipMemcpyAsync(out_iters, diters, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, s));
        if (out_conv) SC_HIP(hipMemcpyAsync(out_conv, dconv, (size_t)batch, hipMemcpyDeviceToHost, s));
    }
    if (!(flags & SCALDPC_F_ASYNC)) SC_HIP(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------
// Monte-Carlo entry points (K6)
// ---------------------------------------------------------------------------
namespace {

u64 bernoulli_threshold(double p) { return p >= 1.0 ? (1ull << 32) : p <= 0.0 ? 0ull : (u64)(p * 4294967296.0); }

int mc_common_args(scaldpc_bp *h, int32_t batch, int32_t method, float alpha, uint8_t *out_success)
{
    if (!h || !out_success) return fail(SCALDPC_EINVAL, "NULL argument");
    if (batch <= 0) return fail(SCALDPC_EINVAL, "batch must be positive (got %d)", batch);
    if (method != SCALDPC_BP_PRODUCT_SUM && method != SCALDPC_BP_MIN_SUM)
        return fail(SCALDPC_EINVAL, "unknown bp method %d", method);
    if (!(alpha >= 0.0f)) return fail(SCALDPC_EINVAL, "ms_scaling_factor must be >= 0");
    return 0;
}

int mc_finish(scaldpc_bp *h, int batch, int T, int nv, bool dev_io, hipStream_t s, uint8_t *out_success,
              int32_t *out_iters)
{
    SC_TRY(grow(&h->d_diff, &h->cap_diff, (size_t)T));
    SC_HIP(hipMemsetAsync(h->d_diff, 0, sizeof(u64) * (size_t)T, s));
    hipLaunchKernelGGL(k_mc_compare, dim3((nv + 255) / 256, T), dim3(256), 0, s, h->d_hard, h->d_mc, h->n, nv,
                       h->d_diff);
    LAUNCH_CHECK();
    uint8_t *ds = out_success;
    int *di = out_iters;
    if (!dev_io) {
        SC_TRY(grow(&h->d_succ, &h->cap_succ, (size_t)batch));
        ds = h->d_succ;
        if (out_iters) {
            if ((size_t)batch > h->cap_out_b) {
                dev_free(h->d_out_iters); dev_free(h->d_out_conv);
                h->cap_out_b = 0;
                SC_TRY(dev_alloc(&h->d_out_iters, (size_t)batch));
                SC_TRY(dev_alloc(&h->d_out_conv, (size_t)batch));
                h->cap_out_b = batch;
            }
            di = h->d_out_iters;
        }
    }
    hipLaunchKernelGGL(k_mc_result, dim3(T), dim3(64), 0, s, h->d_diff, h->d_iters, batch, ds, di);
    LAUNCH_CHECK();
    if (!dev_io) {
        SC_HIP(hipMemcpyAsync(out_success, ds, (size_t)batch, hipMemcpyDeviceToHost, s));
        if (out_iters) SC_HIP(hipMemcpyAsync(out_iters, di, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, s));
    }
    SC_HIP(hipStreamSynchronize(s));
    return 0;
}

int mc_export_bits(scaldpc_bp *h, const u64 *planes, int batch, int T, bool dev_io, hipStream_t s, uint8_t *out)
{
    uint8_t *d = out;
    if (!dev_io) {
        SC_TRY(grow(&h->d_out_bits, &h->cap_out_bits, (size_t)batch * h->n));
        d = h->d_out_bits;
    }
    hipLaunchKernelGGL(k_unpack_bits, dim3((h->n + 255) / 256, T), dim3(256), 0, s, planes, (const u64 *)nullptr, h->n,
                       batch, d);
    LAUNCH_CHECK();
    if (!dev_io) {
        SC_HIP(hipMemcpyAsync(out, d, (size_t)batch * h->n, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));  // d_out_bits is reused below
    }
    return 0;
}

}  // namespace

int scaldpc_mc_fer_run(scaldpc_bp *h, int64_t first_trial, int32_t batch, uint64_t seed, int32_t max_iter,
                       int32_t method, float alpha, uint32_t flags, void *stream, uint8_t *out_success,
                       int32_t *out_iters, uint8_t *out_error)
{
    SC_TRY(mc_common_args(h, batch, method, alpha, out_success));
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    CacheBypass guard(h->async_used);
    if (!h->have_prior || h->prior_n < h->n)
        return fail(SCALDPC_EINVAL, "channel probabilities not set (columns [%d, %d))", h->have_prior ? h->prior_n : 0, h->n);
    if (max_iter <= 0) max_iter = h->n;
    SC_TRY(refresh_full(h));
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO, early = flags & SCALDPC_F_EARLY_EXIT;
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    const int T = (batch + TW - 1) / TW;
    if (T > 65535) return fail(SCALDPC_EINVAL, "batch %d too large for one call", batch);
    const int G = (h->tile_group > 0) ? std::min(h->tile_group, T) : auto_group(h, T);
    SC_TRY(ensure_workspace(h, T, G, false, max_iter));
    SC_TRY(grow(&h->d_mc, &h->cap_mc, (size_t)T * h->n));
    if (!h->thr_valid) {
        if (!h->d_thr) SC_TRY(dev_alloc(&h->d_thr, (size_t)h->n));
        pvec<u64> thr(h->n);
        for (int j = 0; j < h->n; j++) thr[j] = bernoulli_threshold(h->h_probs[j]);
        SC_HIP(hipMemcpy(h->d_thr, thr.data(), sizeof(u64) * h->n, hipMemcpyHostToDevice));
        h->thr_valid = true;
    }
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    // error ~ Bernoulli(p_i) per position (decode.py:166-167), syndrome = H error (decode.py:168)
    hipLaunchKernelGGL(k_mc_bernoulli<false>, dim3((h->n + 63) / 64, T), dim3(256), 0, s, h->d_mc, h->n, batch,
                       (long)first_trial, 0u, k0, k1, (const u64 *)h->d_thr, 0ull);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_parity<false>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), T), dim3(256), 0, s, h->d_row_ptr, h->d_col_idx,
                       h->d_mc, h->m, h->n, h->d_synd, (u64 *)nullptr, (const u64 *)nullptr);
    LAUNCH_CHECK();
    if (out_error) SC_TRY(mc_export_bits(h, h->d_mc, batch, T, dev_io, s, out_error));
    SC_TRY(run_core(h, batch, T, G, max_iter, method, alpha, early, false, s));
    // success iff decoding == error everywhere (decode.py:173-175)
    return mc_finish(h, batch, T, h->n, dev_io, s, out_success, out_iters);
}

int scaldpc_mc_hqc_run(scaldpc_bp *h, int32_t omega, double eps, int64_t first_trial, int32_t batch, uint64_t seed,
                       int32_t max_iter, int32_t method, float alpha, uint32_t flags, void *stream,
                       uint8_t *out_success, int32_t *out_iters, uint8_t *out_msg, int32_t *out_y)
{
    SC_TRY(mc_common_args(h, batch, method, alpha, out_success));
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    CacheBypass guard(h->async_used);
    if (!h->have_prior || h->prior_n < h->n)
        return fail(SCALDPC_EINVAL, "channel probabilities not set (columns [%d, %d))", h->have_prior ? h->prior_n : 0, h->n);
    SC_TRY(refresh_full(h));
    if (h->identity_from < 0) return fail(SCALDPC_EINVAL, "parity-check matrix is not of the form [Hin | I] (hqc.py:680)");
    const int N = h->identity_from;
    if (omega < 0 || omega > N) return fail(SCALDPC_EINVAL, "omega must be in [0, N]");
    if (!(eps >= 0.0 && eps <= 1.0)) return fail(SCALDPC_EINVAL, "eps must be a probability");
    if ((size_t)omega * 64 * 4 > 64 * 1024) return fail(SCALDPC_EDEGREE, "omega %d too large for the LDS-resident sampler", omega);
    if (max_iter <= 0) max_iter = h->n;
    const bool dev_io = flags & SCALDPC_F_DEVICE_IO, early = flags & SCALDPC_F_EARLY_EXIT;
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    const int T = (batch + TW - 1) / TW;
    if (T > 65535) return fail(SCALDPC_EINVAL, "batch %d too large for one call", batch);
    const int G = (h->tile_group > 0) ? std::min(h->tile_group, T) : auto_group(h, T);
    SC_TRY(ensure_workspace(h, T, G, false, max_iter));
    SC_TRY(grow(&h->d_mc, &h->cap_mc, (size_t)T * h->n));
    int *dy = nullptr;
    if (out_y) {
        dy = out_y;
        if (!dev_io) {
            SC_TRY(grow(&h->d_ylist, &h->cap_ylist, (size_t)batch * std::max(omega, 1)));
            dy = h->d_ylist;
        }
    }
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    // x = [y | 0]: omega distinct positions;  checks = Hin y = H x  (hqc.py:1253-1257)
    SC_HIP(hipMemsetAsync(h->d_mc, 0, sizeof(u64) * (size_t)T * h->n, s));
    if (omega) {
        hipLaunchKernelGGL(k_mc_hqc_secret, dim3(T), dim3(64), (size_t)omega * 64 * sizeof(int), s, h->d_mc, h->n, N, omega,
                           batch, (long)first_trial, k0, k1, dy);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_parity<false>, dim3((h->m + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), T), dim3(256), 0, s, h->d_row_ptr, h->d_col_idx,
                       h->d_mc, h->m, h->n, h->d_synd, (u64 *)nullptr, (const u64 *)nullptr);
    LAUNCH_CHECK();
    // each oracle answer wrong with probability eps
    hipLaunchKernelGGL(k_mc_bernoulli<true>, dim3((h->m + 63) / 64, T), dim3(256), 0, s, h->d_synd, h->m, batch,
                       (long)first_trial, 0u, k0, k1, (const u64 *)nullptr, bernoulli_threshold(eps));
    LAUNCH_CHECK();
    // msg = [0]*N ++ checks (hqc.py:703-705); its syndrome under H = [Hin | I] is `checks` itself
    SC_HIP(hipMemsetAsync(h->d_recv, 0, sizeof(u64) * (size_t)T * h->n, s));
    SC_HIP(hipMemcpy2DAsync(h->d_recv + N, sizeof(u64) * h->n, h->d_synd, sizeof(u64) * h->m, sizeof(u64) * h->m, T,
                            hipMemcpyDeviceToDevice, s));
    if (out_msg) SC_TRY(mc_export_bits(h, h->d_recv, batch, T, dev_io, s, out_msg));
    SC_TRY(run_core(h, batch, T, G, max_iter, method, alpha, early, false, s));
    if (out_y && !dev_io) {
        SC_HIP(hipMemcpyAsync(out_y, dy, sizeof(int) * (size_t)batch * omega, hipMemcpyDeviceToHost, s));
    }
    // decoded[:N] = e[:N] XOR msg[:N] = e[:N] must equal the indicator of y (hqc.py:742-749)
    return mc_finish(h, batch, T, N, dev_io, s, out_success, out_iters);
}

int scaldpc_bp_configure(scaldpc_bp *h, const char *key, const char *value)
{
    if (!h || !key || !value) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device);
    const int old_order = h->kn.var_order;
    if (!set_knob(h->kn, key, value)) return fail(SCALDPC_EINVAL, "unknown knob or bad value: %s=%s", key, value);
    if (h->kn.var_order != old_order && h->d_tile_tab) {  // the k_var records are laid out in launch order: rebuild on next use
        CacheBypass guard(true);
        SC_HIP(hipDeviceSynchronize());
        dev_free(h->d_tile_tab);
        h->d_var_meta = h->d_csc_list = h->d_row_list = nullptr;
    }
    return 0;
}

int scaldpc_bp_device_of(scaldpc_bp *h, int32_t *out)
{
    if (!h || !out) return fail(SCALDPC_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    auto where = [](const void *p) -> int {
        if (!p) return -1;
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();
            return -2;
        }
        return at.device;
    };
    out[0] = h->device;
    out[1] = where(h->d_graph);
    out[2] = where(h->d_msg ? (const void *)h->d_msg : (const void *)h->d_emsg);
    out[3] = where(h->d_synd);
    return 0;
}

int scaldpc_bp_time_kernels(scaldpc_bp *h, int32_t iters, int32_t method, float alpha, void *stream, float *ms,
                            int32_t *launches)
{
    if (!h || !ms || !launches || iters <= 0) return fail(SCALDPC_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->broken) return fail(SCALDPC_EHIP, "this decoder is unusable: an earlier scaldpc_bp_append_rows failed part-way; destroy it and build a new one");
    DeviceGuard dg(h->device);
    if (h->last_group <= 0) return fail(SCALDPC_EINVAL, "no previous decode to time");
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    const int g = h->last_group;
    SC_TRY(ensure_msg(h, g, method));  // (a method other than the last decode's may want the record arrays)
    const int nl = std::min(fixed_lanes(h, g), 2);
    // The variable pass is timed in the form the last decode launched it in: early-exit runs write decisions on EVERY
    // pass (all columns), fixed-iteration runs only on the last one (the passes before it leave out the columns of
    // degree <= 1 in the record form).  launches[5] says which.
    const int wo = h->last_early ? 1 : 0;
    const bool recf = rec_form(h, method);
    const bool slim = recf && !wo && h->kn.rec_skip1;
    const int form_bits = (recf ? 1 : 0) | (wo ? 2 : 0) | (slim ? 4 : 0);
    // `iters` back-to-back launches of each kernel between two events: the event
    // overhead (a few us, comparable to a 65 us launch) is amortised, what remains is
    // the kernel plus the ~1.5 us dependent-launch gap it also pays in a real decode.
    // Two-lane schedule (what a fixed-iteration decode runs): the decode's own launch pattern --
    // each lane alternates check and variable launches over its half of the group, the second
    // lane one kernel out of phase -- with an event after EVERY launch; a launch's duration is
    // the distance between its event and the previous one on the same lane (it includes the
    // dependent-launch gap, as the single-lane series do).  ms[] = sum over the measured
    // launches of both lanes, the first two iterations (phase settling) excluded.
    std::vector<hipEvent_t> ev(6);
    for (auto &e : ev) SC_HIP(hipEventCreate(&e));
    int rc = 0;
    if (nl < 2) {
        SC_HIP(hipEventRecord(ev[0], s));
        for (int it = 0; it < iters && !rc; it++) rc = launch_check(h, method, alpha_for(alpha, it + 1), g, h->d_synd, h->d_done, 0, s);
        SC_HIP(hipEventRecord(ev[1], s));
        SC_HIP(hipEventRecord(ev[2], s));
        for (int it = 0; it < iters && !rc; it++) rc = launch_var(h, g, nullptr, h->d_hard, h->d_done, 0, wo, s, 0, nullptr, recf, true);
        SC_HIP(hipEventRecord(ev[3], s));
        if (!rc) SC_HIP(hipStreamSynchronize(s));
        if (!rc) {
            SC_HIP(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
            SC_HIP(hipEventElapsedTime(&ms[1], ev[2], ev[3]));
            launches[0] = launches[1] = iters;
            launches[2] = launches[4] = g * TW;  // codewords swept per launch
            launches[3] = 1;
            launches[5] = form_bits;
        }
    } else {
        SC_TRY(ensure_lanes(h, 2));
        hipStream_t lane[2] = {s, h->aux_stream[1]};
        const int gs[2] = {g - g / 2, g / 2}, t0[2] = {0, g - g / 2};  // as iterate_tiles deals the tiles
        const int total = iters + 2;
        std::vector<hipEvent_t> mark((size_t)2 * (2 * total + 1));  // per lane: start, then one per launch
        for (auto &e : mark) SC_HIP(hipEventCreate(&e));
        auto M = [&](int k, int i) -> hipEvent_t & { return mark[(size_t)k * (2 * total + 1) + i]; };
        SC_HIP(hipEventRecord(ev[4], s));
        SC_HIP(hipStreamWaitEvent(lane[1], ev[4], 0));
        for (int it = 0; it < total && !rc; it++) {
            for (int k = 0; k < 2 && !rc; k++) {
                if (it == 0) SC_HIP(hipEventRecord(M(k, 0), lane[k]));
                rc = launch_check(h, method, alpha_for(alpha, it + 1), gs[k], h->d_synd + (size_t)t0[k] * h->m, h->d_done + t0[k],
                                  0, lane[k], false, t0[k]);
                SC_HIP(hipEventRecord(M(k, 2 * it + 1), lane[k]));
                if (it == 0 && k == 0) {
                    SC_HIP(hipEventRecord(h->ev_phase[1], lane[0]));
                    SC_HIP(hipStreamWaitEvent(lane[1], h->ev_phase[1], 0));
                }
            }
            for (int k = 0; k < 2 && !rc; k++) {
                rc = launch_var(h, gs[k], nullptr, h->d_hard + (size_t)t0[k] * h->n, h->d_done + t0[k], 0, wo, lane[k], t0[k], nullptr,
                                recf, true);
                SC_HIP(hipEventRecord(M(k, 2 * it + 2), lane[k]));
            }
        }
        SC_HIP(hipEventRecord(ev[5], lane[1]));
        SC_HIP(hipStreamWaitEvent(s, ev[5], 0));
        if (!rc) SC_HIP(hipStreamSynchronize(s));
        if (!rc) {
            double tc = 0.0, tv = 0.0;
            int nc = 0, nv = 0;
            for (int k = 0; k < 2; k++)
                for (int it = 2; it < total; it++) {
                    float d = 0.0f;
                    SC_HIP(hipEventElapsedTime(&d, M(k, 2 * it), M(k, 2 * it + 1)));
                    tc += d;
                    nc++;
                    SC_HIP(hipEventElapsedTime(&d, M(k, 2 * it + 1), M(k, 2 * it + 2)));
                    tv += d;
                    nv++;
                }
            // The event after every launch costs the schedule its back-to-back dispatch (~2.5 us per launch: with the 30-50
            // us launches of the record form the evented pair came out 6 % above rocprofv3's).  A second pass with the same
            // launch pattern and NO events in between gives the pair's true duration per lane; the evented series supply
            // only the check : variable ratio.
            double pair_ms = 0.0;
            {
                hipEvent_t &b0 = M(0, 0), &e0 = M(0, 1), &b1 = M(1, 0), &e1 = M(1, 1);
                SC_HIP(hipEventRecord(ev[4], s));
                SC_HIP(hipStreamWaitEvent(lane[1], ev[4], 0));
                for (int it = 0; it < total && !rc; it++) {
                    if (it == 2) {
                        SC_HIP(hipEventRecord(b0, lane[0]));
                        SC_HIP(hipEventRecord(b1, lane[1]));
                    }
                    for (int k = 0; k < 2 && !rc; k++) {
                        rc = launch_check(h, method, alpha_for(alpha, it + 1), gs[k], h->d_synd + (size_t)t0[k] * h->m, h->d_done + t0[k],
                                          0, lane[k], false, t0[k]);
                        if (it == 0 && k == 0) {
                            SC_HIP(hipEventRecord(h->ev_phase[1], lane[0]));
                            SC_HIP(hipStreamWaitEvent(lane[1], h->ev_phase[1], 0));
                        }
                    }
                    for (int k = 0; k < 2 && !rc; k++)
                        rc = launch_var(h, gs[k], nullptr, h->d_hard + (size_t)t0[k] * h->n, h->d_done + t0[k], 0, wo, lane[k], t0[k],
                                        nullptr, recf, true);
                }
                SC_HIP(hipEventRecord(e0, lane[0]));
                SC_HIP(hipEventRecord(e1, lane[1]));
                SC_HIP(hipEventRecord(ev[5], lane[1]));
                SC_HIP(hipStreamWaitEvent(s, ev[5], 0));
                if (!rc) SC_HIP(hipStreamSynchronize(s));
                if (!rc) {
                    float d0 = 0.0f, d1 = 0.0f;
                    SC_HIP(hipEventElapsedTime(&d0, b0, e0));
                    SC_HIP(hipEventElapsedTime(&d1, b1, e1));
                    pair_ms = 0.5 * ((double)d0 + (double)d1) / (total - 2);
                }
            }
            if (pair_ms > 0.0 && tc + tv > 0.0 && nc == nv && nc > 0) {
                const double evented_pair = (tc + tv) / nc, scale = pair_ms / evented_pair;
                tc *= scale;
                tv *= scale;
            }
            ms[0] = (float)tc;
            ms[1] = (float)tv;
            launches[0] = nc;
            launches[1] = nv;
            launches[2] = gs[0] * TW;
            launches[4] = gs[0] * TW;  // (odd groups: the second lane's launches are one tile smaller)
            launches[3] = 2;
            launches[5] = form_bits;
        }
        for (auto &e : mark) (void)hipEventDestroy(e);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

void scaldpc_bp_destroy(scaldpc_bp *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    // A handle that took SCALDPC_F_ASYNC calls may still have work in flight, on the caller's stream and
    // on its own lanes: wait for the device, and send its blocks through hipFree instead of parking
    // them for the next handle.  The guard must be in place BEFORE the first block is released.
    CacheBypass guard(h->async_used);
    if (h->async_used) (void)hipDeviceSynchronize();
    dev_free(h->d_graph);  // graph arrays and d_prior are views into it (until the graph grows)
    dev_free(h->d_csr_rp); dev_free(h->d_csr_ci); dev_free(h->d_prior_buf); dev_free(h->d_pairs);
    cached_free(h->h_pairs);
    dev_free(h->d_tile_tab);
    dev_free(h->d_first_tab);
    dev_free(h->d_el_tab);
    dev_free(h->d_msg); dev_free(h->d_scratch); dev_free(h->d_post);
    dev_free(h->d_rec); dev_free(h->d_mask);
    dev_free(h->d_synd); dev_free(h->d_recv); dev_free(h->d_hard); dev_free(h->d_done);
    dev_free(h->d_conv); dev_free(h->d_unsat); dev_free(h->d_iters); dev_free(h->d_remaining);
    dev_free(h->d_in); dev_free(h->d_out_bits); dev_free(h->d_out_conv); dev_free(h->d_out_llr);
    dev_free(h->d_out_iters);
    for (auto &L : h->lv) {
        dev_free(L.synd); dev_free(L.hard); dev_free(L.done); dev_free(L.conv); dev_free(L.unsat);
        dev_free(L.iters); dev_free(L.ids); dev_free(L.slot_of); dev_free(L.post);
    }
    dev_free(h->d_emsg); dev_free(h->d_el_unsat);
    dev_free(h->d_thr); dev_free(h->d_mc); dev_free(h->d_diff); dev_free(h->d_ylist); dev_free(h->d_succ);
    cached_free(h->h_remaining);
    cached_free(h->h_io);  // the small-call staging pair (decode_batch's fused host path)
    dev_free(h->d_out_all);
    if (h->own_stream) stream_release(h->own_stream, h->device);
    for (int k = 0; k < 4; k++) {
        if (h->aux_stream[k]) stream_release(h->aux_stream[k], h->device);
        if (h->ev_join[k]) (void)hipEventDestroy(h->ev_join[k]);
        if (h->ev_phase[k]) (void)hipEventDestroy(h->ev_phase[k]);
    }
    delete h;
}

}  // extern "C"

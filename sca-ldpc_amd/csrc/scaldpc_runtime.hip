// libscaldpc -- process-wide plumbing shared by the binary and the q-ary decoders: the
// thread-local error string, the block cache behind dev_alloc / dev_free, the pool of
// recycled streams, and the handful of C-ABI entry points that belong to no handle.
#include "scaldpc_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace scaldpc {
std::string &last_error()
{
    static thread_local std::string s;
    return s;
}
int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// ---- block cache behind dev_alloc / dev_free (see scaldpc_common.h) ----------------------
namespace {
struct Block {
    size_t bytes;
    int device;  // -1: pinned host memory
};
struct BlockCache {
    std::mutex mu;
    std::unordered_map<void *, Block> live;       // every block handed out and still owned by a handle
    std::vector<std::pair<void *, Block>> idle;   // released, ready for reuse
    size_t idle_bytes = 0;
};
BlockCache &block_cache()
{
    static BlockCache *c = new BlockCache();  // never destroyed: the HIP runtime may be gone at exit
    return *c;
}
constexpr size_t CACHE_BLOCK_MAX = (size_t)64 << 20, CACHE_TOTAL_MAX = (size_t)512 << 20;
thread_local bool tl_bypass = false;  // set while serving a handle that may still have asynchronous work in flight
bool cache_enabled()
{
    static const bool on = getenv("SCALDPC_NO_CACHE") == nullptr;
    return on && !tl_bypass;
}
void raw_free(void *p, const Block &b)
{
    if (b.device < 0)
        (void)hipHostFree(p);
    else
        (void)hipFree(p);
}
}  // namespace

// SCALDPC_POISON=1 (tests): every block handed out is filled with 0xFF first, so that code
// relying on fresh or recycled memory being zero shows up as a wrong result
int poison(void *p, size_t bytes, bool pinned_host)
{
    static const bool on = getenv("SCALDPC_POISON") != nullptr;
    if (!on) return 0;
    if (pinned_host) {
        memset(p, 0xFF, bytes);
    } else {
        SC_HIP(hipMemset(p, 0xFF, bytes));
        SC_HIP(hipDeviceSynchronize());
    }
    return 0;
}

CacheBypass::CacheBypass(bool on) : prev(tl_bypass) { tl_bypass = prev || on; }
CacheBypass::~CacheBypass() { tl_bypass = prev; }

// scaldpc_debug_fail_alloc (tests): the k-th allocation from now fails with SCALDPC_ENOMEM.  The injector exists only in a
// process started with SCALDPC_DEBUG=1 (read once): without it the export refuses and the allocator never looks at the
// countdown, so no caller or stray thread of a production process can poison the shared library.
std::atomic<int> fail_countdown{0};
bool debug_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SCALDPC_DEBUG");
        return e && *e && strcmp(e, "0") != 0;
    }();
    return on;
}

int cached_alloc(void **p, size_t bytes, bool pinned_host)
{
    *p = nullptr;
    if (debug_enabled() && fail_countdown.load(std::memory_order_relaxed) > 0 && fail_countdown.fetch_sub(1) == 1)
        return fail(SCALDPC_ENOMEM, "allocation of %zu bytes failed: injected by scaldpc_debug_fail_alloc", bytes);
    bytes = (bytes + 255) / 256 * 256;
    int dev = -1;
    if (!pinned_host) SC_HIP(hipGetDevice(&dev));
    BlockCache &bc = block_cache();
    if (cache_enabled() && bytes <= CACHE_BLOCK_MAX) {
        std::lock_guard<std::mutex> lk(bc.mu);
        size_t best = bc.idle.size();
        for (size_t i = 0; i < bc.idle.size(); i++) {  // smallest block that fits without wasting more than half
            const Block &b = bc.idle[i].second;
            if (b.device == dev && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 &&
                (best == bc.idle.size() || b.bytes < bc.idle[best].second.bytes))
                best = i;
        }
        if (best != bc.idle.size()) {
            *p = bc.idle[best].first;
            bc.live.emplace(*p, bc.idle[best].second);
            const size_t got = bc.idle[best].second.bytes;
            bc.idle_bytes -= got;
            bc.idle.erase(bc.idle.begin() + best);
            return poison(*p, got, pinned_host);
        }
    }
    hipError_t e = pinned_host ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {  // give the parked blocks back and try once more
        (void)hipGetLastError();
        scaldpc_trim();
        e = pinned_host ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(e == hipErrorOutOfMemory ? SCALDPC_ENOMEM : SCALDPC_EHIP, "allocation of %zu bytes failed: %s", bytes,
                    hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lk(bc.mu);
        bc.live.emplace(*p, Block{bytes, dev});
    }
    return poison(*p, bytes, pinned_host);
}

void cached_free(void *p)
{
    if (!p) return;
    BlockCache &bc = block_cache();
    Block b{0, 0};
    {
        std::lock_guard<std::mutex> lk(bc.mu);
        auto it = bc.live.find(p);
        if (it == bc.live.end()) return;  // not ours (cannot happen through dev_free)
        b = it->second;
        bc.live.erase(it);
        if (cache_enabled() && b.bytes <= CACHE_BLOCK_MAX && bc.idle_bytes + b.bytes <= CACHE_TOTAL_MAX) {
            bc.idle.emplace_back(p, b);
            bc.idle_bytes += b.bytes;
            return;
        }
    }
    raw_free(p, b);
}

// ---- host pool behind PoolAlloc (see scaldpc_common.h) -----------------------------------------------------
namespace {
struct HostPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> idle;  // released blocks: pointer, capacity
    size_t idle_bytes = 0;
};
HostPool &host_pool()
{
    static HostPool *p = new HostPool();  // never destroyed (static destruction order)
    return *p;
}
constexpr size_t HOST_POOL_MIN = (size_t)32 << 10, HOST_POOL_TOTAL_MAX = (size_t)256 << 20, HOST_HDR = 64;
}  // namespace

// Every pooled block carries its capacity in a 64-byte header in front of the user pointer.
void *host_pool_alloc(size_t bytes)
{
    if (bytes < HOST_POOL_MIN) return ::operator new(bytes);
    HostPool &hp = host_pool();
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        size_t best = hp.idle.size();
        for (size_t i = 0; i < hp.idle.size(); i++)  // smallest block that fits without wasting more than half
            if (hp.idle[i].second >= bytes && hp.idle[i].second <= 2 * bytes + 4096 &&
                (best == hp.idle.size() || hp.idle[i].second < hp.idle[best].second))
                best = i;
        if (best != hp.idle.size()) {
            void *p = hp.idle[best].first;
            hp.idle_bytes -= hp.idle[best].second;
            hp.idle.erase(hp.idle.begin() + best);
            return p;
        }
    }
    const size_t cap = bytes + bytes / 8;  // a little headroom: the next decoder of a growing graph fits the same block
    char *raw = static_cast<char *>(::operator new(cap + HOST_HDR));
    *reinterpret_cast<size_t *>(raw) = cap;
    return raw + HOST_HDR;
}

void host_pool_free(void *p, size_t bytes) noexcept
{
    if (!p) return;
    if (bytes < HOST_POOL_MIN) {
        ::operator delete(p);
        return;
    }
    char *raw = static_cast<char *>(p) - HOST_HDR;
    const size_t cap = *reinterpret_cast<size_t *>(raw);
    HostPool &hp = host_pool();
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        if (hp.idle_bytes + cap <= HOST_POOL_TOTAL_MAX) {
            hp.idle.emplace_back(p, cap);
            hp.idle_bytes += cap;
            return;
        }
    }
    ::operator delete(raw);
}

// Streams of destroyed handles are parked per device and handed to the next handle created
// there (creating and destroying a stream per decoder costs more than a single decode).
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
};
StreamPool &stream_pool()
{
    static StreamPool *p = new StreamPool();  // never destroyed: the HIP runtime may be gone at exit
    return *p;
}
int stream_acquire(hipStream_t *out, int *device)
{
    int dev = 0;
    SC_HIP(hipGetDevice(&dev));
    *device = dev;
    {
        StreamPool &sp = stream_pool();
        std::lock_guard<std::mutex> lk(sp.mu);
        for (size_t i = 0; i < sp.idle.size(); i++)
            if (sp.idle[i].first == dev) {
                *out = sp.idle[i].second;
                sp.idle.erase(sp.idle.begin() + i);
                return 0;
            }
    }
    SC_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return 0;
}
void stream_release(hipStream_t s, int dev)
{
    StreamPool &sp = stream_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    // nothing is in flight: a handle synchronises its streams before every return, and one that took
    // SCALDPC_F_ASYNC calls synchronises the device in its destroy before it gets here
    if (sp.idle.size() < 64) sp.idle.emplace_back(dev, s); else (void)hipStreamDestroy(s);
}


}  // namespace scaldpc

extern "C" int scaldpc_trim(void)
{
    using namespace scaldpc;
    BlockCache &bc = block_cache();
    std::vector<std::pair<void *, Block>> drop;
    {
        std::lock_guard<std::mutex> lk(bc.mu);
        drop.swap(bc.idle);
        bc.idle_bytes = 0;
    }
    for (auto &d : drop) raw_free(d.first, d.second);
    std::vector<std::pair<void *, size_t>> hdrop;
    {
        HostPool &hp = host_pool();
        std::lock_guard<std::mutex> lk(hp.mu);
        hdrop.swap(hp.idle);
        hp.idle_bytes = 0;
    }
    for (auto &d : hdrop) ::operator delete(static_cast<char *>(d.first) - HOST_HDR);
    return 0;
}

using namespace scaldpc;

extern "C" {

const char *scaldpc_last_error(void) { return last_error().c_str(); }
int scaldpc_version(void) { return SCALDPC_VERSION; }

int scaldpc_debug_live_blocks(int64_t *out)
{
    if (!out) return fail(SCALDPC_EINVAL, "out is NULL");
    BlockCache &bc = block_cache();
    std::lock_guard<std::mutex> lk(bc.mu);
    for (int i = 0; i < 6; i++) out[i] = 0;
    for (auto &kv : bc.live) {
        const int k = kv.second.device < 0 ? 2 : 0;
        out[k]++;
        out[k + 1] += (int64_t)kv.second.bytes;
    }
    out[4] = (int64_t)bc.idle.size();
    out[5] = (int64_t)bc.idle_bytes;
    return 0;
}

int scaldpc_debug_fail_alloc(int32_t countdown)
{
    if (!debug_enabled()) {
        fail_countdown.store(0);
        return countdown > 0 ? fail(SCALDPC_EINVAL, "scaldpc_debug_fail_alloc: start the process with SCALDPC_DEBUG=1 to arm the fault injector") : 0;
    }
    fail_countdown.store(countdown > 0 ? countdown : 0);
    return 0;
}

// ---- measurement aid: the in-place stream an in-place BP pass is made of ---------------------------------------------
extern "C++" {
namespace {
// each wave owns `rows` consecutive 256-B rows (64 lanes x 4 B): reads them all, then writes them all back -- the access
// shape of an in-place check pass over a row of `rows` edges (profiles/microbench/rmw_stream.hip, dword form)
__global__ __launch_bounds__(256) void k_rmw_stream(float *buf, size_t nrows, int rows)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t r0 = wave * (size_t)rows;
    if (r0 + rows > nrows) return;
    float *p = buf + r0 * 64 + lane;
    float acc = 0.0f;
    for (int k = 0; k < rows; k++) acc += p[(size_t)k * 64];
    for (int k = 0; k < rows; k++) p[(size_t)k * 64] = acc + (float)k;
}
// the same for rows of up to 64 edges with the row in registers: every load is issued before the first use, as the
// product's row kernels do (the loop form above waits on each load in turn: 5.8 against 7.0 TB/s on a 209 MB buffer,
// profiles/r04/stream_modes.log)
template <int ROWS>
__global__ __launch_bounds__(256) void k_rmw_stream_regs(float *buf, size_t nrows)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t r0 = wave * (size_t)ROWS;
    if (r0 + ROWS > nrows) return;
    float *p = buf + r0 * 64 + lane;
    float x[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) x[k] = p[(size_t)k * 64];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < ROWS; k++) acc += x[k];
#pragma unroll
    for (int k = 0; k < ROWS; k++) p[(size_t)k * 64] = acc + (float)k;
}
void launch_rmw_stream(dim3 grid, float *buf, size_t nrows, int rows)
{
#define RS(R) case R: hipLaunchKernelGGL(k_rmw_stream_regs<R>, grid, dim3(256), 0, 0, buf, nrows); break;
    switch (rows) {  // (the row widths of the bench graphs and a few round ones; any other width takes the loop form)
        RS(8) RS(16) RS(24) RS(32) RS(40) RS(48) RS(51) RS(56) RS(64)
        default: hipLaunchKernelGGL(k_rmw_stream, grid, dim3(256), 0, 0, buf, nrows, rows);
    }
#undef RS
}
}  // namespace
}  // extern "C++"

int scaldpc_measure_rmw_stream(int64_t bytes, int32_t rows_per_wave, int32_t reps, double *gbps)
{
    if (!gbps || bytes < 256 || rows_per_wave < 1 || rows_per_wave > 4096 || reps < 1)
        return fail(SCALDPC_EINVAL, "scaldpc_measure_rmw_stream: bad argument");
    const size_t nrows = (size_t)bytes / 256 / rows_per_wave * rows_per_wave;
    if (nrows == 0) return fail(SCALDPC_EINVAL, "scaldpc_measure_rmw_stream: buffer smaller than one wave's rows");
    float *buf = nullptr;
    SC_HIP(hipMalloc(&buf, nrows * 256));
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMemset(buf, 0, nrows * 256);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    float ms = 0.0f;
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((nrows / rows_per_wave + 3) / 4));
        for (int i = 0; i < 3; i++) launch_rmw_stream(grid, buf, nrows, (int)rows_per_wave);
        e = hipEventRecord(a, 0);
        for (int i = 0; i < reps; i++) launch_rmw_stream(grid, buf, nrows, (int)rows_per_wave);
        if (e == hipSuccess) e = hipEventRecord(b, 0);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(SCALDPC_EHIP, "scaldpc_measure_rmw_stream: %s", hipGetErrorString(e));
    *gbps = 2.0 * (double)nrows * 256.0 * reps / ((double)ms * 1e-3) / 1e9;
    return 0;
}

int scaldpc_device_count(int *count)
{
    if (!count) return fail(SCALDPC_EINVAL, "count is NULL");
    SC_HIP(hipGetDeviceCount(count));
    return 0;
}

int scaldpc_set_device(int device)
{
    SC_HIP(hipSetDevice(device));
    return 0;
}

}  // extern "C"

// Shared host-side plumbing of libscaldpc (error string, HIP check macro).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/scaldpc.h"

namespace scaldpc {

std::string &last_error();
int fail(int code, const char *fmt, ...);

#define SC_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::scaldpc::fail(e__ == hipErrorOutOfMemory ? SCALDPC_ENOMEM : SCALDPC_EHIP, \
                                   "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),   \
                                   __FILE__, __LINE__);                                      \
    } while (0)

#define SC_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__) return rc__;  \
    } while (0)

// Device / pinned-host memory through a small per-process cache of released blocks
// (scaldpc_bp.hip): the reference builds a NEW decoder for every decode (hqc.py:694), and a
// handle's ~20 hipMalloc / hipFree pairs (hipFree synchronises the device) cost several times
// the decode itself.  Blocks up to 64 MiB are parked on release (at most 512 MiB in total,
// scaldpc_trim() returns them to the driver, SCALDPC_NO_CACHE=1 disables the cache); larger
// ones go straight to hipMalloc / hipFree.  Memory comes back uninitialised either way.
int cached_alloc(void **p, size_t bytes, bool pinned_host);
void cached_free(void *p);
// While alive on a thread, released blocks go to hipFree / hipHostFree (which wait for the
// device) instead of the cache: used when a handle may have asynchronous work in flight.
struct CacheBypass {
    explicit CacheBypass(bool on);
    ~CacheBypass();
    bool prev;
};

// Every entry point that takes a handle runs on the handle's device whatever the calling thread's
// current device is (allocations, streams and launches all follow the current device), and puts
// the caller's device back on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Streams of destroyed handles are parked per device and handed to the next handle created there.
int stream_acquire(hipStream_t *out, int *device);
void stream_release(hipStream_t s, int device);

// Host memory for a handle's graph mirrors and tables, through a small per-process pool of released blocks.  The
// reference builds a NEW decoder for every decode (hqc.py:694), i.e. a few multi-megabyte std::vectors are allocated and
// freed per decode; glibc hands such sizes to mmap / trims the heap top on free, so the next decoder page-faults every
// byte again -- and depending on what else the process (Python, NumPy) has on the heap this flips between 0.45 ms and
// 2-8 ms per construction plus 1.5 ms per destroy (measured: profiles/r03/rebuild_probe.log).  Blocks of at least 32 KiB
// are parked on release (at most 256 MiB in total; scaldpc_trim() returns them); smaller requests go to operator new.
void *host_pool_alloc(size_t bytes);
void host_pool_free(void *p, size_t bytes) noexcept;

template <typename T>
struct PoolAlloc {
    using value_type = T;
    PoolAlloc() noexcept = default;
    template <typename U>
    PoolAlloc(const PoolAlloc<U> &) noexcept {}
    T *allocate(size_t n) { return static_cast<T *>(host_pool_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) noexcept { host_pool_free(p, n * sizeof(T)); }
    template <typename U>
    bool operator==(const PoolAlloc<U> &) const noexcept { return true; }
    template <typename U>
    bool operator!=(const PoolAlloc<U> &) const noexcept { return false; }
};
template <typename T>
using pvec = std::vector<T, PoolAlloc<T>>;

template <typename T>
inline int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    return cached_alloc((void **)p, count * sizeof(T), false);
}

template <typename T>
inline void dev_free(T *&p)
{
    if (p) cached_free((void *)p);
    p = nullptr;
}

}  // namespace scaldpc

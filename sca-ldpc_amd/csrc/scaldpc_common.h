// Shared host-side plumbing of libscaldpc (error string, HIP check macro).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

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

template <typename T>
inline int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    SC_HIP(hipMalloc((void **)p, count * sizeof(T)));
    return 0;
}

template <typename T>
inline void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace scaldpc

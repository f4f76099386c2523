// Helpers shared by the translation units of libdspx.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "dspx.h"

namespace dspx {

extern thread_local std::string g_last_error;

// Records a thread-local message and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace dspx

#define DSPX_HIP(call)                                                                                    \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return dspx::fail(e_ == hipErrorOutOfMemory ? DSPX_ERR_NOMEM : DSPX_ERR_HIP, "%s: %s", #call, \
                              hipGetErrorString(e_));                                                     \
    } while (0)

// Helpers shared by the translation units of libdspx.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dspx.h"

namespace dspx {

extern thread_local std::string g_last_error;

// Records a thread-local message and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Workgroups per launch: HIP launches nothing (and reports nothing) when grid * block reaches 2^32 threads,
// so with 256-thread workgroups a grid stays below 2^24.
constexpr unsigned long long kMaxGrid = (1ull << 24) - 1;

// Taps from which the FP64 matrix pipe is faster than the vector pipe (measured on MI355X).
constexpr size_t kMatrixPipeMinTaps = 52;

inline int check_flags(unsigned flags) {
    if (flags & ~(DSPX_FIR_EXACT | DSPX_FIR_MFMA | DSPX_FIR_VECTOR | DSPX_FIR_TILED))
        return fail(DSPX_ERR_INVALID, "unknown flag bits 0x%x", flags);
    if ((flags & DSPX_FIR_MFMA) && (flags & DSPX_FIR_VECTOR))
        return fail(DSPX_ERR_INVALID, "DSPX_FIR_MFMA and DSPX_FIR_VECTOR exclude each other");
    return DSPX_OK;
}

// Which pipe a call runs on; `ntaps` is the (mean) tap count of the filter(s).
inline bool use_matrix_pipe(unsigned flags, size_t ntaps) {
    if (flags & (DSPX_FIR_EXACT | DSPX_FIR_VECTOR)) return false;
    if (flags & DSPX_FIR_MFMA) return true;
    return ntaps >= kMatrixPipeMinTaps;
}

// Copies `rows` rows of `n` doubles from device memory (row r at d_src + r*src_pitch) into host memory (row r
// to dst + (row_map ? row_map[r] : r)*dst_stride) through the pinned staging pipeline of the host-pointer
// entry points (defined in dspx_capi.hip).
int download_rows(int device, const double* d_src, size_t src_pitch, size_t rows, size_t n, double* dst,
                  size_t dst_stride, const size_t* row_map);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Test hook for machines with fewer GPUs than a sharding is meant for.  It is NOT part of the C ABI (include/dspx.h
// does not declare it) and nothing in the environment can switch it on: a test harness calls
//     extern "C" int dspx_test_alias_devices(int enable);   // returns the previous setting
// after which the multi-device entry points accept any ngpus up to kMaxAliasedDevices and run logical device g on
// physical device g % (visible devices), each logical device with its own thread, staging engine and plan, exactly
// as on a machine that has them.  Results do not depend on it (nor on ngpus at all).  An entry point reads the
// setting once, in the calling thread, and hands it to its workers.
constexpr int kMaxAliasedDevices = 16;
bool devices_aliased();  // the current setting (dspx_capi.hip)
int g_alias_devices_exchange(bool on);
inline int logical_device_limit(bool aliased, int visible) { return aliased ? kMaxAliasedDevices : visible; }
inline int physical_device(bool aliased, int logical, int visible) { return aliased && visible > 0 ? logical % visible : logical; }

// Threads that are joined when the owner goes out of scope, whichever way it leaves.
struct Joiner {
    std::vector<std::thread> threads;
    void join() {
        for (auto& t : threads)
            if (t.joinable()) t.join();
        threads.clear();
    }
    ~Joiner() { join(); }
};

// Runs the body of an extern "C" entry point: no C++ exception leaves the library.
template <class F> int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(DSPX_ERR_NOMEM, "out of host memory");
    } catch (const std::invalid_argument& e) {
        return fail(DSPX_ERR_INVALID, "%s", e.what());
    } catch (const std::exception& e) {
        return fail(DSPX_ERR_HOST, "%s", e.what());
    } catch (...) {
        return fail(DSPX_ERR_HOST, "unknown C++ exception");
    }
}

}  // namespace dspx

#define DSPX_HIP(call)                                                                                    \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return dspx::fail(e_ == hipErrorOutOfMemory ? DSPX_ERR_NOMEM : DSPX_ERR_HIP, "%s: %s", #call, \
                              hipGetErrorString(e_));                                                     \
    } while (0)

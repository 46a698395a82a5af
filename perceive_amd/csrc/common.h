// common.h — shared host-side helpers of libperceive_hip (error plumbing, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/perceive_hip.h"

namespace pcv {

// thread-local last-error text behind pcv_last_error()
void set_error(const char* fmt, ...);
const char* last_error();

struct Error {
    pcv_status status;
};

#define PCV_FAIL(status, ...)            \
    do {                                 \
        ::pcv::set_error(__VA_ARGS__);   \
        throw ::pcv::Error{(status)};    \
    } while (0)

#define PCV_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            PCV_FAIL(PCV_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                     __FILE__, __LINE__);                                                      \
    } while (0)

#define PCV_REQUIRE(cond, ...)                              \
    do {                                                    \
        if (!(cond)) PCV_FAIL(PCV_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// Kernels that need more than 64 KB of dynamic LDS have to be allowed that once — per DEVICE (the runtime keeps one
// function object per device) — before their first launch there.  One registry for the whole library, keyed by (current
// device, kernel) under a mutex: two handles on two host threads may launch the same kernel for the first time at once, and
// a process may hold contexts on several devices (pcv_init per device).  Raises the limit only when `bytes` is above what
// this (device, kernel) already has; cheap otherwise.
void allow_dynamic_lds(const void* kernel, size_t bytes);
// compute units of the current device (cached per device)
int current_device_cus();

// Wrap a C-ABI entry point: nothing may escape as an exception.
template <class F>
static inline pcv_status guarded(F&& f) {
    try {
        f();
        return PCV_OK;
    } catch (const Error& e) {
        return e.status;
    } catch (const std::exception& e) {
        set_error("internal error: %s", e.what());
        return PCV_ERR_INTERNAL;
    } catch (...) {
        set_error("internal error: unknown exception");
        return PCV_ERR_INTERNAL;
    }
}

}  // namespace pcv

struct pcv_ctx {
    int device = -1;
    hipStream_t stream = nullptr;      // where every handle of this context queues its work
    hipStream_t own_stream = nullptr;  // the stream pcv_init created (stream == own_stream unless adopted)
    hipDeviceProp_t props;
    int num_cus = 0;
    // scratch of pcv_merge_topk*: device output + pinned host copy, grown on demand
    void* merge_dev = nullptr;
    void* merge_pin = nullptr;
    size_t merge_cap = 0;
};

// context.cpp — device context, error plumbing and the embedding blob codec of the C ABI.
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace pcv {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
const char* last_error() { return g_last_error.c_str(); }

namespace {
std::mutex g_attr_mu;
std::map<std::pair<int, const void*>, size_t> g_lds_allowed;  // (device, kernel) -> dynamic LDS bytes allowed so far
std::map<int, int> g_device_cus;
}  // namespace

void allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return;  // every kernel may have that much without asking
    int dev = 0;
    PCV_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    size_t& have = g_lds_allowed[{dev, kernel}];
    if (bytes <= have) return;
    PCV_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
}

int current_device_cus() {
    int dev = 0;
    PCV_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    int& n = g_device_cus[dev];
    if (n <= 0) {
        PCV_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace pcv

using namespace pcv;

extern "C" {

const char* pcv_last_error(void) { return pcv::last_error(); }

const char* pcv_version(void) { return "perceive-hip 0.1 (gfx950)"; }

int pcv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

pcv_status pcv_init(int device_index, pcv_ctx** out_ctx) {
    return guarded([&] {
        PCV_REQUIRE(out_ctx != nullptr, "pcv_init: out_ctx is NULL");
        *out_ctx = nullptr;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0)
            PCV_FAIL(PCV_ERR_DEVICE, "pcv_init: no HIP device visible (%s)",
                     e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        PCV_REQUIRE(device_index >= 0 && device_index < n, "pcv_init: device %d out of range [0,%d)", device_index, n);
        PCV_HIP(hipSetDevice(device_index));
        auto* ctx = new pcv_ctx();
        ctx->device = device_index;
        PCV_HIP(hipGetDeviceProperties(&ctx->props, device_index));
        ctx->num_cus = ctx->props.multiProcessorCount;
        PCV_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
        *out_ctx = ctx;
    });
}

pcv_status pcv_shutdown(pcv_ctx* ctx) {
    return guarded([&] {
        if (!ctx) return;
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->own_stream) {
            (void)hipStreamSynchronize(ctx->own_stream);
            (void)hipStreamDestroy(ctx->own_stream);
        }
        if (ctx->merge_dev) (void)hipFree(ctx->merge_dev);
        if (ctx->merge_pin) (void)hipHostFree(ctx->merge_pin);
        delete ctx;
    });
}

pcv_status pcv_synchronize(pcv_ctx* ctx) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr, "pcv_synchronize: ctx is NULL");
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipStreamSynchronize(ctx->stream));
    });
}

void* pcv_stream(pcv_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

pcv_status pcv_set_stream(pcv_ctx* ctx, void* hip_stream, int adopt) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr, "pcv_set_stream: ctx is NULL");
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipStreamSynchronize(ctx->stream));  // nothing of ours is left behind on the old stream
        ctx->stream = adopt ? (hipStream_t)hip_stream : ctx->own_stream;
        const hipError_t q = hipStreamQuery(ctx->stream);  // a handle of another runtime fails here
        if (q != hipSuccess && q != hipErrorNotReady) {
            ctx->stream = ctx->own_stream;
            PCV_HIP(q);
        }
    });
}

pcv_status pcv_device_alloc(pcv_ctx* ctx, size_t n_bytes, void** out_dptr) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && out_dptr != nullptr, "device_alloc: NULL argument");
        *out_dptr = nullptr;
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipMalloc(out_dptr, n_bytes ? n_bytes : 1));
    });
}

pcv_status pcv_device_free(pcv_ctx* ctx, void* dptr) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr, "device_free: ctx is NULL");
        if (!dptr) return;
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipStreamSynchronize(ctx->stream));
        PCV_HIP(hipFree(dptr));
    });
}

pcv_status pcv_copy_to_host(pcv_ctx* ctx, void* dst_host, const void* src_dev, size_t n_bytes) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && (n_bytes == 0 || (dst_host && src_dev)), "copy_to_host: NULL argument");
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipMemcpyAsync(dst_host, src_dev, n_bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCV_HIP(hipStreamSynchronize(ctx->stream));
    });
}

pcv_status pcv_copy_to_device(pcv_ctx* ctx, void* dst_dev, const void* src_host, size_t n_bytes) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && (n_bytes == 0 || (dst_dev && src_host)), "copy_to_device: NULL argument");
        PCV_HIP(hipSetDevice(ctx->device));
        PCV_HIP(hipMemcpyAsync(dst_dev, src_host, n_bytes, hipMemcpyHostToDevice, ctx->stream));
        PCV_HIP(hipStreamSynchronize(ctx->stream));
    });
}

// search.rs:281-286
pcv_status pcv_deserialize_embedding(const uint8_t* blob, size_t n_bytes, float* out, size_t out_cap,
                                     size_t* out_len) {
    return guarded([&] {
        PCV_REQUIRE(blob != nullptr || n_bytes == 0, "deserialize_embedding: blob is NULL");
        PCV_REQUIRE(n_bytes % 4 == 0, "deserialize_embedding: %zu bytes is not a whole number of f32", n_bytes);
        const size_t n = n_bytes / 4;
        PCV_REQUIRE(out != nullptr && out_cap >= n, "deserialize_embedding: output holds %zu values, need %zu",
                    out_cap, n);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t u = (uint32_t)blob[4 * i] | ((uint32_t)blob[4 * i + 1] << 8) |
                               ((uint32_t)blob[4 * i + 2] << 16) | ((uint32_t)blob[4 * i + 3] << 24);
            std::memcpy(&out[i], &u, 4);
        }
        if (out_len) *out_len = n;
    });
}

// search.rs:288-294
pcv_status pcv_serialize_embedding(const float* v, size_t n, uint8_t* out, size_t out_cap) {
    return guarded([&] {
        PCV_REQUIRE(v != nullptr || n == 0, "serialize_embedding: input is NULL");
        PCV_REQUIRE(out != nullptr && out_cap >= 4 * n, "serialize_embedding: output holds %zu bytes, need %zu",
                    out_cap, 4 * n);
        for (size_t i = 0; i < n; ++i) {
            uint32_t u;
            std::memcpy(&u, &v[i], 4);
            out[4 * i + 0] = (uint8_t)(u & 0xff);
            out[4 * i + 1] = (uint8_t)((u >> 8) & 0xff);
            out[4 * i + 2] = (uint8_t)((u >> 16) & 0xff);
            out[4 * i + 3] = (uint8_t)((u >> 24) & 0xff);
        }
    });
}

}  // extern "C"

// What does one cold f32 row cost a wave while the HBM is streaming?  Two kernels on two streams:
//   stream_kernel : 3072 waves stream a 38 GB buffer with non-temporal 16-byte loads (the int8 scan's traffic), in a loop;
//   probe_kernel  : a few hundred waves, each reading random 1536-byte rows of a 150 GB buffer, every row timed with
//                   s_memrealtime (100 MHz), in three forms:
//        0  blocked layout: piece f4 of the row at 512-byte stride (96 pieces = 96 sectors), lanes 0..63 then 0..31
//        1  row-major: the same 96 pieces contiguous (1536 B = 12 x 128-byte sectors)
//        2  blocked, only the first 64 pieces (one load instruction)
// Prints the median / mean latency per row with and without the streaming load.
//   make -C tools/ubench && gpurun -- tools/ubench/cold_row
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

__global__ __launch_bounds__(256) void stream_kernel(const f32x4* buf, size_t n16, int reps, float* sink, const int* stop) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps && !*(volatile const int*)stop; ++r)
        for (size_t i = tid; i < n16; i += nthreads * 4) {
            f32x4 a = __builtin_nontemporal_load((const GLOBAL f32x4*)buf + i);
            f32x4 b = i + nthreads < n16 ? __builtin_nontemporal_load((const GLOBAL f32x4*)buf + i + nthreads) : acc;
            f32x4 c = i + 2 * nthreads < n16 ? __builtin_nontemporal_load((const GLOBAL f32x4*)buf + i + 2 * nthreads) : acc;
            f32x4 d = i + 3 * nthreads < n16 ? __builtin_nontemporal_load((const GLOBAL f32x4*)buf + i + 3 * nthreads) : acc;
            acc += a + b + c + d;
        }
    if (acc.x == 12345.f) sink[0] = acc.x;
}

__global__ __launch_bounds__(64) void probe_kernel(const f32x4* rows, size_t nblocks, int form, int nprobe, uint32_t* lat, float* sink, uint64_t seed) {
    const int lane = threadIdx.x;
    uint64_t s = seed + blockIdx.x * 0x9E3779B97F4A7C15ull;
    float acc = 0.f;
    for (int it = 0; it < nprobe; ++it) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const size_t b = (size_t)((s >> 20) % nblocks);
        const int r = (int)((s >> 12) & 31);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        if (form == 1) {  // row-major: row (b*32 + r) of 96 pieces
            const f32x4* p = rows + ((size_t)b * 32 + r) * 96;
            v0 = *((const GLOBAL f32x4*)p + lane);
            if (lane < 32) v1 = *((const GLOBAL f32x4*)p + 64 + lane);
        } else {          // blocked: piece f4 of row r of block b at (b*96 + f4)*32 + r
            const f32x4* p = rows + (size_t)b * 96 * 32 + r;
            v0 = *((const GLOBAL f32x4*)p + (size_t)lane * 32);
            if (form == 0 && lane < 32) v1 = *((const GLOBAL f32x4*)p + (size_t)(64 + lane) * 32);
        }
        float part = v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w;
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        acc += part;
        const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) lat[(size_t)blockIdx.x * nprobe + it] = (uint32_t)(t1 - t0);
    }
    if (acc == 12345.f) sink[1] = acc;
}

int main() {
    const size_t stream_bytes = 38ull << 30, row_bytes = 150ull << 30;
    f32x4 *d_stream, *d_rows;
    float* d_sink;
    int* d_stop;
    CHECK(hipMalloc((void**)&d_stream, stream_bytes));
    CHECK(hipMalloc((void**)&d_rows, row_bytes));
    CHECK(hipMalloc((void**)&d_sink, 64));
    CHECK(hipMalloc((void**)&d_stop, 4));
    CHECK(hipMemset(d_stream, 0, stream_bytes));
    CHECK(hipMemset(d_rows, 0, row_bytes));
    CHECK(hipMemset(d_stop, 0, 4));
    const int nwaves = 256, nprobe = 200;
    uint32_t* d_lat;
    CHECK(hipMalloc((void**)&d_lat, (size_t)nwaves * nprobe * 4));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    const size_t nblocks = row_bytes / (96 * 32 * 16);
    std::vector<uint32_t> lat((size_t)nwaves * nprobe);
    for (int loaded = 0; loaded < 2; ++loaded)
        for (int form = 0; form < 3; ++form) {
            CHECK(hipMemset(d_stop, 0, 4));
            if (loaded) stream_kernel<<<768, 256, 0, s1>>>(d_stream, stream_bytes / 16, 1000, d_sink, d_stop);
            probe_kernel<<<nwaves, 64, 0, s2>>>(d_rows, nblocks, form, nprobe, d_lat, d_sink, 1234 + form);
            CHECK(hipStreamSynchronize(s2));
            const int one = 1;
            CHECK(hipMemcpyAsync(d_stop, &one, 4, hipMemcpyHostToDevice, s2));
            CHECK(hipStreamSynchronize(s2));
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipMemcpy(lat.data(), d_lat, lat.size() * 4, hipMemcpyDeviceToHost));
            std::sort(lat.begin(), lat.end());
            double mean = 0;
            for (uint32_t v : lat) mean += v;
            mean /= lat.size();
            std::printf("%s  form %d (%s): median %.2f us  mean %.2f us  p90 %.2f us per row (%d waves x %d rows)\n",
                        loaded ? "HBM streaming" : "idle         ", form,
                        form == 0 ? "blocked, 96 pieces" : form == 1 ? "row-major, 1536 B " : "blocked, 64 pieces",
                        lat[lat.size() / 2] / 100.0, mean / 100.0, lat[lat.size() * 9 / 10] / 100.0, nwaves, nprobe);
        }
    return 0;
}

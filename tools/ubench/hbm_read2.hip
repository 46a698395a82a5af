// Micro-benchmark: does the ADDRESS PATTERN of a streaming read matter on MI355X?  (ceiling search for the scan)
//   pattern 0: grid-stride, consecutive workgroups read consecutive 4 KB (what hbm_read.hip does)
//   pattern 1: blocked — each workgroup owns one contiguous region of the buffer
//   pattern 2: blocked per wave — each wave owns one contiguous region
//   pattern 3: XCD-blocked — the 1/8 of the buffer a workgroup reads is chosen by blockIdx % 8 (its XCD),
//              grid-stride inside it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n, float* out, int pattern) {
    f32x4 acc = {0, 0, 0, 0};
    size_t i, end, stride;
    const size_t nwg = gridDim.x, wg = blockIdx.x;
    if (pattern == 0) {
        i = wg * 256 + threadIdx.x; end = n; stride = nwg * 256;
    } else if (pattern == 1) {
        const size_t per = n / nwg;  // n is a multiple of everything here
        i = wg * per + threadIdx.x; end = (wg + 1) * per; stride = 256;
    } else if (pattern == 2) {
        const size_t per = n / (nwg * 4);
        const size_t w = wg * 4 + (threadIdx.x >> 6);
        i = w * per + (threadIdx.x & 63); end = (w + 1) * per; stride = 64;
    } else {
        const size_t per = n / 8, x = wg & 7, j = wg >> 3, nj = nwg >> 3;
        i = x * per + j * 256 + threadIdx.x; end = (x + 1) * per; stride = nj * 256;
    }
    for (; i + (UNROLL - 1) * stride < end; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < end; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)38654705664ull;  // 36 GiB: divisible by all grids
    f32x4* d; float* out;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(d, 0, bytes);
    const size_t n = bytes / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pattern = 0; pattern < 4; ++pattern)
        for (int w : {4, 8, 16}) {
            const int grid = 256 * w;
            rd<8><<<grid, 256>>>(d, n, out, pattern);
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                hipEventRecord(e0);
                rd<8><<<grid, 256>>>(d, n, out, pattern);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("pattern %d wg/cu %2d: %.3f ms  %.0f GB/s\n", pattern, w, best, bytes / (best * 1e-3) / 1e9);
        }
    return 0;
}

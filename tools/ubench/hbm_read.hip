// Micro-benchmark: the streaming-read ceiling of one MI355X — what a kernel that does nothing but load
// (non-temporal, 16 B per lane, several loads in flight) reaches; the scan kernel is judged against it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int UNROLL, bool NT>
void run(const f32x4* d, size_t n, float* out, int wg_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    rd<UNROLL, NT><<<grid, 256>>>(d, n, out);
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        rd<UNROLL, NT><<<grid, 256>>>(d, n, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("unroll %d %s wg/cu %2d: avg %.3f ms min %.3f ms  -> %.0f GB/s avg, %.0f GB/s best\n", UNROLL, NT ? "nt" : "  ", wg_per_cu, sum / 5,
           best, n * 16.0 / (sum / 5 * 1e-3) / 1e9, n * 16.0 / (best * 1e-3) / 1e9);
}
int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)38400000000ull;  // 25M x 384 f32 rows
    f32x4* d; float* out;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(d, 0, bytes);
    const size_t n = bytes / 16;
    for (int w : {4, 8, 16}) {
        run<4, true>(d, n, out, w);
        run<8, true>(d, n, out, w);
        run<8, false>(d, n, out, w);
    }
    return 0;
}

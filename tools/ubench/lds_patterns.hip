// Micro-benchmark: LDS throughput of the exact access patterns of gemm_bf16x3 (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDB = 40;  // bf16 elements per LDS row (80 B)

// pattern 0: A-plane writes (8 B per lane, 3 planes, 4 row groups)   1: W-plane writes (16 B per lane)
// pattern 2: fragment reads (16 B per lane)                         3: contiguous b128 writes  4: contiguous b64 writes
// pattern 5: A-plane writes, unpadded 64 B rows (row = lane>>3)      6: contiguous b128 reads
__global__ __launch_bounds__(256) void k(float* out, int iters, int pattern) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int pt = threadIdx.x, lane = pt & 63, wave = pt >> 6;
    const int ag_ = pt >> 3, ac4 = pt & 7;
    const int arow = 8 * ((ag_ >> 1) >> 2) + ((ag_ >> 1) & 3) + 4 * (ag_ & 1);
    const int wg_ = pt >> 3, wsub = pt & 7, wc8 = wsub & 3;
    const int wrow = (wg_ >> 2) * 8 + (wg_ & 3) + 4 * (wsub >> 2);
    const int i = lane & 31, h = lane >> 5;
    f32x2 v2 = {1.0f * pt, 2.0f};
    f32x4 v4 = {1.0f * pt, 2.0f, 3.0f, 4.0f};
    f32x4 acc = {0, 0, 0, 0};
    constexpr int PLANE = 128 * LDB * 2;  // bytes
    for (int it = 0; it < iters; ++it) {
        if (pattern == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *(f32x2*)(sm + pl * PLANE + ((arow + 32 * u) * LDB + ac4 * 4) * 2) = v2;
        } else if (pattern == 1) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *(f32x4*)(sm + pl * PLANE + (wrow * LDB + wc8 * 8) * 2) = v4;
                *(f32x4*)(sm + pl * PLANE + ((wrow + 64) * LDB + wc8 * 8) * 2) = v4;
            }
        } else if (pattern == 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        acc += *(const f32x4*)(sm + pl * PLANE + (((wave >> 1) * 64 + t * 32 + i) * LDB + s * 16 + 8 * h) * 2);
        } else if (pattern == 3) {
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(sm + (j * 256 + pt) * 16) = v4;
        } else if (pattern == 4) {
#pragma unroll
            for (int j = 0; j < 12; ++j) *(f32x2*)(sm + (j * 256 + pt) * 8) = v2;
        } else if (pattern == 5) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *(f32x2*)(sm + pl * 8192 + ((ag_ + 32 * u) * 32 + ac4 * 4) * 2) = v2;
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) acc += *(const f32x4*)(sm + ((j * 256 + pt) * 16 & 32767));
        }
        v2.x += 1.0f;
        v4.x += 1.0f;
    }
    __syncthreads();
    out[blockIdx.x * 256 + pt] = acc.x + acc.y + acc.z + acc.w + ((float*)sm)[pt];
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"A-plane writes b64 (padded 80 B rows)", "W-plane writes b128 (80 B rows)", "fragment reads b128 (80 B rows)",
                           "contiguous b128 writes", "contiguous b64 writes", "A-plane writes b64 (64 B rows, no pad)", "contiguous b128 reads"};
    const double bytes_per_iter[] = {4 * 3 * 8.0 * 256, 6 * 16.0 * 256, 12 * 16.0 * 256, 6 * 16.0 * 256, 12 * 8.0 * 256, 4 * 3 * 8.0 * 256, 12 * 16.0 * 256};
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (int p = 0; p < 7; ++p) {
            const int iters = 20000;
            k<<<256 * wgs_per_cu, 256, 65536>>>(d, 10, p);
            hipEventRecord(e0);
            k<<<256 * wgs_per_cu, 256, 65536>>>(d, iters, p);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%d WG/CU  %-42s %.3f ms  %.1f B/clk/CU (at 2.05 GHz)\n", wgs_per_cu, names[p], ms,
                   bytes_per_iter[p] * iters * wgs_per_cu / (ms * 1e-3) / 2.05e9);
        }
    return 0;
}

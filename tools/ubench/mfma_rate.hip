// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 / f32 32x32x2 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) acc[n][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[n][r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (KIND == 0) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
            if (KIND == 1) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[n], 0, 0, 0);
            if (KIND == 2) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) s += acc[n][r]; for (int r = 0; r < 4; ++r) s += acc4[n][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int KIND>
void run(const char* name, int waves_per_simd, double flop_per_mfma) {
    int cus = 256;
    float* d; hipMalloc(&d, 4 * 64 * cus * 4 * waves_per_simd);
    int iters = 20000;
    dim3 grid(cus * 4 * waves_per_simd / 4), block(256);  // 4 waves per WG
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, KIND><<<grid, block>>>(d, 100);
    hipEventRecord(e0);
    k<NACC, KIND><<<grid, block>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_mfma = (double)grid.x * 4 * iters * NACC;
    double per_simd = (double)iters * NACC * waves_per_simd;          // MFMAs executed by one SIMD
    printf("%-28s nacc %d waves/simd %d: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", name, NACC, waves_per_simd, ms,
           n_mfma * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd);
    hipFree(d);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>("32x32x16 bf16", w, 32768.0);
        run<2, 0>("32x32x16 bf16", w, 32768.0);
        run<4, 0>("32x32x16 bf16", w, 32768.0);
        run<4, 1>("16x16x32 bf16", w, 16384.0);
        run<8, 1>("16x16x32 bf16", w, 16384.0);
        run<4, 2>("32x32x2 f32", w, 4096.0);
    }
    return 0;
}

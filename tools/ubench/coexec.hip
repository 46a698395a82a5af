// Micro-benchmark: do MFMA (wave A) and VALU / LDS work (wave B) on the same SIMD overlap on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run VALU; bit2: waves 4-7 run LDS reads instead
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    __shared__ f32x4 lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    lds[threadIdx.x] = f32x4{1, 2, 3, 4};
    __syncthreads();
    float s = 0;
    if (wave < 4) {
        if (mode & 1) {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
            f32x16 acc[4];
            for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
            for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
        }
    } else if (mode & 2) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters; ++it)  // 32 VALU per iteration (= 128 cycles, same as 4 MFMAs)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
        for (int i = 0; i < 8; ++i) s += x[i];
    } else if (mode & 4) {
        f32x4 v = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {  // 12 ds_read_b128 per iteration
#pragma unroll
            for (int j = 0; j < 12; ++j) v += lds[(threadIdx.x + 64 * j + it) & 4095];
        }
        s = v.x + v.y + v.z + v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"", "MFMA only", "VALU only", "MFMA + VALU", "LDS only", "MFMA + LDS"};
    for (int mode : {1, 2, 3, 4, 5}) {
        k<<<256, 512>>>(d, 100, mode);
        hipEventRecord(e0);
        k<<<256, 512>>>(d, 20000, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %.3f ms\n", names[mode], ms);
    }
    return 0;
}

// Micro-benchmark: global -> LDS direct loads (global_load_lds_dwordx4, gfx950) against the register path
// (global_load_dwordx4 + ds_write_b128) for an L2-resident source.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: DMA, mode 1: load + ds_write
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ src, float* out, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, wave = tid >> 6;
    // each iteration a workgroup brings 24 KB (6 x 4 KB) into LDS, like the W planes of one K step
    const f32x4* base = src + (size_t)(blockIdx.x & 63) * 1536 + tid;  // 64 distinct 24 KB tiles: L2 resident
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + j * 256),
                                                 (__attribute__((address_space(3))) void*)(sm + j * 4096 + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0);
        } else {
            f32x4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = base[j * 256];
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(sm + j * 4096 + tid * 16) = v[j];
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = ((float*)sm)[tid];
}
int main() {
    f32x4* d; float* out;
    hipMalloc(&d, 64 * 24576); hipMemset(d, 0, 64 * 24576);
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {1, 2, 3})
        for (int mode : {0, 1}) {
            const int iters = 20000;
            k<<<256 * wgs, 256, 32768>>>(d, out, 10, mode);
            hipEventRecord(e0);
            k<<<256 * wgs, 256, 32768>>>(d, out, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%d WG/CU %-22s %.3f ms  %.1f B/clk/CU (at 2.05 GHz)\n", wgs, mode == 0 ? "global_load_lds b128" : "load + ds_write_b128", ms,
                   24576.0 * iters * wgs / (ms * 1e-3) / 2.05e9);
        }
    return 0;
}

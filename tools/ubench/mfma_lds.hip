// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate with the operand fragments coming from LDS as in the encoder's GEMM loops:
// per K step of 32 a wave reads its fragments (RD x ds_read_b128) and issues 16 * NA * NB multiplies on NA x NB accumulators.
// Prints TFLOP/s against the 157.3 TF peak for 1..4 waves per SIMD, with and without the LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int NB, bool LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float tile[256 * 36];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    for (int t = threadIdx.x; t < 256 * 36; t += 256) tile[t] = (float)(t & 15) * 0.125f;
    __syncthreads();
    f32x16 acc[NA][NB];
    for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int aoff = ((wave & 1) * 64 + i) * 36 + 16 * kk, boff = (128 + (wave >> 1) * 64 + i) * 36 + 16 * kk;
    float af[NA][16], bf[NB][16];
    for (int a = 0; a < NA; ++a) for (int s = 0; s < 16; ++s) af[a][s] = 1.0f + s;
    for (int b = 0; b < NB; ++b) for (int s = 0; s < 16; ++s) bf[b][s] = 0.5f + s;
    for (int it = 0; it < iters; ++it) {
        if (LDSR) {
            asm volatile("" : "+v"(aoff), "+v"(boff));  // (the reads stay inside the loop)
            const float* ap = tile + aoff;
            const float* bp = tile + boff;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 x = *(const f32x4*)(ap + a * 32 * 36 + 4 * v);
                    af[a][4 * v] = x.x; af[a][4 * v + 1] = x.y; af[a][4 * v + 2] = x.z; af[a][4 * v + 3] = x.w;
                }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 y = *(const f32x4*)(bp + b * 32 * 36 + 4 * v);
                    bf[b][4 * v] = y.x; bf[b][4 * v + 1] = y.y; bf[b][4 * v + 2] = y.z; bf[b][4 * v + 3] = y.w;
                }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
    }
    float sum = 0;
    for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int NA, int NB, bool LDSR>
void run(int wgs_per_cu) {
    const int cus = 256, iters = 300;
    float* d; hipMalloc(&d, (size_t)cus * wgs_per_cu * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NA, NB, LDSR><<<cus * wgs_per_cu, 256>>>(d, 50);
    hipEventRecord(e0);
    k<NA, NB, LDSR><<<cus * wgs_per_cu, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * wgs_per_cu * 4 * iters * 16.0 * NA * NB * 4096.0;
    printf("%dx%d accumulators, %s, %d waves per SIMD: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n", NA, NB, LDSR ? "fragments from LDS" : "no LDS reads      ",
           wgs_per_cu, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(d);
}
template <int NA, int NB, bool LDSR>
void regs() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, (const void*)k<NA, NB, LDSR>);
    printf("%dx%d %s: %d registers\n", NA, NB, LDSR ? "lds" : "no lds", at.numRegs);
}
int main() {
    regs<2, 2, false>(); regs<2, 2, true>(); regs<1, 3, true>(); regs<1, 2, true>(); regs<1, 4, true>(); regs<1, 1, true>();
    for (int w = 1; w <= 4; ++w) {
        run<2, 2, false>(w);
        run<2, 2, true>(w);
    }
    for (int w = 1; w <= 5; ++w) run<1, 3, true>(w);
    for (int w = 2; w <= 6; ++w) run<1, 2, true>(w);
    for (int w = 2; w <= 4; ++w) run<1, 4, true>(w);
    for (int w = 3; w <= 8; ++w) run<1, 1, true>(w);
    return 0;
}

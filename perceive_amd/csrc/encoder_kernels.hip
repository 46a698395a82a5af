// encoder_kernels.hip — gfx950 kernels of the sentence-embedding forward pass (replaces the
// libtorch forward behind model/worker.rs:78-106: BERT-family encoder, pooling, normalisation).
//
// Numerics: f32 end to end like the reference (tch default dtype, no autocast).  Every contraction
// runs on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain in k
// order), so the result differs from a CPU f32 evaluation only by summation order.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "encoder.h"
#include "synth.h"

// Timing experiments (tools/exp_build_enc.sh <n> on the GPU box; results are WRONG in these builds): pieces of gemm_f32_kernel
// taken out to see what the rest costs — 10: no epilogue (one value per lane stored), 11: no operand loads in the K loop,
// 12: no loads, no staging writes and no barriers either (fragment reads + multiplies only).  0 = the product.
#ifndef PCV_ENC_EXP
#define PCV_ENC_EXP 0
#endif

namespace pcv {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D-layout of a 32x32 accumulator: register r of lane (c = lane&31, h = lane>>5) is element
// [row = (r&3) + 8*(r>>2) + 4*h][col = c].
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------
// GEMM: C[M][N] = A[M][K] W[N][K]^T + bias (+GELU | +residual)
// 128x128 output tile per 256-thread workgroup, K in steps of 32 through LDS (rows padded to 36
// floats: the 16-lane ds_read_b128 groups then cover all 64 banks), 4 waves as 2x2, each wave
// 2x2 MFMA tiles of 32x32.  The k index inside a 32-wide step is assigned as k = 16*(lane>>5) + s
// for both operands, so a lane's 16 operand values are 64 contiguous bytes (4 x ds_read_b128).
// Four workgroups per CU (round 4; three before): the operand loads go through buffer descriptors — one address register for
// the eight loads of a step instead of sixteen — and the kernel fits 122 registers (256 x 256 tokens, same box: QKV
// projection 463 -> 447 us, FFN up 641 -> 600 us).
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

// Output tile of a workgroup.  Workgroups are dealt round-robin over the 8 XCDs, each with its own
// L2: with the natural order the N/128 workgroups that share an A row-panel land on different XCDs
// and every one of them pulls the panel from HBM again (3-8x the activation traffic; the
// split-precision kernel was bound by exactly that).  The remap gives each XCD a contiguous run of
// tiles in row-major order, so a row panel is fetched once per XCD and reused from its L2.
__device__ __forceinline__ void tile_of_block(int ncols, int& tile_m, int& tile_n) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    unsigned logical = wg;
    if ((nwg & 7u) == 0) logical = (wg & 7u) * (nwg >> 3) + (wg >> 3);
    tile_m = (int)(logical / (unsigned)ncols);
    tile_n = (int)(logical % (unsigned)ncols);
}

// erf for the GELU epilogues: Abramowitz & Stegun 7.1.28, erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16 for
// x >= 0, |error| <= 3e-7 — an error of <= 1e-6 in a GELU output, two orders below the f32 noise of the
// 1536-term contraction that consumes it.  Branch-free, 14 VALU operations; erff() is ~30 with two divergent
// paths, and VALU work in an epilogue is paid in matrix-pipe time (tools/ubench/coexec.hip): it was 15 % of the
// FFN-up GEMM.
__device__ __forceinline__ float erf_as(float x) {
    const float t = fabsf(x);
    float p = fmaf(t, 0.0000430638f, 0.0002765672f);
    p = fmaf(t, p, 0.0001520143f);
    p = fmaf(t, p, 0.0092705272f);
    p = fmaf(t, p, 0.0422820123f);
    p = fmaf(t, p, 0.0705230784f);
    p = fmaf(t, p, 1.0f);
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;  // (..)^16; +inf for large |x|, whose reciprocal is 0
    const float e = 1.0f - __builtin_amdgcn_rcpf(p);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }
// "gelu_new" of ALBERT (the tanh form): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), written as x * sigmoid(2u)
// so that one exp serves and large |x| saturate to x / 0 without overflow
__device__ __forceinline__ float gelu_tanh(float v) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return v / (1.0f + __expf(-2.0f * u));
}
// The same two functions on PAIRS of values: v_pk_fma_f32 / v_pk_mul_f32 do two lanes' worth per instruction, and VALU time in an
// epilogue is matrix-pipe time (the scalar form was 990 VALU instructions per thread and 128x128 tile, 640 of them fma / mul).
// Element for element the same operations in the same order as the scalar forms: identical results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 x = v * splat2(0.70710678118654752440f);
    const f32x2 t = __builtin_elementwise_abs(x);
    f32x2 p = __builtin_elementwise_fma(t, splat2(0.0000430638f), splat2(0.0002765672f));
    p = __builtin_elementwise_fma(t, p, splat2(0.0001520143f));
    p = __builtin_elementwise_fma(t, p, splat2(0.0092705272f));
    p = __builtin_elementwise_fma(t, p, splat2(0.0422820123f));
    p = __builtin_elementwise_fma(t, p, splat2(0.0705230784f));
    p = __builtin_elementwise_fma(t, p, splat2(1.0f));
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const f32x2 e = splat2(1.0f) - f32x2{__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
    return (splat2(0.5f) * v) * (splat2(1.0f) + __builtin_elementwise_copysign(e, x));
}
template <int EPI>
__device__ __forceinline__ f32x4 epi_act4(f32x4 v) {
    if (EPI == EPI_BIAS_GELU) {
        const f32x2 lo = gelu_erf2(f32x2{v.x, v.y}), hi = gelu_erf2(f32x2{v.z, v.w});
        return f32x4{lo.x, lo.y, hi.x, hi.y};
    }
    if (EPI == EPI_BIAS_GELU_TANH) return f32x4{gelu_tanh(v.x), gelu_tanh(v.y), gelu_tanh(v.z), gelu_tanh(v.w)};
    return v;
}
template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
    if (EPI == EPI_BIAS_GELU) return gelu_erf(v);
    if (EPI == EPI_BIAS_GELU_TANH) return gelu_tanh(v);
    return v;
}

// Epilogue of the 128x128 GEMMs: one wave's 64x64 quarter (2x2 MFMA tiles) -> +bias (+GELU | +residual) ->
// C.  Full quarters take a branch-free path.  With a per-element `if (row < M)` every element became its
// own basic block, the waitcnt pass lost track of the bias load across them and put `s_waitcnt vmcnt(0)` in
// front of every single store (stores count in vmcnt on gfx9): 64 serialised HBM round trips per wave,
// ~5 ms of the 13 ms encoder forward.
template <int EPI>
__device__ __forceinline__ void store_quarter(const f32x16 (&acc)[2][2], const float (&bv)[2],
                                              const float* __restrict__ resid, float* __restrict__ C, int M, int N,
                                              int row0, int col0, int i, int h) {
    if (row0 + 64 <= M) {  // wave-uniform
        const size_t base0 = (size_t)(row0 + 4 * h) * N + col0 + i;
        float rv[2][2][16];
        if (EPI == EPI_BIAS_RESIDUAL) {  // all 64 residual loads in flight together: one latency, not four
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[a][b][r] = resid[base0 + (size_t)(a * 32 + acc_row(r, 0)) * N + b * 32];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[a][b][r] + bv[b];
                    v = epi_act<EPI>(v);
                    if (EPI == EPI_BIAS_RESIDUAL) v += rv[a][b][r];
                    C[base0 + (size_t)(a * 32 + acc_row(r, 0)) * N + b * 32] = v;
                }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = col0 + b * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + a * 32 + acc_row(r, h);
                if (row < M) {
                    float v = acc[a][b][r] + bv[b];
                    v = epi_act<EPI>(v);
                    if (EPI == EPI_BIAS_RESIDUAL) v += resid[(size_t)row * N + col];
                    C[(size_t)row * N + col] = v;
                }
            }
        }
}

// Epilogue of gemm_f32_kernel for a full 64x64 quarter: the accumulators (column on the lane, rows in the
// registers) go through a per-wave LDS tile, half a quarter at a time, and come back row-major, so that bias,
// residual and the output move as 16-byte accesses: 16 stores (and 16 residual loads) per lane instead of 64.
constexpr int LDE = 68;  // floats per row of the [32][64] transpose tile: 16-lane ds_read_b128 groups cover a bank row
template <int EPI>
__device__ __forceinline__ void store_quarter_wide(const f32x16 (&acc)[2][2], const float* __restrict__ bias,
                                                   const float* __restrict__ resid, float* __restrict__ C, int N, int row0,
                                                   int col0, float* __restrict__ tile, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const int c4 = lane & 15, rq = lane >> 4;  // read side: float4 column, row = rq + 4*u
    const f32x4 b4 = bias ? *(const f32x4*)(bias + col0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        f32x4 rv[8];
        if (EPI == EPI_BIAS_RESIDUAL) {  // in flight while the half tile goes through LDS
#pragma unroll
            for (int u = 0; u < 8; ++u)
                rv[u] = *(const f32x4*)(resid + (size_t)(row0 + a * 32 + rq + 4 * u) * N + col0 + 4 * c4);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[acc_row(r, h) * LDE + b * 32 + i] = acc[a][b][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's tile is written (no other wave touches it)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4 v = *(const f32x4*)(tile + (rq + 4 * u) * LDE + 4 * c4) + b4;
            if (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_TANH) v = epi_act4<EPI>(v);
            if (EPI == EPI_BIAS_RESIDUAL) v += rv[u];
            *(f32x4*)(C + (size_t)(row0 + a * 32 + rq + 4 * u) * N + col0 + 4 * c4) = v;
        }
        __builtin_amdgcn_wave_barrier();  // the reads are issued before the next half overwrites the tile
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// the bias values of a wave's two column tiles
__device__ __forceinline__ void load_bias2(const float* __restrict__ bias, int col0, int i, float (&bv)[2]) {
#pragma unroll
    for (int b = 0; b < 2; ++b) bv[b] = bias ? bias[col0 + b * 32 + i] : 0.0f;
}

template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ C, int M,
                                                       int N, int K) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];  // K-step tiles; the epilogue's transpose tiles after the loop
    float* As = smem;
    float* Ws = smem + BM * LDT;
    static_assert(4 * 32 * LDE <= (BM + BN) * LDT, "epilogue tiles fit the staging buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    int tile_m, tile_n;
    tile_of_block(N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // staging: thread t moves float4 (row = (t>>3) + 32*u, 16-byte column c4 = t&7), u = 0..3.  Through buffer descriptors with
    // the tile's base: one address register for all eight loads of a step (the rest of the address is scalar), and the rows of
    // A past M read as zeros
    const int srow = tid >> 3, c4 = tid & 7;
    const uint32_t lane_off = (uint32_t)(srow * K + c4 * 4) * 4u;
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, (uint32_t)(min(M - m0, BM) * K) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, (uint32_t)(BN * K) * 4u, 0x00020000);
    f32x4 ra[4], rw[4];
    auto fetch = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, lane_off, (uint32_t)(32 * u * K + kt * BK) * 4u, 0));
            rw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, lane_off, (uint32_t)(32 * u * K + kt * BK) * 4u, 0));
        }
    };
    fetch(0);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#if PCV_ENC_EXP == 12
        if (kt == 0)
#endif
        {
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *(f32x4*)&As[(srow + 32 * u) * LDT + c4 * 4] = ra[u];
            *(f32x4*)&Ws[(srow + 32 * u) * LDT + c4 * 4] = rw[u];
        }
        __syncthreads();
        }
        // next step's global loads fly under this step's MFMAs (the last step re-reads its own tile instead of branching: a
        // conditional load made the compiler park the staging registers in scratch).  The scheduling barrier keeps them HERE:
        // left alone, the scheduler sinks the loads below the multiplies to the top of the next step — right in front of the
        // LDS writes that wait for them — to save their registers, and the prefetch is gone (round 4)
#if PCV_ENC_EXP == 11 || PCV_ENC_EXP == 12
        if (kt == 1000)
#endif
        fetch(min(kt + 1, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        float af[2][16], bf[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* ap = &As[(wr * 64 + t * 32 + i) * LDT + 16 * kk];
            const float* bp = &Ws[(wc * 64 + t * 32 + i) * LDT + 16 * kk];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = *(const f32x4*)(ap + 4 * v);
                const f32x4 y = *(const f32x4*)(bp + 4 * v);
                af[t][4 * v + 0] = x.x; af[t][4 * v + 1] = x.y; af[t][4 * v + 2] = x.z; af[t][4 * v + 3] = x.w;
                bf[t][4 * v + 0] = y.x; bf[t][4 * v + 1] = y.y; bf[t][4 * v + 2] = y.z; bf[t][4 * v + 3] = y.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
    }

#if PCV_ENC_EXP == 10
    {
        float keep = 0.0f;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[a2][b2][r];
        C[(size_t)(m0 + (tid >> 1)) * N + n0 + (tid & 1) * 64 + lane] = keep;
        return;
    }
#endif
    if (m0 + BM <= M) {  // full tile (workgroup-uniform): wide epilogue through LDS
        __syncthreads();  // every wave is done with the last K-step's fragments
        store_quarter_wide<EPI>(acc, bias, resid, C, N, m0 + wr * 64, n0 + wc * 64, smem + wave * 32 * LDE, lane);
        return;
    }
    float bv[2];
    load_bias2(bias, n0 + wc * 64, i, bv);
    store_quarter<EPI>(acc, bv, resid, C, M, N, m0 + wr * 64, n0 + wc * 64, i, kk);
}

// The same GEMM on 128 x 96 output tiles, + bias (+ GELU), for grids that the 128 x 128 tiling deals unevenly:
// at 8 192 tokens N = 1 152 gives 576 tiles — 2.25 per CU, so the call takes as long as the CUs with three — but 768 tiles of
// 96 columns, three per CU, each 0.75 of the work (round 4).  Four waves, each 32 rows x 96 columns (1 x 3 MFMA tiles: one A
// fragment serves three multiplies); staging, prefetch and the K loop as above; the epilogue takes the wave's tile through LDS in
// two halves of 16 rows and stores rows as 16-byte pieces.
constexpr int BN96 = 96, LDE96 = 100;
template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_f32_n96_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                              int K) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN96) * LDT];
    float* As = smem;
    float* Ws = smem + BM * LDT;
    static_assert(4 * 16 * LDE96 <= (BM + BN96) * LDT, "epilogue tiles fit the staging buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    int tile_m, tile_n;
    tile_of_block(N / BN96, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN96;
    const int srow = tid >> 3, c4 = tid & 7;
    const uint32_t lane_off = (uint32_t)(srow * K + c4 * 4) * 4u;
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, (uint32_t)(min(M - m0, BM) * K) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, (uint32_t)(BN96 * K) * 4u, 0x00020000);
    f32x4 ra[4], rw[3];
    auto fetch = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, lane_off, (uint32_t)(32 * u * K + kt * BK) * 4u, 0));
#pragma unroll
        for (int u = 0; u < 3; ++u) rw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, lane_off, (uint32_t)(32 * u * K + kt * BK) * 4u, 0));
    };
    fetch(0);
    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    const float* ap = &As[(wave * 32 + i) * LDT + 16 * kk];
    const float* bp = &Ws[i * LDT + 16 * kk];
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f32x4*)&As[(srow + 32 * u) * LDT + c4 * 4] = ra[u];
#pragma unroll
        for (int u = 0; u < 3; ++u) *(f32x4*)&Ws[(srow + 32 * u) * LDT + c4 * 4] = rw[u];
        __syncthreads();
        fetch(min(kt + 1, nk - 1));
        __builtin_amdgcn_sched_barrier(0);  // (see gemm_f32_kernel)
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const f32x4 x = *(const f32x4*)(ap + 4 * hh);
            f32x4 y[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) y[c] = *(const f32x4*)(bp + c * 32 * LDT + 4 * hh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s], y[c][s], acc[c], 0, 0, 0);
        }
    }
    const int row0 = m0 + wave * 32;
    if (m0 + BM <= M) {  // full tile (workgroup-uniform)
        __syncthreads();  // every wave is done with the last K step's fragments
        float* T = smem + wave * 16 * LDE96;
        const int row8 = lane >> 3, l8 = lane & 7;
        f32x4 b4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) b4[j] = bias ? *(const f32x4*)(bias + n0 + 4 * (l8 + 8 * j)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 8; ++r) T[(acc_row(8 * half + r, kk) - 16 * half) * LDE96 + c * 32 + i] = acc[c][8 * half + r];
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's tile is written (no other wave touches it)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = row8 + 8 * p;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    *(f32x4*)(C + (size_t)(row0 + 16 * half + row) * N + n0 + 4 * (l8 + 8 * j)) = epi_act4<EPI>(*(const f32x4*)(T + row * LDE96 + 4 * (l8 + 8 * j)) + b4[j]);
            }
            __builtin_amdgcn_wave_barrier();  // the reads are issued before the next half overwrites the tile
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = n0 + c * 32 + i;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + acc_row(r, kk);
            if (row < M) C[(size_t)row * N + col] = epi_act<EPI>(acc[c][r] + bv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out = LayerNorm(A W^T + bias + resid) for N = hidden <= 384 (all-MiniLM widths): the attention-output
// and FFN-down projections with their residual + LayerNorm in the epilogue.  A workgroup owns 64 whole rows
// (64 x N output tile, the four waves side by side, CT = N/128 column tiles of 32 each per wave), so a row's
// mean and variance never leave the workgroup; the standalone LayerNorm pass — at the HBM roofline already,
// 100 MB in + 100 MB out per call at 256 x 256 tokens — disappears.  After the K loop the tile goes through
// LDS (the staging buffers, half the rows at a time) and comes back row-major, eight lanes per row: bias,
// residual and the normalised output move as 16-byte accesses and the two row reductions are three
// shuffles each.  Same arithmetic as layer_norm_fixed_kernel (mean, then sum of squared deviations).
// ------------------------------------------------------------------------------------------------
constexpr int LBM = 64;
template <int CT>
__global__ __launch_bounds__(256, 2) void gemm_f32_ln_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             const float* __restrict__ bias, const float* __restrict__ resid,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             float eps, float* __restrict__ C, int M, int K) {
    constexpr int N = CT * 128;
    constexpr int LDR = N + 4;  // floats per row of the row-major half tile
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float* As = lsm;              // [64][LDT]
    float* Ws = lsm + LBM * LDT;  // [N][LDT]
    static_assert(32 * LDR <= (LBM + N) * LDT, "the epilogue's half tile fits the staging buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int m0 = blockIdx.x * LBM;

    // staging: thread t moves float4 (row = (t>>3) + 32*u, 16-byte column c4 = t&7)
    const int srow = tid >> 3, c4 = tid & 7;
    const float* ag[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) ag[u] = A + (size_t)min(m0 + srow + 32 * u, M - 1) * K + c4 * 4;
    const float* wg = W + (size_t)srow * K + c4 * 4;  // rows srow + 32*u, u = 0 .. 4*CT-1
    f32x4 ra[2], rw[4 * CT];
#pragma unroll
    for (int u = 0; u < 2; ++u) ra[u] = *(const f32x4*)(ag[u]);
#pragma unroll
    for (int u = 0; u < 4 * CT; ++u) rw[u] = *(const f32x4*)(wg + (size_t)32 * u * K);

    f32x16 acc[2][CT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;

    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int u = 0; u < 2; ++u) *(f32x4*)&As[(srow + 32 * u) * LDT + c4 * 4] = ra[u];
#pragma unroll
        for (int u = 0; u < 4 * CT; ++u) *(f32x4*)&Ws[(srow + 32 * u) * LDT + c4 * 4] = rw[u];
        __syncthreads();
        {  // next step's global loads fly under this step's MFMAs (the last step re-reads its own tile)
            const size_t koff = (size_t)min(kt + 1, nk - 1) * BK;
#pragma unroll
            for (int u = 0; u < 2; ++u) ra[u] = *(const f32x4*)(ag[u] + koff);
#pragma unroll
            for (int u = 0; u < 4 * CT; ++u) rw[u] = *(const f32x4*)(wg + (size_t)32 * u * K + koff);
        }
        // fragments in two halves of eight k each: 40 registers of operands instead of 80
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float af[2][8], bf[CT][8];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* ap = &As[(a * 32 + i) * LDT + 16 * kk + 8 * hh];
                const f32x4 x0 = *(const f32x4*)ap, x1 = *(const f32x4*)(ap + 4);
                af[a][0] = x0.x; af[a][1] = x0.y; af[a][2] = x0.z; af[a][3] = x0.w;
                af[a][4] = x1.x; af[a][5] = x1.y; af[a][6] = x1.z; af[a][7] = x1.w;
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float* bp = &Ws[((wave * CT + c) * 32 + i) * LDT + 16 * kk + 8 * hh];
                const f32x4 y0 = *(const f32x4*)bp, y1 = *(const f32x4*)(bp + 4);
                bf[c][0] = y0.x; bf[c][1] = y0.y; bf[c][2] = y0.z; bf[c][3] = y0.w;
                bf[c][4] = y1.x; bf[c][5] = y1.y; bf[c][6] = y1.z; bf[c][7] = y1.w;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[c][s], acc[a][c], 0, 0, 0);
        }
    }

    // epilogue: rows 0..31 then 32..63 of the tile, row-major through LDS; 8 lanes per row, lane j of a row
    // holds its float4 columns j, j+8, ... (N/32 of them)
    constexpr int Q = N / 32;
    const int erow = tid >> 3, ej = tid & 7;
    float* tile = lsm;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        __syncthreads();  // fragment reads (a = 0) / the previous half's row reads (a = 1) are done
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[acc_row(r, kk) * LDR + (wave * CT + c) * 32 + i] = acc[a][c][r];
        __syncthreads();
        const int row = m0 + a * 32 + erow;
        const bool live = row < M;
        const float* rrow = resid + (size_t)(live ? row : 0) * N;
        f32x4 v[Q];
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const int col = 4 * (ej + 8 * t);
            v[t] = *(const f32x4*)(tile + erow * LDR + col) + *(const f32x4*)(bias + col) + *(const f32x4*)(rrow + col);
            sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        const float mean = sum / (float)N;
        float sq = 0.0f;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const f32x4 d = v[t] - mean;
            sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        sq += __shfl_xor(sq, 1);
        sq += __shfl_xor(sq, 2);
        sq += __shfl_xor(sq, 4);
        const float rstd = 1.0f / sqrtf(sq / (float)N + eps);
        if (live) {
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                const int col = 4 * (ej + 8 * t);
                *(f32x4*)(C + (size_t)row * N + col) = (v[t] - mean) * rstd * *(const f32x4*)(ln_w + col) + *(const f32x4*)(ln_b + col);
            }
        }
    }
}

// The same tile, persistent and with eight waves (two row halves x four column quarters, 32 x N/4 per wave).
// The workgroups of one launch move in step, so in the form above all K loops and then all epilogues
// coincide and the epilogue's HBM time (residual in, normalised rows out) is exposed.  Here a workgroup
// walks its tiles: the last K step of a tile already loads the next tile's first operands and the residual
// rows of the epilogue, and the epilogue's stores drain under the next tile's MFMAs.
template <int CT>
__global__ __launch_bounds__(512, 4) void gemm_f32_ln8_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* __restrict__ resid,
                                                              const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                              float eps, float* __restrict__ C, int M, int K) {
    constexpr int N = CT * 128;
    constexpr int LDR = N + 4;
    constexpr int Q = N / 64;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float* As = lsm;              // [64][LDT]
    float* Ws = lsm + LBM * LDT;  // [N][LDT]
    // bias, LayerNorm weight and bias: in LDS behind the staging buffers.  Read from global memory in the epilogue they cost it
    // most of its time: every `load weight, load bias, store` of the six per row half waits with vmcnt(0), and stores count in
    // vmcnt — each store waited for the one before it to be acknowledged by memory (22-26 us per 64-row tile, round 4)
    float* prm = lsm + (LBM + N) * LDT;  // [3][N]
    static_assert(32 * LDR <= (LBM + N) * LDT, "the epilogue's half tile fits the staging buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int rh = wave & 1, wc = wave >> 1;
    const int srow = tid >> 3, c4 = tid & 7;   // staging: float4 (row srow [+ 64u], 16-byte column c4)
    const int erow = tid >> 4, ej = tid & 15;  // epilogue: 16 lanes per row, float4 columns ej + 16t
    const int tiles = (M + LBM - 1) / LBM;
    const int nk = K / BK;
    int tile = blockIdx.x;
    if (tile >= tiles) return;

    // operands through buffer descriptors (one address register for the seven loads of a step; rows of A past M read as zeros)
    const uint32_t lane_off = (uint32_t)(srow * K + c4 * 4) * 4u;
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (uint32_t)(N * K) * 4u, 0x00020000);
    auto a_rsrc = [&](int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)t * LBM * K), 0, (uint32_t)(min(M - t * LBM, LBM) * K) * 4u, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t ares = a_rsrc(tile);
    f32x4 ra, rw[2 * CT];
    auto fetch = [&](int kt) __attribute__((always_inline)) {
        ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, lane_off, (uint32_t)(kt * BK) * 4u, 0));
#pragma unroll
        for (int u = 0; u < 2 * CT; ++u)
            rw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, lane_off, (uint32_t)(64 * u * K + kt * BK) * 4u, 0));
    };
    fetch(0);
    if (tid < 3 * N / 4) {
        const float* src = tid < N / 4 ? bias : tid < N / 2 ? ln_w : ln_b;
        *(f32x4*)(prm + 4 * tid) = *(const f32x4*)(src + 4 * (tid % (N / 4)));
    }  // (visible after the first step's barriers)
    const float* ap0 = &As[(rh * 32 + i) * LDT + 16 * kk];
    const float* bp0 = &Ws[(wc * CT * 32 + i) * LDT + 16 * kk];
    float* tile_lds = lsm;
    const float* trow = tile_lds + erow * LDR + 4 * ej;  // the epilogue's row piece and parameters: one address each + constant offsets
    const float* pcol = prm + 4 * ej;

    for (; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * LBM;
        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
        auto to_lds = [&]() {
            __syncthreads();  // the previous step's fragment reads / the previous tile's row reads are done
            *(f32x4*)&As[srow * LDT + c4 * 4] = ra;
#pragma unroll
            for (int u = 0; u < 2 * CT; ++u) *(f32x4*)&Ws[(srow + 64 * u) * LDT + c4 * 4] = rw[u];
            __syncthreads();
        };
        auto mfma_step = [&]() {
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {  // four k per lane half at a time
                const f32x4 x = *(const f32x4*)(ap0 + 4 * hh);
                f32x4 y[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) y[c] = *(const f32x4*)(bp0 + c * 32 * LDT + 4 * hh);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s], y[c][s], acc[c], 0, 0, 0);
            }
        };
        for (int kt = 0; kt + 1 < nk; ++kt) {
            to_lds();
            fetch(kt + 1);  // the next step's operands fly under this step's MFMAs
            __builtin_amdgcn_sched_barrier(0);  // (issued HERE: see gemm_f32_kernel)
            mfma_step();
        }
        to_lds();
        f32x4 rv[Q];  // under the last step: the residual rows of the epilogue's first half
        {
            const float* rrow = resid + (size_t)min(m0 + erow, M - 1) * N;
#pragma unroll
            for (int t = 0; t < Q; ++t) rv[t] = *(const f32x4*)(rrow + 4 * (ej + 16 * t));
        }
        mfma_step();

        // epilogue: rows 0..31 (from the rh = 0 waves) then 32..63, row-major through LDS
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            __syncthreads();
            if (rh == a) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tile_lds[acc_row(r, kk) * LDR + (wc * CT + c) * 32 + i] = acc[c][r];
            }
            __syncthreads();
            const int row = m0 + a * 32 + erow;
            f32x4 v[Q];
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                const int col = 4 * (ej + 16 * t);
                v[t] = *(const f32x4*)(trow + 64 * t) + *(const f32x4*)(pcol + 64 * t) + rv[t];
                sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
            }
            if (a == 0) {  // the second half's residual rows fly under this half's arithmetic and stores
                const float* rrow = resid + (size_t)min(row + 32, M - 1) * N;
#pragma unroll
                for (int t = 0; t < Q; ++t) rv[t] = *(const f32x4*)(rrow + 4 * (ej + 16 * t));
                __builtin_amdgcn_sched_barrier(0);  // issued BEFORE this half's stores: waiting for them then leaves the stores in flight
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 8);
            const float mean = sum / (float)N;
            float sq = 0.0f;
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                const f32x4 d = v[t] - mean;
                sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
            sq += __shfl_xor(sq, 1);
            sq += __shfl_xor(sq, 2);
            sq += __shfl_xor(sq, 4);
            sq += __shfl_xor(sq, 8);
            const float rstd = 1.0f / sqrtf(sq / (float)N + eps);
            if (row < M) {
#pragma unroll
                for (int t = 0; t < Q; ++t) {
                    const int col = 4 * (ej + 16 * t);
                    *(f32x4*)(C + (size_t)row * N + col) = (v[t] - mean) * rstd * *(const f32x4*)(pcol + N + 64 * t) + *(const f32x4*)(pcol + 2 * N + 64 * t);
                }
            }
        }
        // the next tile's first operands
        ares = a_rsrc(min(tile + (int)gridDim.x, tiles - 1));
        fetch(0);
    }
}

// The same fusion for FEW tokens (round 4): 8 192 tokens — one rank's share of BASELINE configs[4]'s 256 documents — are 128
// tiles of 64 rows: half the chip ran one eight-wave workgroup per CU and the other half nothing (gemm_f32_ln8_kernel: 119 us per
// call on average, 0.32 of the f32-MFMA peak; profiles/r04_encoder_32x256_before_kernel_stats.csv).  Here a workgroup owns 32
// whole rows (256 tiles: every CU has one) and its eight waves are two halves of K x four column quarters (32 x N/4 per wave
// and K half), two waves per SIMD.  The two K halves meet in LDS in the epilogue (two row-major 32 x N tiles), which adds them
// with bias and residual.  Same v_mfma_f32_32x32x2_f32 arithmetic; the sum over K is formed as two partial sums.
//
// There is NO barrier in the K loop: no W value is used by two waves of this workgroup, so each wave stages its own operands —
// its 32 x CT W rows and the 32 A rows over ITS half of K, 32 floats of them a step — through a private region of LDS (128 rows
// x 36 floats = 18 KB per wave, 147 KB per workgroup): loads to registers a step ahead, registers to LDS, fragments back, all
// inside one wave, where the LDS queue keeps the order.  The four waves of a K half each fetch the A rows themselves (L1 hits
// after the first): 16 load instructions a step instead of 13.
// What it took, 32 x 256 tokens, FFN-down call (K = 1536) / attention-output call (K = 384), same box:
//   113 / 32.6 us  the first form: workgroup-wide staging of 64-wide K steps, two barriers a step
//   104 / 32.8 us  wave-private staging, loads written as pointer arithmetic: the scheduler SANK the sixteen loads of the next
//                  step below the step's 48 multiplies, right in front of the LDS writes that wait for them (it saves the staging
//                  registers for the fragments that way) — the prefetch was gone, every step paid the whole L2 latency.  With the
//                  loads taken out 87 us, with the multiplies taken out 42 us, with neither staging nor loads 80 us: nothing overlapped
//    93 / 29.4 us  loads through buffer descriptors (one address register for all sixteen, scalar offsets: 174 registers
//                  instead of 256 + spills) and a scheduling barrier between their issue and the multiplies
// (Measured and dropped on the way: every lane loading its operands' 16 consecutive floats straight from global memory, no LDS:
// 2.35 ms per 32 x 256 forward against 2.13 — 32 rows x 64 bytes per load instruction keep the address path busy four times as
// long as full lines do; 32-wide K steps through two 60 KB workgroup-wide buffers, one barrier a step: 2.27 ms; starting the
// second wave of each SIMD up to 2.7 us late so that the two would not stage at the same time: no change.)
// ------------------------------------------------------------------------------------------------
constexpr int L32M = 32;
template <int CT>
__global__ __launch_bounds__(512, 2) void gemm_f32_ln32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                const float* __restrict__ bias, const float* __restrict__ resid,
                                                                const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                                float eps, float* __restrict__ C, int M, int K) {
    constexpr int N = CT * 128;
    constexpr int LDR = N + 4;
    constexpr int Q = N / 64;
    constexpr int WR = CT * 32;   // W rows of a wave
    constexpr int PR = 32 + WR;   // rows of a wave's staging tile: A rows, then its W rows
    constexpr int WJ = WR / 8;    // load instructions per step for W (8 rows x 128 bytes each), 4 for A
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    static_assert(2 * 32 * LDR <= 8 * PR * LDT, "the epilogue's two partial tiles fit the staging regions");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int kh = wave & 1, wc = wave >> 1;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int erow = tid >> 4, ej = tid & 15;
    const int tiles = (M + L32M - 1) / L32M;
    const int nk = K / 64;  // 32-wide steps in this wave's half of K
    float* mine = lsm + wave * PR * LDT;
    float* tile_lds = lsm;
    float* prm = lsm + 8 * PR * LDT;  // bias, LayerNorm weight and bias [3][N], behind the staging regions (see gemm_f32_ln8_kernel)
    if (tid < 3 * N / 4) {
        const float* src = tid < N / 4 ? bias : tid < N / 2 ? ln_w : ln_b;
        *(f32x4*)(prm + 4 * tid) = *(const f32x4*)(src + 4 * (tid % (N / 4)));
    }  // (visible after the epilogue's first barrier)
    const float* pcol = prm + 4 * ej;
    // operands through buffer descriptors with wave-uniform bases: ONE address register (the lane's offset inside an 8-row group)
    // serves all 16 loads of a step, the rest of the address is scalar; rows past M read as zeros
    const uint32_t lane_off = (uint32_t)(r8 * K + c8 * 4) * 4u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int kh_u = wave_u & 1, wc_u = wave_u >> 1;
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (size_t)wc_u * WR * K + kh_u * (K / 2)), 0, (uint32_t)(((WR - 1) * K + K / 2) * 4), 0x00020000);
    float* st_a = mine + r8 * LDT + c8 * 4;
    const float* ap0 = mine + i * LDT + 16 * kk;
    const float* bp0 = mine + (32 + i) * LDT + 16 * kk;

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * L32M;
        const int rows_here = min(M - m0, L32M);
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(A + (size_t)m0 * K + kh_u * (K / 2)), 0, (uint32_t)(((rows_here - 1) * K + K / 2) * 4), 0x00020000);
        f32x4 ra[4], rw[WJ];
        auto fetch = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, lane_off, (uint32_t)(8 * j * K + kt * 32) * 4u, 0));
#pragma unroll
            for (int j = 0; j < WJ; ++j) rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, lane_off, (uint32_t)(8 * j * K + kt * 32) * 4u, 0));
        };
        fetch(0);
        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
        auto to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(st_a + 8 * j * LDT) = ra[j];
#pragma unroll
            for (int j = 0; j < WJ; ++j) *(f32x4*)(st_a + (32 + 8 * j) * LDT) = rw[j];
            __builtin_amdgcn_wave_barrier();  // (scheduling only: one wave, and its LDS operations complete in order)
        };
        auto mfma_step = [&]() __attribute__((always_inline)) {
            f32x4 x[2], y[2][CT];  // fragments of four k per lane half, the next four on their way under these multiplies
            x[0] = *(const f32x4*)ap0;
#pragma unroll
            for (int c = 0; c < CT; ++c) y[0][c] = *(const f32x4*)(bp0 + c * 32 * LDT);
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                if (hh < 3) {
                    x[(hh + 1) & 1] = *(const f32x4*)(ap0 + 4 * (hh + 1));
#pragma unroll
                    for (int c = 0; c < CT; ++c) y[(hh + 1) & 1][c] = *(const f32x4*)(bp0 + c * 32 * LDT + 4 * (hh + 1));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[hh & 1][s], y[hh & 1][c][s], acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        };
        for (int kt = 0; kt + 1 < nk; ++kt) {
            to_lds();
            fetch(kt + 1);
            __builtin_amdgcn_sched_barrier(0);  // the loads are ISSUED here: the scheduler otherwise sinks them below the multiplies
            mfma_step();
        }
        to_lds();
        f32x4 rv[Q];
        {
            const float* rrow = resid + (size_t)min(m0 + erow, M - 1) * N;
#pragma unroll
            for (int t = 0; t < Q; ++t) rv[t] = *(const f32x4*)(rrow + 4 * (ej + 16 * t));
        }
        mfma_step();

        __syncthreads();  // every wave is done with its staging region: the partial tiles take their place
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile_lds[(kh * 32 + acc_row(r, kk)) * LDR + (wc * CT + c) * 32 + i] = acc[c][r];
        __syncthreads();
        const int row = m0 + erow;
        f32x4 v[Q];
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const int col = 4 * (ej + 16 * t);
            v[t] = (*(const f32x4*)(tile_lds + erow * LDR + col) + *(const f32x4*)(tile_lds + (32 + erow) * LDR + col)) +
                   *(const f32x4*)(pcol + 64 * t) + rv[t];
            sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 8);
        const float mean = sum / (float)N;
        float sq = 0.0f;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const f32x4 d = v[t] - mean;
            sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        sq += __shfl_xor(sq, 1);
        sq += __shfl_xor(sq, 2);
        sq += __shfl_xor(sq, 4);
        sq += __shfl_xor(sq, 8);
        const float rstd = 1.0f / sqrtf(sq / (float)N + eps);
        if (row < M) {
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                const int col = 4 * (ej + 16 * t);
                *(f32x4*)(C + (size_t)row * N + col) = (v[t] - mean) * rstd * *(const f32x4*)(pcol + N + 64 * t) + *(const f32x4*)(pcol + 2 * N + 64 * t);
            }
        }
        __syncthreads();  // the partial tiles are read: the next tile's staging may overwrite them
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 128 (a single query, a few highlight chunks): the 128x128 tiling would run
// N/128 workgroups through K/32 barrier-separated steps each (30-115 us per layer GEMM, launch- and
// latency-bound).  Here a workgroup owns a 32-column strip of the output for ALL rows, its 8 waves
// split K, every lane loads its MFMA operands straight from global memory (all loads of a wave in
// flight at once, no LDS staging), and the 8 partial accumulators meet in LDS.  Same exact-f32
// v_mfma_f32_32x32x2_f32 arithmetic; only the summation order over K differs (8 partial sums).
// ------------------------------------------------------------------------------------------------
template <int EPI, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ resid, float* __restrict__ C,
                                                              int M, int N, int K) {
    __shared__ float red[NW][MT][16][64];  // [wave][row tile][acc reg][lane]; NW waves split K
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kper = K / NW;  // K is a multiple of 128: kper is a multiple of 16
    const int kbeg = wave * kper;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const float* wp = W + (size_t)(n0 + i) * K + kbeg + 8 * kk;
    const float* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) ap[t] = A + (size_t)min(32 * t + i, M - 1) * K + kbeg + 8 * kk;
    for (int k0 = 0; k0 < kper; k0 += 16) {  // lane (i, kk) holds k = k0 + 8*kk + s, s = 0..7, of row i
        const f32x4 b0 = *(const f32x4*)(wp + k0), b1 = *(const f32x4*)(wp + k0 + 4);
        const float bf[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const f32x4 a0 = *(const f32x4*)(ap[t] + k0), a1 = *(const f32x4*)(ap[t] + k0 + 4);
            const float af[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2], bf[s2], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    // all threads sum the NW partials of MT*16*64 values in wave order and finish the epilogue
    for (int e = tid; e < MT * 16 * 64; e += NW * 64) {
        const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][t][r][l];
        const int row = 32 * t + acc_row(r, l >> 5), col = n0 + (l & 31);
        if (row < M) {
            v += bias ? bias[col] : 0.0f;
            v = epi_act<EPI>(v);
            if (EPI == EPI_BIAS_RESIDUAL) v += resid[(size_t)row * N + col];
            C[(size_t)row * N + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-precision GEMM: C = A W^T with x = hi + mid + lo (three bf16 terms, 24 significand bits).
//   a*b ~ ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh        (dropped: am*bl, al*bm, al*bl < 2^-24)
// Six v_mfma_f32_32x32x16_bf16 per 16-deep step instead of eight v_mfma_f32_32x32x2_f32 at 1/16 the
// rate: 6/16 of the matrix-core time of the exact-f32 kernel at f32-level accuracy.
// Same 128x128x32 tiling as gemm_f32_kernel.  LDS holds three bf16 planes per operand, rows padded
// to 40 elements (80 B: the 16-lane ds_read_b128 groups cover all 64 banks).  W arrives pre-split
// (model load), A is split by the staging threads between its global load and the LDS store.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDB = 40;

// Pins memory operations in program order: the asm is a compiler-level memory barrier (IR passes may
// otherwise reorder the loads around it), sched_barrier stops the machine scheduler.
#define PCV_PIN_ORDER()                     \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

// x = hi + mid + lo with round-to-nearest at every level, two elements at a time so that each level is one
// v_cvt_pk_bf16_f32 + shift + and + (packed) subtraction: 5.5 VALU per element; the plain vector form
// compiled to 11.  VALU is not free here: on gfx950 VALU work of ANY wave of a SIMD takes the cycles its
// MFMAs would use (tools/ubench/coexec.hip: MFMA-only 1.31 ms, VALU-only 1.07 ms, both on one SIMD 2.37 ms).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    const bf16x2 p = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3_pair(x.x, x.y, h0, m0, l0);
    split3_pair(x.z, x.w, h1, m1, l1);
    hi = __builtin_bit_cast(bf16x4, u32x2{h0, h1});
    mid = __builtin_bit_cast(bf16x4, u32x2{m0, m1});
    lo = __builtin_bit_cast(bf16x4, u32x2{l0, l1});
}

// (Variant tried and dropped: activations pre-split into planes by the producing kernels.  6 B/element
// of staging traffic instead of 4, 2-byte plane stores in every producer: slower end to end.)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Wh,
                                                          const uint16_t* __restrict__ Wm,
                                                          const uint16_t* __restrict__ Wl,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ resid, float* __restrict__ C,
                                                          int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDB];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[3][BN * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    int tile_m, tile_n;
    tile_of_block(N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // Staging maps.  LDS rows are 80 bytes; a group of lanes that one ds_write services together must
    // touch rows r and r+4 (80*4 = 64 mod 128) to cover all 32 banks exactly once.
    //   A: float4 -> three 8-byte plane pieces; 8 lanes cover a row's 32 k, lane pairs of groups
    //      (2j, 2j+1) take rows r, r+4.  Row of group g: 8*((g>>1)>>2) + ((g>>1)&3) + 4*(g&1), +32u.
    //   W: 16-byte pieces of each pre-split plane; lanes 0-3 of an 8-lane group take row r, 4-7 row r+4.
    const int ag_ = tid >> 3, ac4 = tid & 7;
    const int arow = 8 * ((ag_ >> 1) >> 2) + ((ag_ >> 1) & 3) + 4 * (ag_ & 1);
    const float* ag[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ag[u] = A + (size_t)min(m0 + arow + 32 * u, M - 1) * K + ac4 * 4;
    const int wg_ = tid >> 3, wsub = tid & 7;
    const int wc8 = wsub & 3;
    const int wrow = (wg_ >> 2) * 8 + (wg_ & 3) + 4 * (wsub >> 2);  // 0..63, second piece +64
    const size_t woff0 = (size_t)(n0 + wrow) * K + wc8 * 8;
    const size_t woff1 = woff0 + (size_t)64 * K;

    // two register sets: the loads of K tile kt+2 are issued while tile kt is multiplied, so a tile has
    // two MFMA phases (~1.5 us) to arrive — one phase (0.75 us at six bf16 MFMAs per product) does not
    // cover the memory latency and left the matrix pipe 41 % busy
    f32x4 ra[2][4];
    f32x4 rw[2][3][2];  // 8 bf16 = 16 bytes, moved as f32x4
    auto load_tile = [&](int set, int kt) {
        const size_t koff = (size_t)kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[set][u] = *(const f32x4*)(ag[u] + koff);
        rw[set][0][0] = *(const f32x4*)(Wh + woff0 + koff); rw[set][0][1] = *(const f32x4*)(Wh + woff1 + koff);
        rw[set][1][0] = *(const f32x4*)(Wm + woff0 + koff); rw[set][1][1] = *(const f32x4*)(Wm + woff1 + koff);
        rw[set][2][0] = *(const f32x4*)(Wl + woff0 + koff); rw[set][2][1] = *(const f32x4*)(Wl + woff1 + koff);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = K / BK;  // even: K is a multiple of 128
    auto step = [&](int set, int kt) {
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x4 ph, pm, pl;
            split3(ra[set][u], ph, pm, pl);
            const int o = (arow + 32 * u) * LDB + ac4 * 4;
            *(bf16x4*)&As[0][o] = ph;
            *(bf16x4*)&As[1][o] = pm;
            *(bf16x4*)&As[2][o] = pl;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *(f32x4*)&Ws[pl][wrow * LDB + wc8 * 8] = rw[set][pl][0];
            *(f32x4*)&Ws[pl][(wrow + 64) * LDB + wc8 * 8] = rw[set][pl][1];
        }
        __syncthreads();
        // The refill is pinned here: left to itself the scheduler sinks these loads to the end of the
        // iteration (their registers are dead until then), which turns the s_waitcnt at the next use
        // into vmcnt(0) on loads issued moments earlier — no prefetch distance at all.
        __builtin_amdgcn_sched_barrier(0);
        load_tile(set, min(kt + 2, nk - 1));  // refill this set for tile kt+2 (clamped re-read at the end)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {  // two 16-deep steps per 32-wide K tile
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    af[t][pl] = *(const bf16x8*)&As[pl][(wr * 64 + t * 32 + i) * LDB + s * 16 + 8 * h];
                    bf[t][pl] = *(const bf16x8*)&Ws[pl][(wc * 64 + t * 32 + i) * LDB + s * 16 + 8 * h];
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // smallest terms first
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
    };
    // same issue order as inside the loop (set 0 before set 1), or the merged wait at the loop head
    // degrades to vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    load_tile(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(1, min(1, nk - 1));
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; kt += 2) {  // nk is even (K % 128 == 0): no branch between the two halves
        step(0, kt);
        step(1, kt + 1);
    }

    float bv[2];
    load_bias2(bias, n0 + wc * 64, i, bv);
    store_quarter<EPI>(acc, bv, resid, C, M, N, m0 + wr * 64, n0 + wc * 64, i, h);
}

// ------------------------------------------------------------------------------------------------
// The same GEMM as a PERSISTENT, WAVE-SPECIALISED kernel.  What the kernel above loses (measured with
// diagnostic builds and PMC on the 256x256-token batch): every wave stages, waits at two barriers and
// multiplies in turn, so the matrix pipe idles while it stages (bf16 pipe 39 % busy); and with K = 384 a
// tile is only 12 K steps long, so the fill of the load pipeline at its start and the drain of its stores
// before the workgroup can retire cost about as much as the steps in between.  Here
//   * one 512-thread workgroup per CU walks over tiles (tile = f(blockIdx.x + n * gridDim.x));
//   * waves 0-3 are consumers: fragment reads + 48 MFMAs per K step (a 64x64 quarter each), then the
//     epilogue; waves 4-7 are producers: global loads (two register sets in flight), the 3-way split, LDS
//     writes.  One of each per SIMD, so split VALU and MFMA interleave cycle by cycle;
//   * LDS is double buffered (2 x 60 KB) and ONE barrier per K step orders both directions: barrier g
//     tells the consumers buffer g&1 is full and tells the producers the consumers are done with buffer
//     (g+1)&1 (read in step g-1, before they arrived).  The step counter g runs on across tiles: while
//     the consumers store tile t the producers have already staged step 0 of tile t+1 and hold its next
//     two K tiles in registers, and the stores of tile t drain under the MFMAs of tile t+1.
// Same staging maps, fragment layout, product order and epilogue as above: results are bit-identical.
// ------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence over
// ALL address spaces, which on gfx9 means s_waitcnt vmcnt(0) whenever global stores are in flight: in
// the persistent kernel the consumers would sit out the HBM write latency of tile t at the first barrier
// of tile t+1.  LDS operations of a wave complete in order, so lgkmcnt(0) + s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int WS_PLANE = BM * LDB;     // elements of one plane of one operand (BM == BN)
constexpr int WS_BUF = 6 * WS_PLANE;   // A: 3 planes, W: 3 planes
constexpr size_t kGemmWsLdsBytes = 2 * (size_t)WS_BUF * sizeof(__bf16);

// tile visited by workgroup `wg` in its n-th round.  Workgroups go to XCDs round-robin (wg & 7); the 32
// workgroups of one XCD take 32 consecutive logical tiles of the round, i.e. all column tiles of a few
// row blocks, so a row block of A is fetched from HBM once and then served by that XCD's L2.
__device__ __forceinline__ unsigned ws_slot(unsigned wg, unsigned nwg) {
    return (nwg & 7u) == 0 ? (wg & 7u) * (nwg >> 3) + (wg >> 3) : wg;
}
__device__ __forceinline__ void ws_tile(unsigned slot, unsigned nwg, unsigned round, int ncols, int& tile_m, int& tile_n) {
    const unsigned logical = round * nwg + slot;
    tile_m = (int)(logical / (unsigned)ncols);
    tile_n = (int)(logical % (unsigned)ncols);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16x3_ws_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Wh,
                                                             const uint16_t* __restrict__ Wm,
                                                             const uint16_t* __restrict__ Wl,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, float* __restrict__ C,
                                                             int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    __bf16* const sm = (__bf16*)ws_smem;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // scalar: the role branch is uniform
    const int lane = threadIdx.x & 63;
    const int ncols = N / BN;
    const unsigned ntiles = (unsigned)ncols * (unsigned)((M + BM - 1) / BM);
    const unsigned nwg = gridDim.x, wg = ws_slot(blockIdx.x, gridDim.x);
    const unsigned rounds = (ntiles - wg + nwg - 1) / nwg;  // >= 1: the grid never exceeds the tile count
    const int nk = K / BK;                                  // even: K is a multiple of 128

    if (wave >= 4) {
        // ---- producers (staging maps of gemm_bf16x3_kernel, thread index within the producer half)
        const int pt = (int)threadIdx.x - 256;
        const int ag_ = pt >> 3, ac4 = pt & 7;
        const int arow = 8 * ((ag_ >> 1) >> 2) + ((ag_ >> 1) & 3) + 4 * (ag_ & 1);
        const int wg_ = pt >> 3, wsub = pt & 7;
        const int wc8 = wsub & 3;
        const int wrow = (wg_ >> 2) * 8 + (wg_ & 3) + 4 * (wsub >> 2);
        // source of the NEXT refill: K tile `lk` of round `lround`
        const float* ag[4];
        size_t woff0, woff1;
        auto point_at = [&](unsigned round) {
            int tm, tn;
            ws_tile(wg, nwg, round, ncols, tm, tn);
#pragma unroll
            for (int u = 0; u < 4; ++u) ag[u] = A + (size_t)min(tm * BM + arow + 32 * u, M - 1) * K + ac4 * 4;
            woff0 = (size_t)(tn * BN + wrow) * K + wc8 * 8;
            woff1 = woff0 + (size_t)64 * K;
        };
        unsigned lround = 0;
        int lk = 0;
        point_at(0);
        f32x4 ra[2][4];
        f32x4 rw[2][3][2];
        auto load_next = [&](int set) {  // loads K tile (lround, lk) into register set `set`, then advances
            const size_t koff = (size_t)lk * BK;
#pragma unroll
            for (int u = 0; u < 4; ++u) ra[set][u] = *(const f32x4*)(ag[u] + koff);
            rw[set][0][0] = *(const f32x4*)(Wh + woff0 + koff); rw[set][0][1] = *(const f32x4*)(Wh + woff1 + koff);
            rw[set][1][0] = *(const f32x4*)(Wm + woff0 + koff); rw[set][1][1] = *(const f32x4*)(Wm + woff1 + koff);
            rw[set][2][0] = *(const f32x4*)(Wl + woff0 + koff); rw[set][2][1] = *(const f32x4*)(Wl + woff1 + koff);
            if (++lk == nk) {  // uniform; address arithmetic only (past the last tile: stay on it, the data is unused)
                lk = 0;
                if (lround + 1 < rounds) ++lround;
                point_at(lround);
            }
        };
        auto produce = [&](int set) {  // set == step parity == buffer
            __bf16* const As = sm + set * WS_BUF;
            __bf16* const Ws = As + 3 * WS_PLANE;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bf16x4 ph, pm, pl;
                split3(ra[set][u], ph, pm, pl);
                const int o = (arow + 32 * u) * LDB + ac4 * 4;
                *(bf16x4*)&As[o] = ph;
                *(bf16x4*)&As[WS_PLANE + o] = pm;
                *(bf16x4*)&As[2 * WS_PLANE + o] = pl;
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *(f32x4*)&Ws[pl * WS_PLANE + wrow * LDB + wc8 * 8] = rw[set][pl][0];
                *(f32x4*)&Ws[pl * WS_PLANE + (wrow + 64) * LDB + wc8 * 8] = rw[set][pl][1];
            }
            // The refill is pinned here.  Left alone the scheduler sinks these loads to the end of the
            // iteration (their registers are dead until then) and the s_waitcnt at the next use becomes
            // vmcnt(0) on loads issued moments earlier: no prefetch distance at all.
            PCV_PIN_ORDER();
            load_next(set);
            PCV_PIN_ORDER();
            lds_barrier();  // barrier g: buffer `set` is full
        };
        PCV_PIN_ORDER();
        load_next(0);
        PCV_PIN_ORDER();
        load_next(1);
        PCV_PIN_ORDER();
        // First pair of steps outside the loop: the const __restrict__ prologue loads above may still be
        // issued set 1 first (IR passes ignore both pins for invariant loads), and a loop header that
        // merges that order with the steady-state order waits with vmcnt(0..9) for ever.  After this pair
        // the only loads in flight are the two pinned refills, in loop order.
        produce(0);
        produce(1);
        const unsigned pairs = rounds * (unsigned)(nk / 2);
        for (unsigned g2 = 1; g2 < pairs; ++g2) {
            produce(0);
            produce(1);
        }
        return;
    }

    // ---- consumers
    const int i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int aoff = (wr * 64 + i) * LDB + 8 * h, woff = (wc * 64 + i) * LDB + 8 * h;
    for (unsigned round = 0; round < rounds; ++round) {
        int tile_m, tile_n;
        ws_tile(wg, nwg, round, ncols, tile_m, tile_n);
        float bv[2];  // fetched now, used after the K loop
        load_bias2(bias, tile_n * BN + wc * 64, i, bv);
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        for (int kt = 0; kt < nk; ++kt) {
            lds_barrier();  // barrier g
            const __bf16* const As = sm + (kt & 1) * WS_BUF;
            const __bf16* const Ws = As + 3 * WS_PLANE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[2][3], bf[2][3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        af[t][pl] = *(const bf16x8*)&As[pl * WS_PLANE + aoff + t * 32 * LDB + s * 16];
                        bf[t][pl] = *(const bf16x8*)&Ws[pl * WS_PLANE + woff + t * 32 * LDB + s * 16];
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        // smallest terms first
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                    }
            }
        }
        store_quarter<EPI>(acc, bv, resid, C, M, N, tile_m * BM + wr * 64, tile_n * BN + wc * 64, i, h);
    }
}

// ------------------------------------------------------------------------------------------------
// Two-term f16 split GEMM (PCV_COMPUTE_F16X2): x = hi + lo with hi = RN_f16(x), lo = RN_f16(x - hi).  f16
// carries 11 significand bits, so two terms hold 22+ (|x - hi - lo| <= 2^-24 |x|, the f32 rounding unit,
// as long as lo stays in the normal range), and
//   a*b ~ ah*bh + ah*bl + al*bh                                  (dropped: al*bl < 2^-24 relative)
// is THREE v_mfma_f32_32x32x16_f16 per 16-deep step against six bf16 ones — half the matrix time and two
// thirds of the LDS traffic (the binding resource) of the three-term bf16 form, at the same
// accuracy.  The price is f16's range: operands are scaled by exact powers of two so that the low terms stay
// normal (activations x 2^4 while they are split, weights x 2^8 when their planes are made; the accumulator
// is multiplied by 2^-12 in the epilogue), and |activation| must stay below 65504 / 16 — far above anything
// a BERT-family encoder produces; weights are checked when the planes are built.
// Same tiling, staging maps, prefetch and epilogue as gemm_bf16x3_kernel with two planes per operand.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kF16ActScale = 16.0f, kF16WeightScale = 256.0f;

__device__ __forceinline__ void split2_pair(f32x2 x, uint32_t& hi, uint32_t& lo) {
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split2(const f32x4 x, float scale, f16x4& hi, f16x4& lo) {
    uint32_t h0, l0, h1, l1;
    split2_pair(f32x2{x.x, x.y} * scale, h0, l0);
    split2_pair(f32x2{x.z, x.w} * scale, h1, l1);
    hi = __builtin_bit_cast(f16x4, u32x2{h0, h1});
    lo = __builtin_bit_cast(f16x4, u32x2{l0, l1});
}

// NSET: register sets of prefetched K tiles (2 = two steps of distance at 2 workgroups per CU; 1 = one step,
// small enough for 3 workgroups per CU)
template <int EPI, int NSET>
__global__ __launch_bounds__(256, NSET == 1 ? 3 : 2) void gemm_f16x2_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Wh,
                                                         const uint16_t* __restrict__ Wl,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ resid, float* __restrict__ C,
                                                         int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) _Float16 As[2][BM * LDB];
    __shared__ __attribute__((aligned(16))) _Float16 Ws[2][BN * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    int tile_m, tile_n;
    tile_of_block(N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int ag_ = tid >> 3, ac4 = tid & 7;
    const int arow = 8 * ((ag_ >> 1) >> 2) + ((ag_ >> 1) & 3) + 4 * (ag_ & 1);
    const float* ag[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ag[u] = A + (size_t)min(m0 + arow + 32 * u, M - 1) * K + ac4 * 4;
    const int wg_ = tid >> 3, wsub = tid & 7;
    const int wc8 = wsub & 3;
    const int wrow = (wg_ >> 2) * 8 + (wg_ & 3) + 4 * (wsub >> 2);
    const size_t woff0 = (size_t)(n0 + wrow) * K + wc8 * 8;
    const size_t woff1 = woff0 + (size_t)64 * K;

    f32x4 ra[NSET][4];
    f32x4 rw[NSET][2][2];
    auto load_tile = [&](int set, int kt) {
        const size_t koff = (size_t)kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[set][u] = *(const f32x4*)(ag[u] + koff);
        rw[set][0][0] = *(const f32x4*)(Wh + woff0 + koff); rw[set][0][1] = *(const f32x4*)(Wh + woff1 + koff);
        rw[set][1][0] = *(const f32x4*)(Wl + woff0 + koff); rw[set][1][1] = *(const f32x4*)(Wl + woff1 + koff);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = K / BK;  // even: K is a multiple of 128
    auto step = [&](int set, int kt) {
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x4 ph, pl;
            split2(ra[set][u], kF16ActScale, ph, pl);
            const int o = (arow + 32 * u) * LDB + ac4 * 4;
            *(f16x4*)&As[0][o] = ph;
            *(f16x4*)&As[1][o] = pl;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            *(f32x4*)&Ws[pl][wrow * LDB + wc8 * 8] = rw[set][pl][0];
            *(f32x4*)&Ws[pl][(wrow + 64) * LDB + wc8 * 8] = rw[set][pl][1];
        }
        __syncthreads();
        PCV_PIN_ORDER();  // refill pinned: see gemm_bf16x3_kernel
        load_tile(set, min(kt + NSET, nk - 1));
        PCV_PIN_ORDER();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 af[2][2], bf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    af[t][pl] = *(const f16x8*)&As[pl][(wr * 64 + t * 32 + i) * LDB + s * 16 + 8 * h];
                    bf[t][pl] = *(const f16x8*)&Ws[pl][(wc * 64 + t * 32 + i) * LDB + s * 16 + 8 * h];
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // smallest terms first
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
    };
    PCV_PIN_ORDER();
    load_tile(0, 0);
    PCV_PIN_ORDER();
    if constexpr (NSET == 2) {
        load_tile(1, min(1, nk - 1));
        PCV_PIN_ORDER();
        for (int kt = 0; kt < nk; kt += 2) {
            step(0, kt);
            step(1, kt + 1);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) step(0, kt);
    }
    constexpr float inv = 1.0f / (kF16ActScale * kF16WeightScale);  // exact power of two
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] *= inv;
    float bv[2];
    load_bias2(bias, n0 + wc * 64, i, bv);
    store_quarter<EPI>(acc, bv, resid, C, M, N, m0 + wr * 64, n0 + wc * 64, i, h);
}

// f32 [n] -> two f16 planes of 2^8 * x; *overflow is raised if a scaled weight leaves f16's range
__global__ __launch_bounds__(256) void split_planes_f16_kernel(const float* __restrict__ src, int64_t n,
                                                               uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                               int* __restrict__ overflow) {
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;  // n is a multiple of 4 for every weight matrix
    const f32x4 x = *(const f32x4*)(src + p);
    f16x4 ph, pl;
    split2(x, kF16WeightScale, ph, pl);
    *(f16x4*)(hi + p) = ph;
    *(f16x4*)(lo + p) = pl;
    const float lim = 65504.0f / kF16WeightScale;
    if (!(fabsf(x.x) < lim && fabsf(x.y) < lim && fabsf(x.z) < lim && fabsf(x.w) < lim)) *overflow = 1;
}

// ------------------------------------------------------------------------------------------------
// Attention with two-term f16 splits (PCV_COMPUTE_F16X2): the same S^T = K Q^T / online softmax / O += P V
// scheme as attention_kernel, with each product as three v_mfma_f32_32x32x16_f16 on hi/lo planes —
// 12 MFMAs of 32 cycles per 32-key tile at head_dim 32 against 32 of 64 cycles.
//   * K is staged as [key][d] f16 hi/lo planes (rows padded by 16 B: conflict-free ds_read_b128 fragments),
//     V as the TRANSPOSE [d][slot] with the 32 keys of a tile permuted so that the eight keys a lane owns in
//     an S^T accumulator — acc_row(8j..8j+7, h) — are eight consecutive slots: P^T is then the B operand of
//     P V straight from the softmax registers, and the V^T fragment is one 16-byte read.
//   * O^T = V^T P^T puts the query on the lane for the output too: the running rescale and the final 1/l are
//     per-lane scalars, no shuffles.
//   * exact power-of-two scalings keep the low terms normal: Q, K, V x 2^4 while split, P x 2^10
//     (P <= 1); undone in f32 (x 2^-8 on the scores, x 2^-14 on the output).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split2x8(const float (&x)[8], float scale, f16x8& hi, f16x8& lo) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_pair(f32x2{x[2 * e], x[2 * e + 1]} * scale, hw[e], lw[e]);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(f16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    lo = __builtin_bit_cast(f16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// NW waves per workgroup (32 queries each; K / V^T are staged once per workgroup), TPC 32-key tiles per softmax chunk
template <int HD, int NW, int TPC>
__global__ __launch_bounds__(NW * 64) void attention_f16_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ mask_add, float* __restrict__ ctx,
                                                            int B, int L, int Lp, int H) {
    constexpr int NC = HD / 16;      // 16-deep chunks of the head dimension (QK^T k-steps)
    constexpr int CT = HD / 32;      // 32-row tiles of V^T / O^T
    constexpr int KP = HD * 2 + 16;  // bytes per K row (one plane)
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
    const int VP = Lp * 2 + 16;  // bytes per V^T row (one plane); (VP/4) mod 64 is 4 * odd: conflict-free b128 reads
    unsigned char* const Kh = att_lds;
    unsigned char* const Kl = Kh + (size_t)Lp * KP;
    unsigned char* const Vh = Kl + (size_t)Lp * KP;
    unsigned char* const Vl = Vh + (size_t)HD * VP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x * NW + wave, head = blockIdx.y, b = blockIdx.z;
    const int H3 = 3 * H;
    constexpr float kIn = 16.0f, kP = 1024.0f;

    {  // stage K: [key][d] planes; keys in [L, Lp) are zero (their scores are masked to -inf anyway)
        constexpr int C4 = HD / 4;
        for (int e = threadIdx.x; e < Lp * C4; e += NW * 64) {
            const int key = e / C4, c4 = e - key * C4;
            f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
            if (key < L) x = *(const f32x4*)(qkv + (size_t)(b * L + key) * H3 + H + head * HD + c4 * 4);
            f16x4 ph, pl;
            split2(x, kIn, ph, pl);
            *(f16x4*)(Kh + (size_t)key * KP + c4 * 8) = ph;
            *(f16x4*)(Kl + (size_t)key * KP + c4 * 8) = pl;
        }
        // stage V^T: a thread takes two neighbouring keys (neighbours in slot order too) and four d
        for (int e = threadIdx.x; e < (Lp / 2) * C4; e += NW * 64) {
            const int kp2 = e / C4, c4 = e - kp2 * C4;
            const int k0 = 2 * kp2;
            f32x4 x0 = {0.0f, 0.0f, 0.0f, 0.0f}, x1 = x0;
            if (k0 < L) x0 = *(const f32x4*)(qkv + (size_t)(b * L + k0) * H3 + 2 * H + head * HD + c4 * 4);
            if (k0 + 1 < L) x1 = *(const f32x4*)(qkv + (size_t)(b * L + k0 + 1) * H3 + 2 * H + head * HD + c4 * 4);
            // key = 32T + 16j + 8g + 4hh + e3  ->  slot = 32T + 16j + 8hh + 4g + e3
            const int w = k0 & 31;
            const int slot = (k0 & ~31) + (w & 16) + ((w & 4) << 1) + ((w & 8) >> 1) + (w & 3);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                uint32_t hw, lw;
                split2_pair(f32x2{x0[dd], x1[dd]} * kIn, hw, lw);
                *(uint32_t*)(Vh + (size_t)(c4 * 4 + dd) * VP + slot * 2) = hw;
                *(uint32_t*)(Vl + (size_t)(c4 * 4 + dd) * VP + slot * 2) = lw;
            }
        }
        __syncthreads();
    }
    if (qb * 32 >= L) return;
    constexpr float kLog2e = 1.44269504088896340736f;
    const float sscale = (kLog2e / (kIn * kIn)) / sqrtf((float)HD);
    const float ninf = -__builtin_inff();

    // Q^T fragments: lane (i, h) holds query i, d = 16c + 8h .. + 7
    f16x8 qh[NC], ql[NC];
    {
        const int qrow = min(qb * 32 + i, L - 1);
        const float* qp = qkv + (size_t)(b * L + qrow) * H3 + head * HD + 8 * h;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 a = *(const f32x4*)(qp + 16 * c), bq = *(const f32x4*)(qp + 16 * c + 4);
            const float x[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
            split2x8(x, kIn, qh[c], ql[c]);
        }
    }

    float m_run = ninf, l_run = 0.0f;
    f32x16 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.0f;

    for (int k0 = 0; k0 < L; k0 += 32 * TPC) {
        f32x16 s[TPC];
        float mc = ninf;
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
            const int kt = k0 + 32 * t;
            if (kt >= L) {  // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = ninf;
                continue;
            }
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const size_t off = (size_t)(kt + i) * KP + (16 * c + 8 * h) * 2;
                const f16x8 kh = *(const f16x8*)(Kh + off), kl = *(const f16x8*)(Kl + off);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], a, 0, 0, 0);
            }
            const float* mp = mask_add + (size_t)b * Lp + kt + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 ma = *(const float4*)(mp + 8 * g);  // keys kt + 8g + 4h + 0..3 (-inf beyond L)
                s[t][4 * g + 0] = fmaf(a[4 * g + 0], sscale, ma.x * kLog2e);  // scores in log2 units
                s[t][4 * g + 1] = fmaf(a[4 * g + 1], sscale, ma.y * kLog2e);
                s[t][4 * g + 2] = fmaf(a[4 * g + 2], sscale, ma.z * kLog2e);
                s[t][4 * g + 3] = fmaf(a[4 * g + 3], sscale, ma.w * kLog2e);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mc = fmaxf(mc, s[t][r]);
        }
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float m_new = fmaxf(m_run, mc);
        // exp2 on scores kept in log2 units: one v_exp_f32 per probability (libm expf is ~25 VALU, and VALU
        // time is matrix-pipe time on this part).  m_new is finite — key 0 is always a real key — so
        // exp2(-inf - m_new) is an exact 0 for masked-out keys and no select is needed.
        const float factor = (m_run == ninf) ? 0.0f : __builtin_amdgcn_exp2f(m_run - m_new);
        float lc = 0.0f;
#pragma unroll
        for (int t = 0; t < TPC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[t][r] - m_new);
                s[t][r] = p;
                lc += p;
            }
        lc += __shfl_xor(lc, 32);
        l_run = l_run * factor + lc;
        m_run = m_new;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] *= factor;  // the lane IS the query: its own factor
        // O^T += V^T P^T, 16 key slots per MFMA: slots 8h..8h+7 of MFMA j are this lane's registers 8j..8j+7
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
            const int kt = k0 + 32 * t;
            if (kt >= L) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float pv[8] = {s[t][8 * j], s[t][8 * j + 1], s[t][8 * j + 2], s[t][8 * j + 3],
                                     s[t][8 * j + 4], s[t][8 * j + 5], s[t][8 * j + 6], s[t][8 * j + 7]};
                f16x8 ph, pl;
                split2x8(pv, kP, ph, pl);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const size_t off = (size_t)(32 * c + i) * VP + (size_t)(kt + 16 * j + 8 * h) * 2;
                    const f16x8 vh = *(const f16x8*)(Vh + off), vl = *(const f16x8*)(Vl + off);
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[c], 0, 0, 0);
                }
            }
        }
    }
    // lane (i, h), register 4g + e of tile c: query i, d = 32c + 8g + 4h + e -> four consecutive floats
    if (qb * 32 + i < L) {
        const float fin = (1.0f / (kIn * kP)) / l_run;
        float* const op = ctx + (size_t)(b * L + qb * 32 + i) * H + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(op + 32 * c + 8 * g) = f32x4{o[c][4 * g] * fin, o[c][4 * g + 1] * fin, o[c][4 * g + 2] * fin, o[c][4 * g + 3] * fin};
    }
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t n,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ mid,
                                                           uint16_t* __restrict__ lo) {
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;  // n is a multiple of 4 for every weight matrix
    bf16x4 ph, pm, pl;
    split3(*(const f32x4*)(src + p), ph, pm, pl);
    *(bf16x4*)(hi + p) = ph;
    *(bf16x4*)(mid + p) = pm;
    *(bf16x4*)(lo + p) = pl;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per token row, H/64 values per lane (H <= 1024)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxPerLane = 16;

__device__ __forceinline__ void ln_row(float (&v)[kMaxPerLane], int n_per_lane, int H, int lane,
                                       const float* __restrict__ w, const float* __restrict__ b, float eps,
                                       float* __restrict__ dst) {
    float s = 0.0f;
    for (int j = 0; j < n_per_lane; ++j) s += v[j];
    const float mean = wave_sum(s) / (float)H;
    float q = 0.0f;
    for (int j = 0; j < n_per_lane; ++j) {
        const int c = lane + 64 * j;
        const float d = c < H ? v[j] - mean : 0.0f;
        q += d * d;
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    for (int j = 0; j < n_per_lane; ++j) {
        const int c = lane + 64 * j;
        if (c < H) dst[c] = (v[j] - mean) * inv * w[c] + b[c];
    }
}

__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* __restrict__ ids,
                                                       const int64_t* __restrict__ mask, int B, int L, int Lp, int H,
                                                       int vocab, const float* __restrict__ word,
                                                       const float* __restrict__ pos, const float* __restrict__ type,
                                                       const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                       float eps, float* __restrict__ hidden,
                                                       float* __restrict__ mask_add, float* __restrict__ mask01) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);  // token slot in the padded [B][Lp] grid
    if (t >= B * Lp) return;
    const int b = t / Lp, l = t - b * Lp;
    if (l >= L) {  // padding keys of the last 32-key tile: excluded from every softmax
        if (lane == 0) mask_add[t] = -__builtin_inff();
        return;
    }
    const int tok = b * L + l;
    if (lane == 0) {
        const float m = (float)mask[tok];
        mask_add[t] = (1.0f - m) * -10000.0f;  // rust-bert BertModel: additive attention mask
        mask01[tok] = m;
    }
    int64_t id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* we = word + (size_t)id * H;
    const float* pe = pos + (size_t)l * H;
    float v[kMaxPerLane];
    const int npl = (H + 63) / 64;
    for (int j = 0; j < npl; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < H ? we[c] + pe[c] + type[c] : 0.0f;
    }
    ln_row(v, npl, H, lane, ln_w, ln_b, eps, hidden + (size_t)tok * H);
}

// H a multiple of 64 (every BERT-family width): NPL values per lane known at compile time, TOK tokens per wave
// with their loads and both reduction chains interleaved (the generic kernel is latency-bound: one token's two
// dependent wave reductions per wave at a time).  Same arithmetic per token, in the same order.
template <int NPL, int TOK>
__global__ __launch_bounds__(256) void layer_norm_fixed_kernel(float* __restrict__ x, int T, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * TOK;
    if (t0 >= T) return;
    float v[TOK][NPL];
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
        const float* row = x + (size_t)min(t0 + k, T - 1) * H;
#pragma unroll
        for (int j = 0; j < NPL; ++j) v[k][j] = row[lane + 64 * j];
    }
    float wv[NPL], bv[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        wv[j] = w[lane + 64 * j];
        bv[j] = b[lane + 64 * j];
    }
    float mean[TOK], inv[TOK];
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) s += v[k][j];
        mean[k] = s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < TOK; ++k) mean[k] += __shfl_xor(mean[k], off);
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
        mean[k] = mean[k] / (float)H;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const float d = v[k][j] - mean[k];
            q += d * d;
        }
        inv[k] = q;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < TOK; ++k) inv[k] += __shfl_xor(inv[k], off);
#pragma unroll
    for (int k = 0; k < TOK; ++k) {
        if (t0 + k >= T) break;  // wave-uniform
        const float r = 1.0f / sqrtf(inv[k] / (float)H + eps);
        float* row = x + (size_t)(t0 + k) * H;
#pragma unroll
        for (int j = 0; j < NPL; ++j) row[lane + 64 * j] = (v[k][j] - mean[k]) * r * wv[j] + bv[j];
    }
}

__global__ __launch_bounds__(256) void layer_norm_kernel(float* __restrict__ x, int T, int H,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         float eps) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float* row = x + (size_t)t * H;
    float v[kMaxPerLane];
    const int npl = (H + 63) / 64;
    for (int j = 0; j < npl; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < H ? row[c] : 0.0f;
    }
    ln_row(v, npl, H, lane, w, b, eps, row);
}

// ------------------------------------------------------------------------------------------------
// Attention: one wave per (batch, head, 32-query block), online softmax over chunks of 32*TPC keys.
// S^T = K Q^T is computed so that the query sits on the lane and the keys in the accumulator
// registers: row max / sum are in-register reductions plus one cross-half exchange.  The output is
// accumulated transposed as well, O^T = V^T P^T: the probabilities, as they stand in the S^T
// accumulators, are the B operand of that product (k index = key acc_row(r,h), column = query = lane),
// so the running output of a query lives on the query's own lane and the online-softmax rescale is a
// per-lane multiply (the untransposed form needed 16 ds_bpermute per chunk to fetch each row's factor).
// q is scaled by log2(e)/sqrt(HD) once, so the scores come out of the MFMA in log2 units.
// ------------------------------------------------------------------------------------------------
// STAGED: the NW waves of a workgroup are NW query blocks of the same (batch, head), so the
// workgroup first copies that head's K and V for the whole sequence into LDS (rows padded by 4 floats:
// conflict-free ds_read_b128 for the K fragments, ds_read_b32 for V) plus the additive mask in log2
// units, and the inner loops never wait on global memory.  All of a thread's staging loads are issued
// before the first LDS write (the rolled loop waited for each pair of loads in turn: four dependent
// memory round trips per workgroup at 256 keys).  Used whenever the copies fit the LDS.
// NW waves per workgroup (32 queries each), TPC 32-key tiles per softmax chunk
// One wave's 32 queries against all L keys.  qf: this lane's half of its query row, already scaled.  NEXTQ: once the
// last chunk's scores are out, qf is dead, and the half row at `next_qp` is requested into it (the persistent form's
// next query block: the request then flies under the last softmax and P V product, the stores and the workgroup barriers).
template <int HD, bool STAGED, int TPC, bool NEXTQ>
__device__ __forceinline__ void attn_wave(float (&qf)[HD / 2], const float* next_qp, const float* Ks, const float* Vs, const float* Ms,
                                          const float* __restrict__ qkv, const float* __restrict__ mask_add,
                                          float* __restrict__ ctx, int b, int head, int qb, int L, int Lp, int H, int i, int h) {
    constexpr int KS = HD / 2;   // k-steps of the QK^T product; also floats of a row held per lane
    constexpr int CT = HD / 32;  // 32-wide row tiles of the transposed output
    constexpr int LDK = HD + 4;  // padded LDS row (floats)
    constexpr float kLog2e = 1.44269504088896340736f;
    const int H3 = 3 * H;
    const float ninf = -__builtin_inff();
    float m_run = ninf, l_run = 0.0f;
    f32x16 o[CT];  // O^T: o[c][r] = output dim 32c + acc_row(r,h) of query i
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.0f;

    // scores of the chunk of keys that starts at k0 (S^T tiles in log2 units, mask included) and their maximum per lane
    auto scores = [&](int k0, f32x16 (&s)[TPC], float& mc) {
        mc = ninf;
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
            const int kt = k0 + 32 * t;
            if (kt >= L) {  // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = ninf;
                continue;
            }
            const float* kp = STAGED ? &Ks[(kt + i) * LDK + KS * h]
                                     : qkv + (size_t)(b * L + min(kt + i, L - 1)) * H3 + H + head * HD + KS * h;
            float kf[KS];
#pragma unroll
            for (int v = 0; v < KS / 4; ++v) {
                const float4 x = *(const float4*)(kp + 4 * v);
                kf[4 * v] = x.x; kf[4 * v + 1] = x.y; kf[4 * v + 2] = x.z; kf[4 * v + 3] = x.w;
            }
            // the accumulators start at the additive mask of their keys kt + 8g + 4h + 0..3 in log2 units (0 for a real
            // token, -10000 log2(e) for a masked one, -inf beyond L), so the last MFMA delivers the finished scores
            f32x16 a;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 ma;
                if (STAGED) {
                    ma = *(const float4*)&Ms[kt + 4 * h + 8 * g];
                } else {
                    ma = *(const float4*)(mask_add + (size_t)b * Lp + kt + 4 * h + 8 * g);
                    ma.x *= kLog2e; ma.y *= kLog2e; ma.z *= kLog2e; ma.w *= kLog2e;
                }
                a[4 * g + 0] = ma.x; a[4 * g + 1] = ma.y; a[4 * g + 2] = ma.z; a[4 * g + 3] = ma.w;
            }
#pragma unroll
            for (int e = 0; e < KS; ++e) a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[e], a, 0, 0, 0);
            s[t] = a;
#pragma unroll
            for (int r = 0; r < 16; ++r) mc = fmaxf(mc, a[r]);
        }
    };
    // online-softmax step on one chunk's scores, then O^T += V^T P^T
    auto fold = [&](int k0, f32x16 (&s)[TPC], float mc) {
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float m_new = fmaxf(m_run, mc);
        // exp2 on scores kept in log2 units: one v_exp_f32 per probability (libm expf is ~25 VALU).  m_new is finite —
        // key 0 is always a real key — so exp2(-inf - m_new) is an exact 0 for masked-out keys and no select is needed.
        const float factor = (m_run == ninf) ? 0.0f : __builtin_amdgcn_exp2f(m_run - m_new);
        float lc = 0.0f;
#pragma unroll
        for (int t = 0; t < TPC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[t][r] - m_new);
                s[t][r] = p;
                lc += p;
            }
        lc += __shfl_xor(lc, 32);
        l_run = l_run * factor + lc;
        m_run = m_new;
        // rescale the running output: this lane's query, this lane's factor
        if (k0 > 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] *= factor;
        }
        // O^T += V^T P^T : k index = key acc_row(r,h) of tile t; A operand = V[key][32c + i], B operand = p of that key
        // (keys beyond L carry p = 0; their staged rows are zeros, the unstaged form clamps the index)
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
            const int kt = k0 + 32 * t;
            if (kt >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vp = STAGED ? &Vs[(kt + acc_row(r, h)) * LDK + i]
                                         : qkv + (size_t)(b * L + min(kt + acc_row(r, h), L - 1)) * H3 + 2 * H + head * HD + i;
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * c], s[t][r], o[c], 0, 0, 0);
            }
        }
    };
    // (measured and dropped, 256 x 256 tokens, 224 us per call either way: the scores of chunk c+1 computed before the
    // softmax of chunk c so that its vector instructions issue between those MFMAs; two alternating accumulators for the
    // P V product instead of one chain of 32 dependent MFMAs; s_setprio 1 for the younger half of the workgroup; 128-key
    // chunks: 254 us, 32-key chunks: 229 us.  With the loads, the exps, the LDS addresses and the stores all stubbed out
    // the kernel still took 218 us: the time was in workgroup turnover, see attention_persist_kernel.  Round 4, the same question
    // asked piece by piece: 237 us as it stands, 229 without the exponentials, 214 without any per-element softmax arithmetic.
    // And turnover is not it either: persistent workgroups that stream an item's keys and values through LDS in two double-buffered
    // chunks of 128 keys (the next chunk requested into 16 registers under this chunk's multiplies, one barrier per chunk, no
    // staging round trip in front of any item) took 230 us against 222 us on the same box (44.8 against 46.1 us at 32 x 256).  Half of the first
    // wave of workgroups started 18 us late, so that the two workgroups of a CU do not stage in step: 224 us as well.)
    f32x16 s[TPC];
    float mc;
    int k0 = 0;
    for (; k0 + 32 * TPC < L; k0 += 32 * TPC) {
        scores(k0, s, mc);
        fold(k0, s, mc);
    }
    scores(k0, s, mc);  // the last chunk (L >= 1: there always is one)
    if (NEXTQ) {
#pragma unroll
        for (int v = 0; v < KS / 4; ++v) {
            const float4 x = *(const float4*)(next_qp + 4 * v);
            qf[4 * v] = x.x; qf[4 * v + 1] = x.y; qf[4 * v + 2] = x.z; qf[4 * v + 3] = x.w;
        }
    }
    fold(k0, s, mc);
    // this lane holds, of query i, the output dims 32c + 8g + 4h + 0..3: four 16-byte stores per tile
    if (qb * 32 + i < L) {
        const float linv = 1.0f / l_run;
        float* const cp = ctx + (size_t)(b * L + qb * 32 + i) * H + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[c][4 * g] * linv, o[c][4 * g + 1] * linv, o[c][4 * g + 2] * linv, o[c][4 * g + 3] * linv};
                *(f32x4*)(cp + 32 * c + 8 * g) = v;
            }
    }
}

template <int HD, bool STAGED, int NW = 4, int TPC = 4>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const float* __restrict__ qkv,
                                                        const float* __restrict__ mask_add, float* __restrict__ ctx,
                                                        int B, int L, int Lp, int H) {
    constexpr int KS = HD / 2;
    constexpr int LDK = HD + 4;
    constexpr float kLog2e = 1.44269504088896340736f;
    extern __shared__ float kv_lds[];  // [Lp][LDK] keys, [Lp][LDK] values, [Lp] mask * log2(e)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x * NW + wave, head = blockIdx.y, b = blockIdx.z;
    const int H3 = 3 * H;
    float* Ks = kv_lds;
    float* Vs = kv_lds + (size_t)Lp * LDK;
    float* Ms = Vs + (size_t)Lp * LDK;

    // this lane's half of its query row (in flight while the keys and values are staged)
    const int qrow = min(qb * 32 + i, L - 1);
    const float* qp = qkv + (size_t)(b * L + qrow) * H3 + head * HD + KS * h;
    float qf[KS];
#pragma unroll
    for (int v = 0; v < KS / 4; ++v) {
        const float4 x = *(const float4*)(qp + 4 * v);
        qf[4 * v] = x.x; qf[4 * v + 1] = x.y; qf[4 * v + 2] = x.z; qf[4 * v + 3] = x.w;
    }
    if (STAGED) {
        constexpr int C4 = HD / 4;  // float4 pieces per row
        constexpr int U = 4;        // pieces of K and of V a thread has in flight
        // rows L..Lp-1 are staged as zeros, so that no key index in the loops needs clamping (an index clamp per
        // LDS read was a third of the kernel's vector instructions)
        const int total = Lp * C4, real = L * C4;
        for (int e0 = threadIdx.x; e0 < total; e0 += NW * 64 * U) {
            f32x4 kk[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = min(e0 + u * NW * 64, real - 1);
                const int key = e / C4, c4 = e - key * C4;
                const float* src = qkv + (size_t)(b * L + key) * H3 + H + head * HD + c4 * 4;
                kk[u] = *(const f32x4*)src;
                vv[u] = *(const f32x4*)(src + H);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * NW * 64;
                if (e < total) {
                    const int key = e / C4, c4 = e - key * C4;
                    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                    *(f32x4*)&Ks[key * LDK + c4 * 4] = e < real ? kk[u] : zero;
                    *(f32x4*)&Vs[key * LDK + c4 * 4] = e < real ? vv[u] : zero;
                }
            }
        }
        for (int e = threadIdx.x; e < Lp; e += NW * 64) Ms[e] = mask_add[(size_t)b * Lp + e] * kLog2e;
        __syncthreads();
    }
    if (qb * 32 >= L) return;
    const float scale = kLog2e / sqrtf((float)HD);
#pragma unroll
    for (int e = 0; e < KS; ++e) qf[e] *= scale;
    attn_wave<HD, STAGED, TPC, false>(qf, nullptr, Ks, Vs, Ms, qkv, mask_add, ctx, b, head, qb, L, Lp, H, i, h);
}

// Persistent form of the staged kernel for the shapes the encoder runs at full batches (sequences of up to NW query
// blocks whose K and V a thread stages as PF pieces each): a workgroup walks (batch, head) items blockIdx.x,
// + gridDim.x, ...; the next item's K / V pieces and mask are requested into registers as soon as the current item's
// are in LDS and fly under its MFMAs, and each wave requests its next query rows when the last chunk's scores are out.
// Between items: one barrier (everyone is done reading), the LDS writes, one barrier.  Why: with one item per
// workgroup 19 % of the wave slots stood empty (SQ_WAVE_CYCLES against 4 x 1024 SIMDs x kernel cycles: a 75 KB
// workgroup starts only when all eight waves of its predecessor have ended), and a new workgroup's staging round trip
// came on top.
template <int HD, int NW, int TPC, int PF, bool AHEAD>
__global__ __launch_bounds__(NW * 64, HD == 32 ? 4 : 2) void attention_persist_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ mask_add,
                                                                float* __restrict__ ctx, int L, int Lp, int H, int heads,
                                                                int n_items) {
    constexpr int KS = HD / 2;
    constexpr int LDK = HD + 4;
    constexpr int C4 = HD / 4;
    constexpr float kLog2e = 1.44269504088896340736f;
    extern __shared__ float kv_lds[];  // [Lp][LDK] keys, [Lp][LDK] values, [Lp] mask * log2(e)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int H3 = 3 * H;
    float* Ks = kv_lds;
    float* Vs = kv_lds + (size_t)Lp * LDK;
    float* Ms = Vs + (size_t)Lp * LDK;
    const int total = Lp * C4, real = L * C4;  // host: total <= PF * NW * 64, Lp <= NW * 64, Lp <= 32 * NW
    const int qb = wave;
    const int qrow = min(qb * 32 + i, L - 1);
    const float scale = kLog2e / sqrtf((float)HD);

    f32x4 kk[PF], vv[PF];
    float mk;
    auto fetch = [&](int item) {
        const int b = item / heads, head = item - b * heads;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = min((int)threadIdx.x + u * NW * 64, real - 1);
            const int key = e / C4, c4 = e - key * C4;
            const float* src = qkv + (size_t)(b * L + key) * H3 + H + head * HD + c4 * 4;
            kk[u] = *(const f32x4*)src;
            vv[u] = *(const f32x4*)(src + H);
        }
        mk = mask_add[(size_t)b * Lp + min((int)threadIdx.x, Lp - 1)];
    };
    auto q_ptr = [&](int item) {
        const int b = item / heads, head = item - b * heads;
        return qkv + (size_t)(b * L + qrow) * H3 + head * HD + KS * h;
    };
    int item = blockIdx.x;  // host: gridDim.x <= n_items
    if (AHEAD) fetch(item);
    float qf[KS];
    {
        const float* qp = q_ptr(item);
#pragma unroll
        for (int v = 0; v < KS / 4; ++v) {
            const float4 x = *(const float4*)(qp + 4 * v);
            qf[4 * v] = x.x; qf[4 * v + 1] = x.y; qf[4 * v + 2] = x.z; qf[4 * v + 3] = x.w;
        }
    }
    while (item < n_items) {  // workgroup-uniform
        const int b = item / heads, head = item - b * heads;
        if (!AHEAD) fetch(item);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = threadIdx.x + u * NW * 64;
            if (e < total) {
                const int key = e / C4, c4 = e - key * C4;
                const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                *(f32x4*)&Ks[key * LDK + c4 * 4] = e < real ? kk[u] : zero;
                *(f32x4*)&Vs[key * LDK + c4 * 4] = e < real ? vv[u] : zero;
            }
        }
        if ((int)threadIdx.x < Lp) Ms[threadIdx.x] = mk * kLog2e;
#pragma unroll
        for (int e = 0; e < KS; ++e) qf[e] *= scale;
        __syncthreads();
        const int next = item + gridDim.x;
        const int ahead = next < n_items ? next : item;  // (the last item re-requests itself: no branch around loads)
        if (AHEAD) fetch(ahead);
        if (qb * 32 < L)
            attn_wave<HD, true, TPC, true>(qf, q_ptr(ahead), Ks, Vs, Ms, qkv, mask_add, ctx, b, head, qb, L, Lp, H, i, h);
        __syncthreads();  // every wave is done with this item's K and V
        item = next;
    }
}

// ------------------------------------------------------------------------------------------------
// pooling + normalisation (worker.rs:88-103), one workgroup per sequence
// ------------------------------------------------------------------------------------------------
template <int NWV = 4>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) r += red[w];
    __syncthreads();
    return r;
}

// NWV waves, TG tokens requested together per wave, MPL columns per lane (H <= 64 * MPL)
template <int NWV, int TG, int MPL>
__global__ __launch_bounds__(NWV * 64) void pool_kernel(const float* __restrict__ hidden, const float* __restrict__ mask01,
                                                        int L, int H, int mode, int normalize, float* __restrict__ out) {
    // The waves split the tokens (wave w: l = w, w + NWV, ...), every lane owning H/64 columns, and meet in LDS.  The
    // walk over the tokens is a chain of memory round trips: what counts is how many tokens a workgroup has in flight
    // (one column per thread walking all L tokens: 124 us for 256 documents of 256 tokens; four waves x four tokens:
    // 46 us whatever the batch — a fortieth of a 32 x 256 forward; eight waves x eight tokens at MiniLM's width: ~10 us).
    constexpr int kMaxPerLane = MPL;
    __shared__ float red[NWV];
    __shared__ float part[NWV][64 * MPL];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* hb = hidden + (size_t)b * L * H;
    const float* mb = mask01 + (size_t)b * L;
    float msum = 0.0f;
    for (int l = tid; l < L; l += NWV * 64) msum += mb[l];
    msum = block_sum<NWV>(msum, red);
    const float den = fmaxf(msum, 1e-9f);  // clamp_min(1e-9) of rust-bert's mean pooling
    const int npl = (H + 63) / 64;
    if (mode != PCV_POOL_CLS) {
        float acc[kMaxPerLane];
#pragma unroll
        for (int j = 0; j < kMaxPerLane; ++j) acc[j] = mode == PCV_POOL_MAX ? -__builtin_inff() : 0.0f;
        // four tokens' loads are requested together (one token per iteration ran at one memory round trip per
        // token: 124 us for 256 documents of 256 tokens); accumulation order unchanged
        for (int l0 = wave; l0 < L; l0 += NWV * TG) {
            float m[TG], x[TG][kMaxPerLane];
#pragma unroll
            for (int g = 0; g < TG; ++g) {
                const int l = l0 + NWV * g;
                const bool live = l < L;
                m[g] = live ? mb[l] : 0.0f;
                const float* row = hb + (size_t)(live ? l : 0) * H;
#pragma unroll
                for (int j = 0; j < kMaxPerLane; ++j) {
                    const int c = lane + 64 * j;
                    x[g][j] = (live && j < npl && c < H) ? row[c] : 0.0f;
                }
            }
#pragma unroll
            for (int g = 0; g < TG; ++g) {
                if (l0 + NWV * g >= L) break;
#pragma unroll
                for (int j = 0; j < kMaxPerLane; ++j) {
                    const int c = lane + 64 * j;
                    if (j < npl && c < H)
                        acc[j] = mode == PCV_POOL_MAX ? fmaxf(acc[j], m[g] != 0.0f ? x[g][j] : -1e9f) : acc[j] + x[g][j] * m[g];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxPerLane; ++j) {
            const int c = lane + 64 * j;
            if (j < npl && c < H) part[wave][c] = acc[j];
        }
        __syncthreads();
    }
    constexpr int VPT = (MPL + NWV - 1) / NWV;  // output columns per thread
    float vals[VPT];
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = tid + NWV * 64 * j;
        float v = 0.0f;
        if (c < H) {
            if (mode == PCV_POOL_CLS) {
                v = hb[c];
            } else if (mode == PCV_POOL_MAX) {
                v = part[0][c];
#pragma unroll
                for (int w = 1; w < NWV; ++w) v = fmaxf(v, part[w][c]);
            } else {
                float s = part[0][c];
#pragma unroll
                for (int w = 1; w < NWV; ++w) s += part[w][c];
                v = mode == PCV_POOL_MEAN_SQRT_LEN ? s / sqrtf(den) : s / den;
            }
        }
        vals[j] = v;
        ss += c < H ? v * v : 0.0f;
    }
    float inv = 1.0f;
    if (normalize) {  // x / clamp_min(||x||, 1e-12), worker.rs:95-103
        const float nrm = sqrtf(block_sum<NWV>(ss, red));
        inv = 1.0f / fmaxf(nrm, 1e-12f);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = tid + NWV * 64 * j;
        if (c < H) out[(size_t)b * H + c] = normalize ? vals[j] * inv : vals[j];
    }
}

// Dense module (worker.rs:90-94): one workgroup per sequence, thread per output feature
__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ bvec, int in, int out, int act,
                                                    int normalize, float* __restrict__ y) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * in;
    float vals[4];
    float ss = 0.0f;
    for (int j = 0; j < 4; ++j) {
        const int o = tid + 256 * j;
        float v = 0.0f;
        if (o < out) {
            const float* wr = W + (size_t)o * in;
            for (int c = 0; c < in; ++c) v = fmaf(xb[c], wr[c], v);
            v += bvec ? bvec[o] : 0.0f;
            if (act == PCV_ACT_TANH) v = tanhf(v);
            ss += v * v;
        }
        vals[j] = v;
    }
    float inv = 1.0f;
    if (normalize) inv = 1.0f / fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    for (int j = 0; j < 4; ++j) {
        const int o = tid + 256 * j;
        if (o < out) y[(size_t)b * out + o] = normalize ? vals[j] * inv : vals[j];
    }
}

__global__ __launch_bounds__(256) void synth_weights_kernel(float* __restrict__ dst, int64_t n, uint64_t seed,
                                                            uint32_t tensor_index, float scale, float offset) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one float4 piece per thread
    if (p * 4 >= n) return;
    const float4 v = synth_piece(seed, (int64_t)tensor_index, (uint32_t)p);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int j = 0; j < 4; ++j)
        if (p * 4 + j < n) dst[p * 4 + j] = offset + scale * vv[j];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
template <int MT, int NW>
static void launch_skinny(hipStream_t st, const float* A, const float* W, const float* bias, const float* resid,
                          float* C, int M, int N, int K, int epilogue) {
    const dim3 grid(N / 32);
    switch (epilogue) {
        case EPI_BIAS_GELU:
            gemm_skinny_f32_kernel<EPI_BIAS_GELU, MT, NW><<<grid, NW * 64, 0, st>>>(A, W, bias, resid, C, M, N, K);
            break;
        case EPI_BIAS_GELU_TANH:
            gemm_skinny_f32_kernel<EPI_BIAS_GELU_TANH, MT, NW><<<grid, NW * 64, 0, st>>>(A, W, bias, resid, C, M, N, K);
            break;
        case EPI_BIAS_RESIDUAL:
            gemm_skinny_f32_kernel<EPI_BIAS_RESIDUAL, MT, NW><<<grid, NW * 64, 0, st>>>(A, W, bias, resid, C, M, N, K);
            break;
        default: gemm_skinny_f32_kernel<EPI_BIAS, MT, NW><<<grid, NW * 64, 0, st>>>(A, W, bias, resid, C, M, N, K); break;
    }
}

bool launch_gemm_f32_ln(hipStream_t st, const float* A, const float* W, const float* bias, const float* resid, const float* ln_w,
                        const float* ln_b, float eps, float* C, int M, int N, int K) {
    if (M <= 128 || (N != 128 && N != 256 && N != 384) || K % BK != 0 || bias == nullptr || resid == nullptr) return false;
    const size_t lds = (size_t)(LBM + N) * LDT * sizeof(float);
    const dim3 grid((M + LBM - 1) / LBM);
    static const bool no32 = getenv("PCV_NO_LN32") != nullptr;  // (comparison switch)
    static const int max32 = getenv("PCV_LN32_MAX_M") ? atoi(getenv("PCV_LN32_MAX_M")) : -1;  // (tuning)
    if (N == 384 && K % 64 == 0 && !no32) {
        // 32-row tiles where they fill the chip better than 64-row ones: a CU's time is its number of tiles, a 32-row tile
        // costing 0.6 of a 64-row one (half the work, at the lower reuse of W).  8 192 tokens: 1 x 0.6 against 1 (half the CUs
        // idle); 16 384: 2 x 0.6 against 1; 24 576: 3 x 0.6 against 2 (5.06 against 5.49 ms a forward); 65 536: 8 x 0.6 against 4
        const int cus = current_device_cus();
        const int tiles32 = (M + L32M - 1) / L32M, tiles64 = (M + LBM - 1) / LBM;
        const bool use32 = max32 >= 0 ? M <= max32 : 6 * ((tiles32 + cus - 1) / cus) < 10 * ((tiles64 + cus - 1) / cus);
        if (use32) {
            const size_t lds32 = ((size_t)8 * (32 + 96) * LDT + 3 * N) * sizeof(float);  // 152 KB of dynamic LDS: one workgroup per CU
            allow_dynamic_lds((const void*)gemm_f32_ln32_kernel<3>, lds32);
            gemm_f32_ln32_kernel<3><<<dim3(std::min<int>(tiles32, cus)), 512, lds32, st>>>(A, W, bias, resid, ln_w, ln_b, eps, C, M, K);
            return true;
        }
    }
    if (N == 384) {  // persistent eight-wave form: 1 % of a 256 x 256 forward over the four-wave one
        const size_t lds8 = lds + (size_t)3 * N * sizeof(float);  // + bias and the LayerNorm parameters
        allow_dynamic_lds((const void*)gemm_f32_ln8_kernel<3>, lds8);  // 69 KB of dynamic LDS
        const int resident = 2 * 256;  // two workgroups per CU
        gemm_f32_ln8_kernel<3><<<dim3(std::min<int>(grid.x, resident)), 512, lds8, st>>>(A, W, bias, resid, ln_w, ln_b, eps, C, M, K);
        return true;
    }
    switch (N / 128) {
        case 1: gemm_f32_ln_kernel<1><<<grid, 256, lds, st>>>(A, W, bias, resid, ln_w, ln_b, eps, C, M, K); break;
        default: gemm_f32_ln_kernel<2><<<grid, 256, lds, st>>>(A, W, bias, resid, ln_w, ln_b, eps, C, M, K); break;
    }
    return true;
}

void launch_gemm_f32(hipStream_t st, const float* A, const float* W, const float* bias, const float* resid, float* C,
                     int M, int N, int K, int epilogue) {
    if (M <= 0) return;
    if (M <= 128 && K % 128 == 0) {  // single queries / a few short chunks
        if (M <= 32)
            launch_skinny<1, 8>(st, A, W, bias, resid, C, M, N, K, epilogue);
        else if (M <= 64)
            launch_skinny<2, 8>(st, A, W, bias, resid, C, M, N, K, epilogue);
        else
            launch_skinny<4, 4>(st, A, W, bias, resid, C, M, N, K, epilogue);
        return;
    }
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU) && N % BN96 == 0) {
        // 96-column tiles where they deal the CUs a lighter worst case (a 96-column tile = 0.75 of a 128-column one; ties keep
        // the wider tile and its better operand reuse): 8 192 x 1 152 -> 3 x 0.75 against 3, 65 536 x 1 152 -> 24 x 0.75 against 18
        static const bool no96 = getenv("PCV_NO_N96") != nullptr;  // (comparison switch)
        const long cus = current_device_cus(), mt = (M + BM - 1) / BM;
        const long t128 = mt * (N / BN), t96 = mt * (N / BN96);
        const long c96 = 3 * ((t96 + cus - 1) / cus), c128 = 4 * ((t128 + cus - 1) / cus);
        if (!no96 && N % BN == 0 && c96 < c128) {  // (at ties the 96-column tiles measured 1-2 % slower: 90.6 / 89.4 us, 601 / 588 us)
            if (epilogue == EPI_BIAS) gemm_f32_n96_kernel<EPI_BIAS><<<dim3((unsigned)t96), 256, 0, st>>>(A, W, bias, C, M, N, K);
            else gemm_f32_n96_kernel<EPI_BIAS_GELU><<<dim3((unsigned)t96), 256, 0, st>>>(A, W, bias, C, M, N, K);
            return;
        }
    }
    dim3 grid((N / BN) * ((M + BM - 1) / BM));
    switch (epilogue) {
        case EPI_BIAS_GELU: gemm_f32_kernel<EPI_BIAS_GELU><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K); break;
        case EPI_BIAS_GELU_TANH: gemm_f32_kernel<EPI_BIAS_GELU_TANH><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K); break;
        case EPI_BIAS_RESIDUAL:
            gemm_f32_kernel<EPI_BIAS_RESIDUAL><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K);
            break;
        default: gemm_f32_kernel<EPI_BIAS><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K); break;
    }
}

template <int EPI>
static void launch_gemm_bf16x3_ws(hipStream_t st, const float* A, const uint16_t* Wh, const uint16_t* Wm,
                                  const uint16_t* Wl, const float* bias, const float* resid, float* C, int M, int N, int K) {
    allow_dynamic_lds((const void*)gemm_bf16x3_ws_kernel<EPI>, kGemmWsLdsBytes);
    const int num_cus = current_device_cus();
    const unsigned ntiles = (unsigned)(N / BN) * (unsigned)((M + BM - 1) / BM);
    const unsigned grid = ntiles < (unsigned)num_cus ? ntiles : (unsigned)num_cus;  // one workgroup per CU (120 KB of LDS each)
    gemm_bf16x3_ws_kernel<EPI><<<grid, 512, kGemmWsLdsBytes, st>>>(A, Wh, Wm, Wl, bias, resid, C, M, N, K);
}

void launch_gemm_bf16x3(hipStream_t st, const float* A, const uint16_t* Wh, const uint16_t* Wm, const uint16_t* Wl,
                        const float* bias, const float* resid, float* C, int M, int N, int K, int epilogue) {
    if (M <= 0) return;
    dim3 grid((N / BN) * ((M + BM - 1) / BM));
    // the persistent wave-specialised form measures the same as the plain one (both are bound by LDS
    // traffic); kept selectable as the skeleton for a wider-tile version
    static const bool ws = getenv("PCV_GEMM_WS") && getenv("PCV_GEMM_WS")[0] == '1';
    if (ws) {
        switch (epilogue) {
            case EPI_BIAS_GELU: launch_gemm_bf16x3_ws<EPI_BIAS_GELU>(st, A, Wh, Wm, Wl, bias, resid, C, M, N, K); break;
            case EPI_BIAS_GELU_TANH: launch_gemm_bf16x3_ws<EPI_BIAS_GELU_TANH>(st, A, Wh, Wm, Wl, bias, resid, C, M, N, K); break;
            case EPI_BIAS_RESIDUAL: launch_gemm_bf16x3_ws<EPI_BIAS_RESIDUAL>(st, A, Wh, Wm, Wl, bias, resid, C, M, N, K); break;
            default: launch_gemm_bf16x3_ws<EPI_BIAS>(st, A, Wh, Wm, Wl, bias, resid, C, M, N, K); break;
        }
        return;
    }
    switch (epilogue) {
        case EPI_BIAS_GELU:
            gemm_bf16x3_kernel<EPI_BIAS_GELU><<<grid, 256, 0, st>>>(A, Wh, Wm, Wl, bias, resid, C, M, N, K);
            break;
        case EPI_BIAS_GELU_TANH:
            gemm_bf16x3_kernel<EPI_BIAS_GELU_TANH><<<grid, 256, 0, st>>>(A, Wh, Wm, Wl, bias, resid, C, M, N, K);
            break;
        case EPI_BIAS_RESIDUAL:
            gemm_bf16x3_kernel<EPI_BIAS_RESIDUAL><<<grid, 256, 0, st>>>(A, Wh, Wm, Wl, bias, resid, C, M, N, K);
            break;
        default: gemm_bf16x3_kernel<EPI_BIAS><<<grid, 256, 0, st>>>(A, Wh, Wm, Wl, bias, resid, C, M, N, K); break;
    }
}

template <int NSET>
static void launch_gemm_f16x2_n(hipStream_t st, dim3 grid, const float* A, const uint16_t* Wh, const uint16_t* Wl,
                                const float* bias, const float* resid, float* C, int M, int N, int K, int epilogue) {
    switch (epilogue) {
        case EPI_BIAS_GELU: gemm_f16x2_kernel<EPI_BIAS_GELU, NSET><<<grid, 256, 0, st>>>(A, Wh, Wl, bias, resid, C, M, N, K); break;
        case EPI_BIAS_GELU_TANH: gemm_f16x2_kernel<EPI_BIAS_GELU_TANH, NSET><<<grid, 256, 0, st>>>(A, Wh, Wl, bias, resid, C, M, N, K); break;
        case EPI_BIAS_RESIDUAL:
            gemm_f16x2_kernel<EPI_BIAS_RESIDUAL, NSET><<<grid, 256, 0, st>>>(A, Wh, Wl, bias, resid, C, M, N, K);
            break;
        default: gemm_f16x2_kernel<EPI_BIAS, NSET><<<grid, 256, 0, st>>>(A, Wh, Wl, bias, resid, C, M, N, K); break;
    }
}

void launch_gemm_f16x2(hipStream_t st, const float* A, const uint16_t* Wh, const uint16_t* Wl, const float* bias,
                       const float* resid, float* C, int M, int N, int K, int epilogue) {
    if (M <= 0) return;
    dim3 grid((N / BN) * ((M + BM - 1) / BM));
    // one register set: 144 VGPRs and 41 KB of LDS let three workgroups share a CU (8.44 ms per 256x256-token
    // forward against 8.69 ms with two sets at two workgroups per CU)
    launch_gemm_f16x2_n<1>(st, grid, A, Wh, Wl, bias, resid, C, M, N, K, epilogue);
}

void launch_split_planes_f16(hipStream_t st, const float* src, int64_t n, uint16_t* hi, uint16_t* lo, int* d_overflow) {
    if (n <= 0) return;
    split_planes_f16_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(src, n, hi, lo, d_overflow);
}

void launch_split_planes(hipStream_t st, const float* src, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo) {
    if (n <= 0) return;
    split_planes_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(src, n, hi, mid, lo);
}

void launch_embed_ln(hipStream_t st, const int64_t* ids, const int64_t* mask, int B, int L, int H, int vocab,
                     const float* word, const float* pos, const float* type, const float* ln_w, const float* ln_b,
                     float eps, float* hidden, float* mask_add, float* mask01) {
    const int Lp = (L + 31) / 32 * 32;
    embed_ln_kernel<<<(B * Lp + 3) / 4, 256, 0, st>>>(ids, mask, B, L, Lp, H, vocab, word, pos, type, ln_w, ln_b, eps,
                                                      hidden, mask_add, mask01);
}

void launch_layer_norm(hipStream_t st, float* x, int T, int H, const float* w, const float* b, float eps) {
    constexpr int TOK = 2;
    const unsigned grid = (unsigned)((T + 4 * TOK - 1) / (4 * TOK));
    switch (T >= 64 ? H : 0) {  // a handful of tokens: one token per wave spreads over more CUs
        case 384: layer_norm_fixed_kernel<6, TOK><<<grid, 256, 0, st>>>(x, T, w, b, eps); return;
        case 768: layer_norm_fixed_kernel<12, TOK><<<grid, 256, 0, st>>>(x, T, w, b, eps); return;
        case 1024: layer_norm_fixed_kernel<16, TOK><<<grid, 256, 0, st>>>(x, T, w, b, eps); return;
        default: break;
    }
    layer_norm_kernel<<<(T + 3) / 4, 256, 0, st>>>(x, T, H, w, b, eps);
}

// head_dim 32 / 64 with K and V^T planes resident in LDS; false when they do not fit (caller falls back to f32)
template <int HD, int NW, int TPC>
static void launch_attention_f16_v(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int Lp, int H,
                                   int heads, size_t lds) {
    allow_dynamic_lds((const void*)attention_f16_kernel<HD, NW, TPC>, lds);  // (static LDS comes on top)
    dim3 grid((Lp / 32 + NW - 1) / NW, heads, B);
    attention_f16_kernel<HD, NW, TPC><<<grid, NW * 64, lds, st>>>(qkv, mask_add, ctx, B, L, Lp, H);
}

bool launch_attention_f16(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int H,
                          int heads) {
    const int Lp = (L + 31) / 32 * 32;
    const int HD = H / heads;
    if (HD != 32 && HD != 64) return false;
    const size_t lds = (size_t)2 * Lp * (HD * 2 + 16) + (size_t)2 * HD * (Lp * 2 + 16);
    if (lds > 150 * 1024) return false;
    // 64-key softmax chunks keep the kernel at 96 VGPRs (4+ waves per SIMD); eight waves per workgroup stage
    // K / V^T once for 256 queries (measured on 256 x 256 tokens: 7.29 ms per forward against 7.92 ms with
    // four waves and 128-key chunks)
    const bool wide = Lp >= 256;
    if (HD == 32) {
        if (wide) launch_attention_f16_v<32, 8, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
        else launch_attention_f16_v<32, 4, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
    } else {
        if (wide) launch_attention_f16_v<64, 8, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
        else launch_attention_f16_v<64, 4, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
    }
    return true;
}

template <int HD, int NW, int TPC>
static void launch_attention_staged(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int Lp,
                                    int H, int heads, size_t lds) {
    allow_dynamic_lds((const void*)attention_kernel<HD, true, NW, TPC>, lds);
    dim3 grid((Lp / 32 + NW - 1) / NW, heads, B);
    attention_kernel<HD, true, NW, TPC><<<grid, NW * 64, lds, st>>>(qkv, mask_add, ctx, B, L, Lp, H);
}

// (batch, head) items walked by as many workgroups as are resident at once
template <int HD, int NW, int TPC, int PF>
static void launch_attention_persist(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int Lp,
                                     int H, int heads, size_t lds) {
    constexpr bool AHEAD = HD == 64;  // (head_dim 32 would run two workgroups per CU at 128 registers: no room for the pieces in flight)
    const void* fn = (const void*)attention_persist_kernel<HD, NW, TPC, PF, AHEAD>;
    allow_dynamic_lds(fn, lds);
    const int per_cu = std::max(1, std::min<int>(HD == 32 ? 2 : 1, (int)((160 * 1024) / lds)));  // LDS, and 128 / 256 registers
    const int n_items = B * heads;
    const int grid = std::min(n_items, current_device_cus() * per_cu);
    attention_persist_kernel<HD, NW, TPC, PF, AHEAD><<<grid, NW * 64, lds, st>>>(qkv, mask_add, ctx, L, Lp, H, heads, n_items);
}

void launch_attention(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int H,
                      int heads) {
    const int Lp = (L + 31) / 32 * 32;
    const int HD = H / heads;
    const size_t lds = ((size_t)2 * Lp * (HD + 4) + Lp) * sizeof(float);  // K, V (rows padded to whole 32-key tiles), mask
    const bool staged = lds <= 150 * 1024;
    if (staged) {  // eight waves + 64-key chunks where the sequence has eight query tiles (13.61 vs 13.93 ms per forward)
        const bool wide = Lp >= 256;
        static const bool no_persist = getenv("PCV_ATTN_NO_PERSIST") != nullptr;
        // head_dim 64 at 256 keys (BERT-base): one 139 KB workgroup per CU, 244 registers: the persistent form with the
        // next item's pieces in flight, 128 -> 117 us per call (64 x 256 tokens).  head_dim 32 (two 75 KB workgroups
        // per CU, 128 registers: no room for pieces in flight) measured 232 us persistent against 224 us: not used; nor with
        // 32-key chunks, which leave room for the pieces (128 registers, 20 B of scratch): 236 us (round 4).
        if (Lp == 256 && HD == 64 && !no_persist) {
            launch_attention_persist<64, 8, 2, 8>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
            return;
        }
        if (HD == 32) {
            if (wide) launch_attention_staged<32, 8, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
            else launch_attention_staged<32, 4, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
        } else {
            if (wide) launch_attention_staged<64, 8, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
            else launch_attention_staged<64, 4, 2>(st, qkv, mask_add, ctx, B, L, Lp, H, heads, lds);
        }
        return;
    }
    dim3 grid((Lp / 32 + 3) / 4, heads, B);
    if (HD == 32)
        attention_kernel<32, false><<<grid, 256, 0, st>>>(qkv, mask_add, ctx, B, L, Lp, H);
    else
        attention_kernel<64, false><<<grid, 256, 0, st>>>(qkv, mask_add, ctx, B, L, Lp, H);
}

void launch_pool(hipStream_t st, const float* hidden, const float* mask01, int B, int L, int H, int mode,
                 int normalize, float* out) {
    if (H <= 512)
        pool_kernel<8, 8, 8><<<B, 512, 0, st>>>(hidden, mask01, L, H, mode, normalize, out);
    else
        pool_kernel<8, 4, 16><<<B, 512, 0, st>>>(hidden, mask01, L, H, mode, normalize, out);
}

void launch_dense(hipStream_t st, const float* x, const float* W, const float* b, int B, int in, int out, int act,
                  int normalize, float* y) {
    dense_kernel<<<B, 256, 0, st>>>(x, W, b, in, out, act, normalize, y);
}

void launch_synth_weights(hipStream_t st, float* dst, int64_t n, uint64_t seed, uint32_t tensor_index, float scale,
                          float offset) {
    if (n <= 0) return;
    synth_weights_kernel<<<(unsigned)((n / 4 + 256) / 256), 256, 0, st>>>(dst, n, seed, tensor_index, scale, offset);
}


namespace {
// One workgroup per document; each wave takes every fourth chunk, the 64 lanes share a chunk's dot product.
__global__ __launch_bounds__(256) void chunk_argmax_kernel(const float* __restrict__ query, const float* __restrict__ chunks, int D,
                                                           const int32_t* __restrict__ bounds, int32_t* __restrict__ best,
                                                           int32_t* __restrict__ nan_flag) {
    extern __shared__ float sqv[];  // [D]
    __shared__ float w_s[4];
    __shared__ int32_t w_c[4];
    __shared__ int32_t w_nan[4];
    const int doc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < D; i += 256) sqv[i] = query[i];
    __syncthreads();
    const int c0 = bounds[doc], c1 = bounds[doc + 1];
    float bs = -__builtin_inff();
    int bc = -1, nan = 0;
    for (int c = c0 + wave; c < c1; c += 4) {
        const float* x = chunks + (size_t)c * D;
        float part = 0.0f;
        for (int i = lane; i < D; i += 64) part = fmaf(sqv[i], x[i], part);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (!(part == part)) nan = 1;
        if (bc < 0 || part >= bs) {  // chunks come in ascending order: >= keeps the last of equal maxima
            bs = part;
            bc = c;
        }
    }
    if (lane == 0) {
        w_s[wave] = bs;
        w_c[wave] = bc;
        w_nan[wave] = nan;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        int c = -1, f = 0;
        for (int w = 0; w < 4; ++w) {
            f |= w_nan[w];
            if (w_c[w] < 0) continue;
            if (c < 0 || w_s[w] > s || (w_s[w] == s && w_c[w] > c)) {
                s = w_s[w];
                c = w_c[w];
            }
        }
        best[doc] = c;
        nan_flag[doc] = f;
    }
}
}  // namespace

void launch_chunk_argmax(hipStream_t st, const float* query, const float* chunks, int D, const int32_t* bounds, int n_docs,
                         int32_t* best, int32_t* nan_flag) {
    if (n_docs <= 0) return;
    chunk_argmax_kernel<<<n_docs, 256, (size_t)D * sizeof(float), st>>>(query, chunks, D, bounds, best, nan_flag);
}

}  // namespace pcv

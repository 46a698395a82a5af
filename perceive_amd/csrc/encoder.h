// encoder.h — launcher interface of the sentence-embedding encoder kernels (encoder_kernels.hip).
// Replaces the libtorch forward behind crates/perceive-core/model/worker.rs:78-106.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/perceive_hip.h"

namespace pcv {

enum GemmEpilogue { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESIDUAL = 2, EPI_BIAS_GELU_TANH = 3 };

// C[M][N] = A[M][K] * W[N][K]^T + bias (+ GELU | + resid).  N % 128 == 0, K % 32 == 0.
// exact f32 on v_mfma_f32_32x32x2_f32.
void launch_gemm_f32(hipStream_t st, const float* A, const float* W, const float* bias, const float* resid, float* C,
                     int M, int N, int K, int epilogue);

// out = LayerNorm(A W^T + bias + resid) * ln_w + ln_b in one kernel (N = hidden in {128, 256, 384}, M > 128, exact
// f32); false when the shape is not covered: the caller then runs launch_gemm_f32 + launch_layer_norm.
bool launch_gemm_f32_ln(hipStream_t st, const float* A, const float* W, const float* bias, const float* resid, const float* ln_w,
                        const float* ln_b, float eps, float* C, int M, int N, int K);

// Same contraction with every f32 operand split into three bf16 terms (x = hi + mid + lo): six
// v_mfma_f32_32x32x16_bf16 per product, f32 accumulate.  Wp = the three pre-split weight planes,
// each [N][K] bf16; A is split on the fly while it is staged into LDS.
void launch_gemm_bf16x3(hipStream_t st, const float* A, const uint16_t* Wh, const uint16_t* Wm, const uint16_t* Wl,
                        const float* bias, const float* resid, float* C, int M, int N, int K, int epilogue);
// C = A W^T with two-term f16 splits (three v_mfma_f32_32x32x16_f16 per product); Wh/Wl = planes of 2^8 * W
void launch_gemm_f16x2(hipStream_t st, const float* A, const uint16_t* Wh, const uint16_t* Wl, const float* bias,
                       const float* resid, float* C, int M, int N, int K, int epilogue);
// f32 [n] -> two f16 planes of 2^8 * x; *d_overflow (device int) is set if a weight does not fit
void launch_split_planes_f16(hipStream_t st, const float* src, int64_t n, uint16_t* hi, uint16_t* lo, int* d_overflow);
// f32 [n] -> three bf16 planes
void launch_split_planes(hipStream_t st, const float* src, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo);

// tokens -> embeddings + LayerNorm; also converts the int64 mask to the additive float mask
// (1-m)*-10000 used by the attention kernel and to a float 0/1 mask for pooling.
void launch_embed_ln(hipStream_t st, const int64_t* ids, const int64_t* mask, int B, int L, int H, int vocab,
                     const float* word, const float* pos, const float* type, const float* ln_w, const float* ln_b,
                     float eps, float* hidden, float* mask_add, float* mask01);

// in-place LayerNorm over rows of x[T][H]
void launch_layer_norm(hipStream_t st, float* x, int T, int H, const float* w, const float* b, float eps);

// softmax(QK^T/sqrt(hd) + mask) V for every (batch, head): qkv[T][3H] -> ctx[T][H]; head_dim 32 or 64
void launch_attention(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int H,
                      int heads);

// the same with two-term f16 splits (PCV_COMPUTE_F16X2); returns false if the shape is not covered
bool launch_attention_f16(hipStream_t st, const float* qkv, const float* mask_add, float* ctx, int B, int L, int H,
                          int heads);

// pooling (mean | cls | max | mean_sqrt_len) + optional L2 normalisation (worker.rs:88-103)
void launch_pool(hipStream_t st, const float* hidden, const float* mask01, int B, int L, int H, int mode,
                 int normalize, float* out);
// optional Dense module: y = act(W x + b), then optional normalisation
void launch_dense(hipStream_t st, const float* x, const float* W, const float* b, int B, int in, int out, int act,
                  int normalize, float* y);

// highlight.rs:109-127 on the device: score of chunk c = dot(query, chunks[c]) in f32 (lib.rs:63-65); per
// document d the LAST best-scoring chunk of [bounds[d], bounds[d+1]) goes to best[d] (-1: no chunk;
// itertools' position_max_by keeps the last maximum), nan_flag[d] != 0 if one of its scores is NaN.
void launch_chunk_argmax(hipStream_t st, const float* query, const float* chunks, int D, const int32_t* bounds, int n_docs,
                         int32_t* best, int32_t* nan_flag);

void launch_synth_weights(hipStream_t st, float* dst, int64_t n, uint64_t seed, uint32_t tensor_index, float scale,
                          float offset);

}  // namespace pcv

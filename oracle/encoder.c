/*
 * encoder.c — CPU restatement of the sentence-embedding forward pass.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h: parity unpinned by the reference; cross-checked against Hugging Face BertModel on
 * CPU with seeded random weights, tests/golden/gen_encoder_golden.py).
 *
 * Restates crates/perceive-core/model/worker.rs:78-106:
 *   transformer.forward(ids, mask)            worker.rs:85-86   (rust-bert BertModel: embeddings +
 *                                             LayerNorm, N x [self-attention, add&norm, GELU FFN,
 *                                             add&norm]; additive mask (1-m)*-10000; f32, no_grad)
 *   pooling_layer.forward(tokens, mask)       worker.rs:88-89   (mean: sum(h*m)/clamp_min(sum m,1e-9))
 *   dense_layer.forward (optional)            worker.rs:90-94
 *   x / clamp_min(||x||_2, 1e-12)             worker.rs:95-103
 * Token ids / masks are laid out as generate_token_tensors does (tokenize.rs:13-51).
 * Everything is plain f32 loops in index order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct {
    int vocab, hidden, layers, heads, inter, max_pos, type_vocab;
    float eps;
    int pooling, normalize, dense_out, dense_act;
} orc_model_desc;

/* weights: one flat array per tensor, PyTorch layout (Linear weight = [out][in]) */
typedef struct {
    const float *word, *pos, *type, *emb_ln_w, *emb_ln_b;
    const float **qw, **qb, **kw, **kb, **vw, **vb, **ow, **ob, **ln1w, **ln1b;
    const float **iw, **ib, **fw, **fb, **ln2w, **ln2b;
    const float *dense_w, *dense_b;
} orc_weights;

static void layer_norm(float* x, int H, const float* w, const float* b, float eps) {
    float mean = 0.0f;
    for (int i = 0; i < H; ++i) mean += x[i];
    mean /= (float)H;
    float var = 0.0f;
    for (int i = 0; i < H; ++i) {
        float d = x[i] - mean;
        var += d * d;
    }
    var /= (float)H;
    float inv = 1.0f / sqrtf(var + eps);
    for (int i = 0; i < H; ++i) x[i] = (x[i] - mean) * inv * w[i] + b[i];
}

/* y[t][o] = sum_i x[t][i] * W[o][i] + b[o] */
static void linear(const float* x, int T, int in, const float* W, const float* b, int out, float* y) {
    for (int t = 0; t < T; ++t)
        for (int o = 0; o < out; ++o) {
            float acc = 0.0f;
            const float* xr = x + (size_t)t * in;
            const float* wr = W + (size_t)o * in;
            for (int i = 0; i < in; ++i) acc += xr[i] * wr[i];
            y[(size_t)t * out + o] = acc + (b ? b[o] : 0.0f);
        }
}

static float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

/*
 * ids, mask: [B][L] int64.  hidden_out (optional): [(layers+1)][B][L][H] — embedding output then
 * every layer's output.  out: [B][out_dim].
 */
void orc_encode_tokens(const orc_model_desc* d, const orc_weights* w, const int64_t* ids, const int64_t* mask,
                       int B, int L, float* hidden_out, float* out) {
    const int H = d->hidden, NH = d->heads, HD = H / NH, F = d->inter, T = B * L;
    float* h = (float*)malloc((size_t)T * H * sizeof(float));
    float* q = (float*)malloc((size_t)T * H * sizeof(float));
    float* k = (float*)malloc((size_t)T * H * sizeof(float));
    float* v = (float*)malloc((size_t)T * H * sizeof(float));
    float* ctx = (float*)malloc((size_t)T * H * sizeof(float));
    float* tmp = (float*)malloc((size_t)T * H * sizeof(float));
    float* ff = (float*)malloc((size_t)T * F * sizeof(float));
    float* sc = (float*)malloc((size_t)L * sizeof(float));

    /* embeddings: word + position + token_type(0), LayerNorm */
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < L; ++l) {
            float* x = h + ((size_t)b * L + l) * H;
            const float* we = w->word + (size_t)ids[(size_t)b * L + l] * H;
            const float* pe = w->pos + (size_t)l * H;
            for (int i = 0; i < H; ++i) x[i] = we[i] + pe[i] + w->type[i];
            layer_norm(x, H, w->emb_ln_w, w->emb_ln_b, d->eps);
        }
    if (hidden_out) memcpy(hidden_out, h, (size_t)T * H * sizeof(float));

    const float scale = 1.0f / sqrtf((float)HD);
    for (int ly = 0; ly < d->layers; ++ly) {
        linear(h, T, H, w->qw[ly], w->qb[ly], H, q);
        linear(h, T, H, w->kw[ly], w->kb[ly], H, k);
        linear(h, T, H, w->vw[ly], w->vb[ly], H, v);
        for (int b = 0; b < B; ++b)
            for (int hd = 0; hd < NH; ++hd)
                for (int i = 0; i < L; ++i) {
                    const float* qi = q + ((size_t)b * L + i) * H + hd * HD;
                    float mx = -INFINITY;
                    for (int j = 0; j < L; ++j) {
                        const float* kj = k + ((size_t)b * L + j) * H + hd * HD;
                        float s = 0.0f;
                        for (int e = 0; e < HD; ++e) s += qi[e] * kj[e];
                        s = s * scale + (1.0f - (float)mask[(size_t)b * L + j]) * -10000.0f;
                        sc[j] = s;
                        if (s > mx) mx = s;
                    }
                    float sum = 0.0f;
                    for (int j = 0; j < L; ++j) {
                        sc[j] = expf(sc[j] - mx);
                        sum += sc[j];
                    }
                    float* o = ctx + ((size_t)b * L + i) * H + hd * HD;
                    for (int e = 0; e < HD; ++e) o[e] = 0.0f;
                    for (int j = 0; j < L; ++j) {
                        const float p = sc[j] / sum;
                        const float* vj = v + ((size_t)b * L + j) * H + hd * HD;
                        for (int e = 0; e < HD; ++e) o[e] += p * vj[e];
                    }
                }
        linear(ctx, T, H, w->ow[ly], w->ob[ly], H, tmp);
        for (int t = 0; t < T; ++t) {
            float* x = h + (size_t)t * H;
            for (int i = 0; i < H; ++i) x[i] = tmp[(size_t)t * H + i] + x[i];
            layer_norm(x, H, w->ln1w[ly], w->ln1b[ly], d->eps);
        }
        linear(h, T, H, w->iw[ly], w->ib[ly], F, ff);
        for (size_t i = 0; i < (size_t)T * F; ++i) ff[i] = gelu_erf(ff[i]);
        linear(ff, T, F, w->fw[ly], w->fb[ly], H, tmp);
        for (int t = 0; t < T; ++t) {
            float* x = h + (size_t)t * H;
            for (int i = 0; i < H; ++i) x[i] = tmp[(size_t)t * H + i] + x[i];
            layer_norm(x, H, w->ln2w[ly], w->ln2b[ly], d->eps);
        }
        if (hidden_out) memcpy(hidden_out + (size_t)(ly + 1) * T * H, h, (size_t)T * H * sizeof(float));
    }

    /* pooling (worker.rs:88-89; rust-bert Pooling: mean / cls / max / mean_sqrt_len) */
    const int OD = d->dense_out > 0 ? d->dense_out : H;
    float* pooled = (float*)malloc((size_t)B * H * sizeof(float));
    for (int b = 0; b < B; ++b) {
        float* p = pooled + (size_t)b * H;
        float msum = 0.0f;
        for (int l = 0; l < L; ++l) msum += (float)mask[(size_t)b * L + l];
        if (d->pooling == 1) { /* cls */
            memcpy(p, h + (size_t)b * L * H, (size_t)H * sizeof(float));
        } else if (d->pooling == 2) { /* max over unmasked tokens (masked -> -1e9) */
            for (int i = 0; i < H; ++i) {
                float m = -INFINITY;
                for (int l = 0; l < L; ++l) {
                    float val = mask[(size_t)b * L + l] ? h[((size_t)b * L + l) * H + i] : -1e9f;
                    if (val > m) m = val;
                }
                p[i] = m;
            }
        } else {
            for (int i = 0; i < H; ++i) {
                float s = 0.0f;
                for (int l = 0; l < L; ++l) s += h[((size_t)b * L + l) * H + i] * (float)mask[(size_t)b * L + l];
                float den = msum < 1e-9f ? 1e-9f : msum;
                p[i] = d->pooling == 3 ? s / sqrtf(den) : s / den;
            }
        }
    }
    for (int b = 0; b < B; ++b) {
        float* o = out + (size_t)b * OD;
        if (d->dense_out > 0) { /* worker.rs:90-94 */
            linear(pooled + (size_t)b * H, 1, H, w->dense_w, w->dense_b, OD, o);
            if (d->dense_act == 1)
                for (int i = 0; i < OD; ++i) o[i] = tanhf(o[i]);
        } else {
            memcpy(o, pooled + (size_t)b * H, (size_t)H * sizeof(float));
        }
        if (d->normalize) { /* worker.rs:95-103 */
            float ss = 0.0f;
            for (int i = 0; i < OD; ++i) ss += o[i] * o[i];
            float nrm = sqrtf(ss);
            if (nrm < 1e-12f) nrm = 1e-12f;
            for (int i = 0; i < OD; ++i) o[i] = o[i] / nrm;
        }
    }
    free(h); free(q); free(k); free(v); free(ctx); free(tmp); free(ff); free(sc); free(pooled);
}

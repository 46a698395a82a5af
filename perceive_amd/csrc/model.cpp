// model.cpp — Model C ABI: weights, workspace and the forward pass orchestration
// (replaces model.rs:56-191 + model/worker.rs:78-106 of the reference; kernels: encoder_kernels.hip).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "encoder.h"
#include "model_internal.h"

using namespace pcv;

namespace {

constexpr int64_t kDebugTokenLimit = 16384;
constexpr int kGraphTokens = 128;  // forwards up to this many tokens are replayed as hipGraphs (256 x 256 tokens replayed: 12.90 against 12.53 ms)

void drop_graphs(pcv_model* m) {
    for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear();
    m->shape_seen.clear();
}

Tensor alloc_tensor(pcv_model* m, int64_t n) {
    Tensor t;
    t.n = n;
    PCV_HIP(hipMalloc((void**)&t.p, (size_t)n * sizeof(float)));
    m->owned.push_back(t.p);
    return t;
}

void reg(pcv_model* m, const std::string& name, float* p, int64_t n) { m->table[name] = Tensor{p, n}; }

void alloc_planes(pcv_model* m, Planes& pl, int64_t n) {
    for (int i = 0; i < (m->d.compute == PCV_COMPUTE_F16X2 ? 2 : 3); ++i) {
        PCV_HIP(hipMalloc((void**)&pl.p[i], (size_t)n * sizeof(uint16_t)));
        m->owned_planes.push_back(pl.p[i]);
    }
}

void refresh_planes(pcv_model* m) {
    if (m->d.compute == PCV_COMPUTE_F32 || !m->planes_dirty) return;
    hipStream_t st = m->ctx->stream;
    if (m->d.compute == PCV_COMPUTE_F16X2) {
        int* d_over = nullptr;
        PCV_HIP(hipMalloc((void**)&d_over, sizeof(int)));
        int over = 0;
        try {
            PCV_HIP(hipMemsetAsync(d_over, 0, sizeof(int), st));
            if (m->map_w.p) launch_split_planes_f16(st, m->map_w.p, m->map_w.n, m->map_p.p[0], m->map_p.p[1], d_over);
            for (Layer& L : m->layers) {
                launch_split_planes_f16(st, L.qkv_w.p, L.qkv_w.n, L.qkv_p.p[0], L.qkv_p.p[1], d_over);
                launch_split_planes_f16(st, L.ao_w.p, L.ao_w.n, L.ao_p.p[0], L.ao_p.p[1], d_over);
                launch_split_planes_f16(st, L.i_w.p, L.i_w.n, L.i_p.p[0], L.i_p.p[1], d_over);
                launch_split_planes_f16(st, L.f_w.p, L.f_w.n, L.f_p.p[0], L.f_p.p[1], d_over);
            }
            PCV_HIP(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, st));
            PCV_HIP(hipStreamSynchronize(st));
        } catch (...) {
            (void)hipFree(d_over);
            throw;
        }
        (void)hipFree(d_over);
        if (over) PCV_FAIL(PCV_ERR_UNSUPPORTED, "PCV_COMPUTE_F16X2: a linear-layer weight is outside (-255, 255) or not finite; use PCV_COMPUTE_F32 or PCV_COMPUTE_BF16X3");
        m->planes_dirty = false;
        return;
    }
    if (m->map_w.p) launch_split_planes(st, m->map_w.p, m->map_w.n, m->map_p.p[0], m->map_p.p[1], m->map_p.p[2]);
    for (Layer& L : m->layers) {
        launch_split_planes(st, L.qkv_w.p, L.qkv_w.n, L.qkv_p.p[0], L.qkv_p.p[1], L.qkv_p.p[2]);
        launch_split_planes(st, L.ao_w.p, L.ao_w.n, L.ao_p.p[0], L.ao_p.p[1], L.ao_p.p[2]);
        launch_split_planes(st, L.i_w.p, L.i_w.n, L.i_p.p[0], L.i_p.p[1], L.i_p.p[2]);
        launch_split_planes(st, L.f_w.p, L.f_w.n, L.f_p.p[0], L.f_p.p[1], L.f_p.p[2]);
    }
    m->planes_dirty = false;
}

// one Linear layer of the encoder in the model's compute mode
void gemm(pcv_model* m, const float* A, const Tensor& W, const Planes& P, const float* bias, const float* resid, float* C,
          int M, int N, int K, int epi) {
    if (m->d.compute == PCV_COMPUTE_BF16X3)
        launch_gemm_bf16x3(m->ctx->stream, A, P.p[0], P.p[1], P.p[2], bias, resid, C, M, N, K, epi);
    else if (m->d.compute == PCV_COMPUTE_F16X2)
        launch_gemm_f16x2(m->ctx->stream, A, P.p[0], P.p[1], bias, resid, C, M, N, K, epi);
    else
        launch_gemm_f32(m->ctx->stream, A, W.p, bias, resid, C, M, N, K, epi);
}

void build_tensors(pcv_model* m) {
    const pcv_model_desc& d = m->d;
    const int64_t H = d.hidden, F = d.intermediate;
    const int64_t E = d.embedding_size > 0 ? d.embedding_size : H;  // ALBERT factorises the embedding tables
    m->word = alloc_tensor(m, (int64_t)d.vocab_size * E);
    m->pos = alloc_tensor(m, (int64_t)d.max_positions * E);
    m->type = alloc_tensor(m, (int64_t)d.type_vocab * E);
    m->eln_w = alloc_tensor(m, E);
    m->eln_b = alloc_tensor(m, E);
    reg(m, "embeddings.word_embeddings.weight", m->word.p, m->word.n);
    reg(m, "embeddings.position_embeddings.weight", m->pos.p, m->pos.n);
    reg(m, "embeddings.token_type_embeddings.weight", m->type.p, m->type.n);
    reg(m, "embeddings.LayerNorm.weight", m->eln_w.p, E);
    reg(m, "embeddings.LayerNorm.bias", m->eln_b.p, E);
    if (d.embedding_size > 0) {
        m->map_w = alloc_tensor(m, H * E);
        m->map_b = alloc_tensor(m, H);
        if (d.compute != PCV_COMPUTE_F32) alloc_planes(m, m->map_p, H * E);
        reg(m, "encoder.embedding_hidden_mapping_in.weight", m->map_w.p, H * E);
        reg(m, "encoder.embedding_hidden_mapping_in.bias", m->map_b.p, H);
    }
    const int n_weight_sets = d.shared_layers ? 1 : d.layers;  // ALBERT: one set of layer weights, run `layers` times
    m->layers.resize(n_weight_sets);
    for (int i = 0; i < n_weight_sets; ++i) {
        Layer& L = m->layers[i];
        L.qkv_w = alloc_tensor(m, 3 * H * H);
        L.qkv_b = alloc_tensor(m, 3 * H);
        L.ao_w = alloc_tensor(m, H * H);
        L.ao_b = alloc_tensor(m, H);
        L.ln1_w = alloc_tensor(m, H);
        L.ln1_b = alloc_tensor(m, H);
        L.i_w = alloc_tensor(m, F * H);
        L.i_b = alloc_tensor(m, F);
        L.f_w = alloc_tensor(m, H * F);
        L.f_b = alloc_tensor(m, H);
        L.ln2_w = alloc_tensor(m, H);
        L.ln2_b = alloc_tensor(m, H);
        if (d.compute != PCV_COMPUTE_F32) {
            alloc_planes(m, L.qkv_p, 3 * H * H);
            alloc_planes(m, L.ao_p, H * H);
            alloc_planes(m, L.i_p, F * H);
            alloc_planes(m, L.f_p, H * F);
        }
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        reg(m, p + "attention.self.query.weight", L.qkv_w.p, H * H);
        reg(m, p + "attention.self.key.weight", L.qkv_w.p + H * H, H * H);
        reg(m, p + "attention.self.value.weight", L.qkv_w.p + 2 * H * H, H * H);
        reg(m, p + "attention.self.query.bias", L.qkv_b.p, H);
        reg(m, p + "attention.self.key.bias", L.qkv_b.p + H, H);
        reg(m, p + "attention.self.value.bias", L.qkv_b.p + 2 * H, H);
        reg(m, p + "attention.output.dense.weight", L.ao_w.p, H * H);
        reg(m, p + "attention.output.dense.bias", L.ao_b.p, H);
        reg(m, p + "attention.output.LayerNorm.weight", L.ln1_w.p, H);
        reg(m, p + "attention.output.LayerNorm.bias", L.ln1_b.p, H);
        reg(m, p + "intermediate.dense.weight", L.i_w.p, F * H);
        reg(m, p + "intermediate.dense.bias", L.i_b.p, F);
        reg(m, p + "output.dense.weight", L.f_w.p, H * F);
        reg(m, p + "output.dense.bias", L.f_b.p, H);
        reg(m, p + "output.LayerNorm.weight", L.ln2_w.p, H);
        reg(m, p + "output.LayerNorm.bias", L.ln2_b.p, H);
    }
    if (d.dense_out > 0) {
        m->dense_w = alloc_tensor(m, (int64_t)d.dense_out * H);
        m->dense_b = alloc_tensor(m, d.dense_out);
        reg(m, "dense.linear.weight", m->dense_w.p, m->dense_w.n);
        reg(m, "dense.linear.bias", m->dense_b.p, m->dense_b.n);
    }
}

// Seeded synthetic weights (no checkpoint exists offline): BERT-like scales, non-trivial LayerNorm
// and biases so that every term of the forward matters.  Tensor order = name order of the table.
void fill_synthetic(pcv_model* m, uint64_t seed) {
    uint32_t idx = 0;
    for (auto& kv : m->table) {
        const std::string& name = kv.first;
        float scale = 0.05f, offset = 0.0f;
        if (name.find("LayerNorm.weight") != std::string::npos) {
            scale = 0.1f;
            offset = 1.0f;
        } else if (name.size() >= 4 && name.compare(name.size() - 4, 4, "bias") == 0) {
            scale = 0.05f;
        } else if (name.find("embeddings") != std::string::npos) {
            scale = 0.5f;
        }
        launch_synth_weights(m->ctx->stream, kv.second.p, kv.second.n, seed, idx++, scale, offset);
    }
    PCV_HIP(hipStreamSynchronize(m->ctx->stream));
    PCV_HIP(hipGetLastError());
}

// flat weight file: "PCVW0001", u32 count, then { u32 name_len, name, u64 numel, f32[numel] } (LE)
void load_weight_file(pcv_model* m, const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) PCV_FAIL(PCV_ERR_IO, "cannot open weight file %s", path);
    auto fail = [&](const char* what) {
        std::fclose(f);
        PCV_FAIL(PCV_ERR_IO, "weight file %s: %s", path, what);
    };
    char magic[8];
    uint32_t count = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "PCVW0001", 8) != 0) fail("bad magic");
    if (std::fread(&count, 4, 1, f) != 1) fail("truncated header");
    std::vector<float> buf;
    size_t seen = 0;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t nl = 0;
        uint64_t numel = 0;
        if (std::fread(&nl, 4, 1, f) != 1 || nl > 4096) fail("bad tensor name length");
        std::string name(nl, '\0');
        if (std::fread(&name[0], 1, nl, f) != nl || std::fread(&numel, 8, 1, f) != 1) fail("truncated entry");
        auto it = m->table.find(name);
        if (it == m->table.end()) {  // tensors this architecture does not use (e.g. pooler, position_ids)
            if (std::fseek(f, (long)(numel * 4), SEEK_CUR) != 0) fail("truncated data");
            continue;
        }
        if ((int64_t)numel != it->second.n) {
            std::fclose(f);
            PCV_FAIL(PCV_ERR_IO, "weight file %s: tensor %s has %llu elements, model expects %lld", path, name.c_str(),
                     (unsigned long long)numel, (long long)it->second.n);
        }
        buf.resize(numel);
        if (std::fread(buf.data(), 4, numel, f) != numel) fail("truncated data");
        PCV_HIP(hipMemcpy(it->second.p, buf.data(), numel * 4, hipMemcpyHostToDevice));
        ++seen;
    }
    std::fclose(f);
    if (seen != m->table.size())
        PCV_FAIL(PCV_ERR_IO, "weight file %s provides %zu of the %zu tensors the model needs", path, seen,
                 m->table.size());
}

void free_workspace(pcv_model* m) {
    for (void* p : {(void*)m->d_ids, (void*)m->d_mask, (void*)m->hidden, (void*)m->qkv, (void*)m->ctxbuf, (void*)m->tmp,
                    (void*)m->ff, (void*)m->mask_add, (void*)m->mask01, (void*)m->pooled, (void*)m->out, (void*)m->dbg})
        if (p) (void)hipFree(p);
    m->d_ids = m->d_mask = nullptr;
    m->hidden = m->qkv = m->ctxbuf = m->tmp = m->ff = m->mask_add = m->mask01 = m->pooled = m->out = m->dbg = nullptr;
    m->cap_tokens = m->cap_padded = m->cap_batch = 0;
}

void ensure_workspace(pcv_model* m, int B, int L) {
    int64_t T = (int64_t)B * L;
    int64_t Tp = (int64_t)B * ((L + 31) / 32 * 32);
    if (T <= m->cap_tokens && Tp <= m->cap_padded && B <= m->cap_batch) return;
    // every buffer is sized from the capacities that are recorded, and a capacity never shrinks: a later
    // call with more tokens at the same padded count (B=4,L=33 then B=4,L=60) must not reuse short buffers
    T = std::max(T, m->cap_tokens);
    Tp = std::max(Tp, m->cap_padded);
    B = (int)std::max<int64_t>(B, m->cap_batch);
    drop_graphs(m);  // they hold the old workspace pointers
    PCV_HIP(hipStreamSynchronize(m->ctx->stream));
    free_workspace(m);
    const int64_t H = m->d.hidden, F = m->d.intermediate;
    const int64_t OD = m->d.dense_out > 0 ? m->d.dense_out : H;
    PCV_HIP(hipMalloc((void**)&m->d_ids, T * 8));
    PCV_HIP(hipMalloc((void**)&m->d_mask, T * 8));
    PCV_HIP(hipMalloc((void**)&m->hidden, T * H * 4));
    PCV_HIP(hipMalloc((void**)&m->qkv, T * 3 * H * 4));
    PCV_HIP(hipMalloc((void**)&m->ctxbuf, T * H * 4));
    PCV_HIP(hipMalloc((void**)&m->tmp, T * H * 4));
    PCV_HIP(hipMalloc((void**)&m->ff, T * F * 4));
    PCV_HIP(hipMalloc((void**)&m->mask_add, Tp * 4));
    PCV_HIP(hipMalloc((void**)&m->mask01, T * 4));
    PCV_HIP(hipMalloc((void**)&m->pooled, (int64_t)B * H * 4));
    PCV_HIP(hipMalloc((void**)&m->out, (int64_t)B * OD * 4));
    if (T <= kDebugTokenLimit) PCV_HIP(hipMalloc((void**)&m->dbg, (int64_t)(m->d.layers + 1) * T * H * 4));  // T = capacity: any smaller batch fits
    m->cap_tokens = T;
    m->cap_padded = Tp;
    m->cap_batch = B;
}

// worker.rs:78-106 on the device.  Leaves [B][out_dim] in m->out.
// The kernel sequence of one forward (no allocation, no synchronisation: capturable in a hipGraph).
void launch_forward(pcv_model* m, int B, int L) {
    const pcv_model_desc& d = m->d;
    const int H = d.hidden, F = d.intermediate;
    const int T = B * L;
    hipStream_t st = m->ctx->stream;
    if (d.embedding_size > 0) {  // embeddings + LayerNorm at the narrow width, then the Linear up to hidden
        launch_embed_ln(st, m->d_ids, m->d_mask, B, L, d.embedding_size, d.vocab_size, m->word.p, m->pos.p, m->type.p, m->eln_w.p,
                        m->eln_b.p, d.layer_norm_eps, m->tmp, m->mask_add, m->mask01);
        gemm(m, m->tmp, m->map_w, m->map_p, m->map_b.p, nullptr, m->hidden, T, H, d.embedding_size, EPI_BIAS);
    } else {
        launch_embed_ln(st, m->d_ids, m->d_mask, B, L, H, d.vocab_size, m->word.p, m->pos.p, m->type.p, m->eln_w.p,
                        m->eln_b.p, d.layer_norm_eps, m->hidden, m->mask_add, m->mask01);
    }
    const int act_epi = d.hidden_act == PCV_GELU_TANH ? EPI_BIAS_GELU_TANH : EPI_BIAS_GELU;
    const bool dbg = m->dbg != nullptr && T <= kDebugTokenLimit;
    auto snap = [&](int layer) {
        if (dbg)
            PCV_HIP(hipMemcpyAsync(m->dbg + (size_t)layer * T * H, m->hidden, (size_t)T * H * 4,
                                   hipMemcpyDeviceToDevice, st));
    };
    snap(0);
    for (int ly = 0; ly < d.layers; ++ly) {
        const Layer& W = m->layers[d.shared_layers ? 0 : ly];
        gemm(m, m->hidden, W.qkv_w, W.qkv_p, W.qkv_b.p, nullptr, m->qkv, T, 3 * H, H, EPI_BIAS);
        if (!(d.compute == PCV_COMPUTE_F16X2 && launch_attention_f16(st, m->qkv, m->mask_add, m->ctxbuf, B, L, H, d.heads)))
            launch_attention(st, m->qkv, m->mask_add, m->ctxbuf, B, L, H, d.heads);
        // projection + residual + LayerNorm: one kernel where the width allows whole rows per workgroup
        const bool fuse = d.compute == PCV_COMPUTE_F32 && m->fuse_ln;
        if (!(fuse && launch_gemm_f32_ln(st, m->ctxbuf, W.ao_w.p, W.ao_b.p, m->hidden, W.ln1_w.p, W.ln1_b.p, d.layer_norm_eps,
                                         m->tmp, T, H, H))) {
            gemm(m, m->ctxbuf, W.ao_w, W.ao_p, W.ao_b.p, m->hidden, m->tmp, T, H, H, EPI_BIAS_RESIDUAL);
            launch_layer_norm(st, m->tmp, T, H, W.ln1_w.p, W.ln1_b.p, d.layer_norm_eps);
        }
        gemm(m, m->tmp, W.i_w, W.i_p, W.i_b.p, nullptr, m->ff, T, F, H, act_epi);
        if (!(fuse && launch_gemm_f32_ln(st, m->ff, W.f_w.p, W.f_b.p, m->tmp, W.ln2_w.p, W.ln2_b.p, d.layer_norm_eps, m->hidden,
                                         T, H, F))) {
            gemm(m, m->ff, W.f_w, W.f_p, W.f_b.p, m->tmp, m->hidden, T, H, F, EPI_BIAS_RESIDUAL);
            launch_layer_norm(st, m->hidden, T, H, W.ln2_w.p, W.ln2_b.p, d.layer_norm_eps);
        }
        snap(ly + 1);
    }
    if (d.dense_out > 0) {
        launch_pool(st, m->hidden, m->mask01, B, L, H, d.pooling, 0, m->pooled);
        launch_dense(st, m->pooled, m->dense_w.p, m->dense_b.p, B, H, d.dense_out, d.dense_activation, d.normalize,
                     m->out);
    } else {
        launch_pool(st, m->hidden, m->mask01, B, L, H, d.pooling, d.normalize, m->out);
    }
    m->dbg_tokens = dbg ? T : 0;
}

// worker.rs:78-106 on the device.  Leaves [B][out_dim] in m->out.
// Up to kGraphTokens tokens (a query, a few highlight chunks) the forward is ~45 kernels of a few
// microseconds each: launch-bound.  The second time a (B, L) shape is seen its launch sequence is
// captured into a hipGraph and replayed from then on (workspace and weight pointers are stable; a
// workspace reallocation drops the cached graphs).
void forward(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L) {
    const int T = B * L;
    hipStream_t st = m->ctx->stream;
    ensure_workspace(m, B, L);
    refresh_planes(m);
    // Tensor::stack(ids/masks).to(device), worker.rs:82-83
    PCV_HIP(hipMemcpyAsync(m->d_ids, ids, (size_t)T * 8, hipMemcpyHostToDevice, st));
    PCV_HIP(hipMemcpyAsync(m->d_mask, mask, (size_t)T * 8, hipMemcpyHostToDevice, st));
    PCV_HIP(hipEventRecord(m->ev0, st));
    const std::pair<int, int> key{B, L};
    auto it = m->graphs.find(key);
    if (T <= kGraphTokens && m->use_graphs && it != m->graphs.end()) {
        PCV_HIP(hipGraphLaunch(it->second, st));
        m->dbg_tokens = (m->dbg != nullptr && T <= kDebugTokenLimit) ? T : 0;
    } else if (T <= kGraphTokens && m->use_graphs && ++m->shape_seen[key] >= 2) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        PCV_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        bool ok = true;
        try {
            launch_forward(m, B, L);
        } catch (...) {
            ok = false;
        }
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (ok && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            m->graphs[key] = exec;
            PCV_HIP(hipGraphLaunch(exec, st));
        } else {  // capture not possible here: stay eager for good
            (void)hipGetLastError();
            m->use_graphs = false;
            launch_forward(m, B, L);
        }
        if (graph) (void)hipGraphDestroy(graph);
    } else {
        launch_forward(m, B, L);
    }
    PCV_HIP(hipEventRecord(m->ev1, st));
    m->last_B = B;
    m->last_L = L;
}


void check_tokens(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L) {
    PCV_REQUIRE(m != nullptr && ids != nullptr && mask != nullptr, "encode_tokens: NULL argument");
    PCV_REQUIRE(B > 0 && L > 0, "encode_tokens: empty batch (B=%d, L=%d)", B, L);
    PCV_REQUIRE(L <= m->d.max_positions, "encode_tokens: sequence length %d exceeds max_position_embeddings %d", L,
                m->d.max_positions);
    PCV_REQUIRE((int64_t)B * L < ((int64_t)1 << 31), "encode_tokens: batch of %d x %d tokens is too large", B, L);
    // an id outside the embedding table is a tokenizer / checkpoint mismatch: the reference's embedding lookup
    // fails on it (worker.rs:85-86 -> libtorch index error); it must not be clamped into a plausible vector
    const int64_t V = m->d.vocab_size;
    for (int64_t i = 0, n = (int64_t)B * L; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= V)
            PCV_FAIL(PCV_ERR_INVALID, "encode_tokens: token id %lld at [%lld][%lld] is outside the vocabulary [0,%lld)",
                     (long long)ids[i], (long long)(i / L), (long long)(i % L), (long long)V);
}

void finish_stats(pcv_model* m) {
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, m->ev0, m->ev1);
    const double H = m->d.hidden, F = m->d.intermediate, L = m->last_L, T = (double)m->last_B * m->last_L;
    m->stats.total_ms = ms;
    m->stats.flops = T * m->d.layers * (8.0 * H * H + 4.0 * H * F + 4.0 * L * H);
    if (m->d.embedding_size > 0) m->stats.flops += 2.0 * T * H * m->d.embedding_size;
    m->stats.batch = m->last_B;
    m->stats.seq_len = m->last_L;
}

}  // namespace

namespace pcv {
void model_check_tokens(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L) { check_tokens(m, ids, mask, B, L); }
void model_forward(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L) { forward(m, ids, mask, B, L); }
void model_finish_stats(pcv_model* m) { finish_stats(m); }
void model_check_f16_output(pcv_model* m, const float* out, size_t n) {
    if (m->d.compute != PCV_COMPUTE_F16X2) return;
    // the one way this mode can fail silently: an activation beyond f16's (rescaled) range turns into inf
    // and the embedding into NaN.  The f32 path would have produced numbers, so say so instead.
    for (size_t e = 0; e < n; ++e)
        if (!std::isfinite(out[e]))
            PCV_FAIL(PCV_ERR_UNSUPPORTED, "PCV_COMPUTE_F16X2: non-finite embedding (activation outside the f16 range?); "
                                          "use PCV_COMPUTE_F32 or PCV_COMPUTE_BF16X3 for this model");
}
}  // namespace pcv

extern "C" {

void pcv_model_desc_minilm_l6(pcv_model_desc* d) {
    if (!d) return;
    d->vocab_size = 30522;
    d->hidden = 384;
    d->layers = 6;
    d->heads = 12;
    d->intermediate = 1536;
    d->max_positions = 512;
    d->type_vocab = 2;
    d->layer_norm_eps = 1e-12f;
    d->pooling = PCV_POOL_MEAN;
    d->normalize = 1;
    d->dense_out = 0;
    d->dense_activation = PCV_ACT_IDENTITY;
    d->max_seq_length = 256;
    d->compute = PCV_COMPUTE_F32;
    d->embedding_size = 0;
    d->shared_layers = 0;
    d->hidden_act = PCV_GELU_ERF;
}

pcv_status pcv_model_create(pcv_ctx* ctx, const pcv_model_desc* desc, const char* weights_path,
                            uint64_t synthetic_seed, pcv_model** out) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && desc != nullptr && out != nullptr, "model_create: NULL argument");
        *out = nullptr;
        const pcv_model_desc& d = *desc;
        PCV_REQUIRE(d.hidden > 0 && d.layers > 0 && d.heads > 0 && d.intermediate > 0 && d.vocab_size > 0 &&
                        d.max_positions > 0 && d.type_vocab > 0,
                    "model_create: non-positive dimension in the model description");
        if (d.hidden % 128 != 0 || d.intermediate % 128 != 0 || d.hidden > 1024)
            PCV_FAIL(PCV_ERR_UNSUPPORTED, "model_create: hidden %d / intermediate %d must be multiples of 128 (hidden <= 1024)",
                     d.hidden, d.intermediate);
        if (d.hidden % d.heads != 0 || (d.hidden / d.heads != 32 && d.hidden / d.heads != 64))
            PCV_FAIL(PCV_ERR_UNSUPPORTED, "model_create: head dimension %d not supported (32 or 64)",
                     d.heads ? d.hidden / d.heads : 0);
        if (d.embedding_size < 0 || d.embedding_size % 128 != 0 || d.embedding_size > d.hidden)
            PCV_FAIL(PCV_ERR_UNSUPPORTED, "model_create: embedding_size %d must be a multiple of 128 up to hidden (0 = hidden)", d.embedding_size);
        PCV_REQUIRE(d.hidden_act == PCV_GELU_ERF || d.hidden_act == PCV_GELU_TANH, "model_create: unknown hidden_act %d", d.hidden_act);
        PCV_REQUIRE(d.pooling >= PCV_POOL_MEAN && d.pooling <= PCV_POOL_MEAN_SQRT_LEN, "model_create: unknown pooling %d",
                    d.pooling);
        PCV_REQUIRE(d.dense_out >= 0 && d.dense_out <= 1024, "model_create: dense_out %d outside [0,1024]", d.dense_out);
        PCV_REQUIRE(d.compute == PCV_COMPUTE_F32 || d.compute == PCV_COMPUTE_BF16X3 || d.compute == PCV_COMPUTE_F16X2,
                    "model_create: unknown compute mode %d",
                    d.compute);
        PCV_HIP(hipSetDevice(ctx->device));
        auto* m = new pcv_model();
        m->ctx = ctx;
        m->d = d;
        if (getenv("PCV_NO_GRAPHS")) m->use_graphs = false;  // diagnostics: always launch eagerly
        if (getenv("PCV_NO_FUSED_LN")) m->fuse_ln = false;   // diagnostics / A-B: projection and LayerNorm as two kernels
        try {
            build_tensors(m);
            PCV_HIP(hipEventCreate(&m->ev0));
            PCV_HIP(hipEventCreate(&m->ev1));
            if (weights_path && weights_path[0])
                load_weight_file(m, weights_path);
            else
                fill_synthetic(m, synthetic_seed);
        } catch (...) {
            pcv_model_destroy(m);
            throw;
        }
        *out = m;
    });
}

pcv_status pcv_model_destroy(pcv_model* m) {
    return guarded([&] {
        if (!m) return;
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->ctx->stream);
        drop_graphs(m);
        free_workspace(m);
        for (float* p : m->owned) (void)hipFree(p);
        for (void* p : m->owned_planes) (void)hipFree(p);
        if (m->ev0) (void)hipEventDestroy(m->ev0);
        if (m->ev1) (void)hipEventDestroy(m->ev1);
        for (void* p : {(void*)m->hl_emb, (void*)m->hl_query, (void*)m->hl_bounds, (void*)m->hl_best})
            if (p) (void)hipFree(p);
        if (m->tok && m->own_tok) pcv_tokenizer_destroy(m->tok);
        delete m;
    });
}

pcv_status pcv_model_output_dim(pcv_model* m, int* out_dim) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && out_dim != nullptr, "model_output_dim: NULL argument");
        *out_dim = m->d.dense_out > 0 ? m->d.dense_out : m->d.hidden;
    });
}

pcv_status pcv_model_set_tensor(pcv_model* m, const char* name, const float* data, int64_t n) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && name != nullptr && data != nullptr, "model_set_tensor: NULL argument");
        std::lock_guard<std::mutex> lk(m->mu);
        auto it = m->table.find(name);
        PCV_REQUIRE(it != m->table.end(), "model_set_tensor: unknown tensor '%s'", name);
        PCV_REQUIRE(it->second.n == n, "model_set_tensor: '%s' has %lld elements, got %lld", name,
                    (long long)it->second.n, (long long)n);
        PCV_HIP(hipSetDevice(m->ctx->device));
        PCV_HIP(hipStreamSynchronize(m->ctx->stream));
        PCV_HIP(hipMemcpy(it->second.p, data, (size_t)n * 4, hipMemcpyHostToDevice));
        m->planes_dirty = true;
    });
}

pcv_status pcv_model_get_tensor(pcv_model* m, const char* name, float* out, int64_t cap, int64_t* out_n) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && name != nullptr, "model_get_tensor: NULL argument");
        std::lock_guard<std::mutex> lk(m->mu);
        auto it = m->table.find(name);
        PCV_REQUIRE(it != m->table.end(), "model_get_tensor: unknown tensor '%s'", name);
        if (out_n) *out_n = it->second.n;
        if (!out) return;  // size query
        PCV_REQUIRE(cap >= it->second.n, "model_get_tensor: '%s' needs room for %lld values", name,
                    (long long)it->second.n);
        PCV_HIP(hipSetDevice(m->ctx->device));
        PCV_HIP(hipStreamSynchronize(m->ctx->stream));
        PCV_HIP(hipMemcpy(out, it->second.p, (size_t)it->second.n * 4, hipMemcpyDeviceToHost));
    });
}

pcv_status pcv_model_encode_tokens(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L, float* out) {
    return guarded([&] {
        check_tokens(m, ids, mask, B, L);
        PCV_REQUIRE(out != nullptr, "encode_tokens: out is NULL");
        std::lock_guard<std::mutex> lk(m->mu);  // one forward at a time, like the worker channel (model.rs:161,187)
        PCV_HIP(hipSetDevice(m->ctx->device));
        forward(m, ids, mask, B, L);
        const int OD = m->d.dense_out > 0 ? m->d.dense_out : m->d.hidden;
        PCV_HIP(hipMemcpyAsync(out, m->out, (size_t)B * OD * 4, hipMemcpyDeviceToHost, m->ctx->stream));
        PCV_HIP(hipStreamSynchronize(m->ctx->stream));
        PCV_HIP(hipGetLastError());
        finish_stats(m);
        model_check_f16_output(m, out, (size_t)B * OD);
    });
}

pcv_status pcv_model_encode_tokens_device(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L,
                                          void* d_out, int async) {
    return guarded([&] {
        check_tokens(m, ids, mask, B, L);
        PCV_REQUIRE(d_out != nullptr, "encode_tokens_device: d_out is NULL");
        std::lock_guard<std::mutex> lk(m->mu);
        PCV_HIP(hipSetDevice(m->ctx->device));
        forward(m, ids, mask, B, L);
        const int OD = m->d.dense_out > 0 ? m->d.dense_out : m->d.hidden;
        PCV_HIP(hipMemcpyAsync(d_out, m->out, (size_t)B * OD * 4, hipMemcpyDeviceToDevice, m->ctx->stream));
        if (!async) {
            PCV_HIP(hipStreamSynchronize(m->ctx->stream));
            PCV_HIP(hipGetLastError());
            finish_stats(m);
        }
    });
}

pcv_status pcv_model_debug_hidden(pcv_model* m, int layer, float* out, int64_t cap) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && out != nullptr, "model_debug_hidden: NULL argument");
        std::lock_guard<std::mutex> lk(m->mu);
        PCV_REQUIRE(m->dbg_tokens > 0, "model_debug_hidden: no hidden states kept (batch larger than %lld tokens?)",
                    (long long)kDebugTokenLimit);
        PCV_REQUIRE(layer >= 0 && layer <= m->d.layers, "model_debug_hidden: layer %d outside [0,%d]", layer, m->d.layers);
        const int64_t n = m->dbg_tokens * m->d.hidden;
        PCV_REQUIRE(cap >= n, "model_debug_hidden: need room for %lld values", (long long)n);
        PCV_HIP(hipSetDevice(m->ctx->device));
        PCV_HIP(hipStreamSynchronize(m->ctx->stream));
        PCV_HIP(hipMemcpy(out, m->dbg + (size_t)layer * n, (size_t)n * 4, hipMemcpyDeviceToHost));
    });
}

pcv_status pcv_model_last_stats(pcv_model* m, pcv_encode_stats* out) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && out != nullptr, "model_last_stats: NULL argument");
        std::lock_guard<std::mutex> lk(m->mu);
        *out = m->stats;
    });
}

}  // extern "C"

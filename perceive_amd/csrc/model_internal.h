// model_internal.h — the Model handle as model.cpp (device forward) and text_model.cpp (text in: tokenizer,
// model directories, highlight) share it.  Not part of the C ABI.
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"
#include "encoder.h"

struct pcv_tokenizer;

namespace pcv {

struct Tensor {
    float* p = nullptr;
    int64_t n = 0;
};

struct Planes {  // hi / mid / lo bf16 terms (PCV_COMPUTE_BF16X3) or hi / lo f16 terms of 2^8 * W (PCV_COMPUTE_F16X2)
    uint16_t* p[3] = {nullptr, nullptr, nullptr};
};

struct Layer {
    Tensor qkv_w, qkv_b;  // fused [3H][H], [3H]: rows 0..H = query, H..2H = key, 2H..3H = value
    Tensor ao_w, ao_b, ln1_w, ln1_b;
    Tensor i_w, i_b, f_w, f_b, ln2_w, ln2_b;
    Planes qkv_p, ao_p, i_p, f_p;
};

}  // namespace pcv

struct pcv_model {
    pcv_ctx* ctx = nullptr;
    pcv_model_desc d{};
    pcv::Tensor word, pos, type, eln_w, eln_b, dense_w, dense_b;
    pcv::Tensor map_w, map_b;  // ALBERT: embedding_hidden_mapping_in [hidden][embedding_size], [hidden]
    pcv::Planes map_p;
    std::vector<pcv::Layer> layers;
    // name -> (device pointer, element count): HF / rust-bert tensor names
    std::map<std::string, pcv::Tensor> table;
    std::vector<float*> owned;
    std::vector<void*> owned_planes;
    bool planes_dirty = true;  // weights changed since the bf16 planes were derived
    std::mutex mu;
    pcv_encode_stats stats{};

    // workspace, grown on demand: capacities in tokens (B*L), padded tokens (B*roundup32(L)) and batch rows
    int64_t cap_tokens = 0, cap_padded = 0, cap_batch = 0;
    int64_t* d_ids = nullptr;
    int64_t* d_mask = nullptr;
    float *hidden = nullptr, *qkv = nullptr, *ctxbuf = nullptr, *tmp = nullptr, *ff = nullptr;
    float *mask_add = nullptr, *mask01 = nullptr, *pooled = nullptr, *out = nullptr;
    float* dbg = nullptr;  // [(layers+1)][T][H] of the last encode when it is small
    int64_t dbg_tokens = 0;
    int last_B = 0, last_L = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // hipGraph replay of small forwards, keyed by (B, L)
    std::map<std::pair<int, int>, hipGraphExec_t> graphs;
    std::map<std::pair<int, int>, int> shape_seen;
    bool use_graphs = true;
    bool fuse_ln = true;  // residual + LayerNorm inside the projection GEMMs where the width allows (PCV_NO_FUSED_LN=1: off)
    // text side (text_model.cpp): Model::tokenizer / sentence_bert_config of model.rs:61-63
    pcv_tokenizer* tok = nullptr;
    bool own_tok = false;
    int64_t pad_id = 0;                       // get_pad_id().unwrap_or(0), tokenize.rs:19
    int arch = 0;                             // 0 bert, 1 distilbert, 2 roberta, 3 albert: how checkpoint tensor names map
    int pos_shift = 0;                        // rows of the position table skipped (RoBERTa: padding_idx + 1)
    std::map<std::string, bool> loaded;       // tensors provided so far by pcv_model_load_hf_tensor
    // highlight scratch (grown on demand)
    float* hl_emb = nullptr;                  // [chunks][out_dim] chunk embeddings
    size_t hl_emb_cap = 0;
    float* hl_query = nullptr;                // [out_dim]
    int32_t* hl_bounds = nullptr;             // [docs + 1] chunk ranges per document
    int32_t* hl_best = nullptr;               // [docs] winning chunk (+ [docs] NaN flags behind it)
    size_t hl_docs_cap = 0;
};


namespace pcv {
// model.cpp internals used by text_model.cpp; the caller holds m->mu
void model_check_tokens(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L);
void model_forward(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L);  // leaves [B][out_dim] in m->out
void model_finish_stats(pcv_model* m);
void model_check_f16_output(pcv_model* m, const float* out, size_t n);
}  // namespace pcv

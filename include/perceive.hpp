// perceive.hpp — C++ host-side mirror of perceive-core's public surface over the C ABI
// (include/perceive_hip.h).  Header-only.  The reference is a compiled (Rust) library; no Rust
// toolchain exists in the build image, so this is the compiled-language form of the shim in
// shim/perceive-core: same names, argument meaning and error behaviour as
//   crates/perceive-core/search.rs   (Searcher, SearchItem, serialize/deserialize_embedding)
//   crates/perceive-core/model.rs    (Model, ModelError, SentenceEmbeddingsModelType)
//   crates/perceive-core/lib.rs:63-77 (dot_product, cosine_similarity_*)
// Errors: the reference returns Result<_, ModelError/DbError/eyre::Report>; here a failing status
// throws perceive::Error carrying pcv_last_error().
#pragma once
#include <cstdint>
#include <array>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_set>
#include <utility>
#include <vector>

#include "perceive_hip.h"

namespace perceive {

struct Error : std::runtime_error {
    pcv_status status;
    Error(pcv_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};
struct ModelError : Error {  // model.rs:29-42
    using Error::Error;
};

inline void check(pcv_status s) {
    if (s != PCV_OK) throw Error(s, pcv_last_error());
}

// one process drives one GPU; replaces tch::Device::cuda_if_available() (model.rs:117)
class Context {
public:
    explicit Context(int device_index = 0) { check(pcv_init(device_index, &h_)); }
    ~Context() { pcv_shutdown(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    pcv_ctx* handle() const { return h_; }
    void synchronize() { check(pcv_synchronize(h_)); }

private:
    pcv_ctx* h_ = nullptr;
};

// search.rs:18-22
struct SearchItem {
    int64_t id;
    float score;
};

// search.rs:281-294
inline std::vector<float> deserialize_embedding(const std::vector<uint8_t>& value) {
    std::vector<float> out(value.size() / 4);
    size_t n = 0;
    check(pcv_deserialize_embedding(value.data(), value.size(), out.data(), out.size(), &n));
    out.resize(n);
    return out;
}
inline std::vector<uint8_t> serialize_embedding(const std::vector<float>& embedding) {
    std::vector<uint8_t> out(embedding.size() * 4);
    check(pcv_serialize_embedding(embedding.data(), embedding.size(), out.data(), out.size()));
    return out;
}

// One row of the query Searcher::build runs (search.rs:87-93): (items.id, source_id, embedding BLOB)
struct EmbeddingRow {
    int64_t item_id;
    int64_t source_id;
    std::vector<uint8_t> embedding;
};

enum class Metric { Cosine = PCV_METRIC_COSINE, Dot = PCV_METRIC_DOT };

// A persistent RCCL communicator for the row-sharded Searcher (one process per GPU).  Rank 0 makes the
// id and the host ships its 128 bytes to every rank by its own means; no reference counterpart (the
// reference is one process), it is the multi-GPU form of search.rs:163-181.
class Comm {
public:
    static std::array<uint8_t, 128> unique_id() {
        std::array<uint8_t, 128> id{};
        check(pcv_comm_unique_id(id.data()));
        return id;
    }
    Comm(Context& ctx, int world_size, int rank, const std::array<uint8_t, 128>& id) {
        check(pcv_comm_create(ctx.handle(), world_size, rank, id.data(), &h_));
    }
    ~Comm() { pcv_comm_destroy(h_); }  // before its Context
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    pcv_comm* handle() const { return h_; }

private:
    pcv_comm* h_ = nullptr;
};

// search.rs:29-260.  Metric::Dot reproduces the reference Searcher's scores exactly
// (max(0, 1 - dot/len), ascending); Metric::Cosine is lib.rs:67-77.
class Searcher {
public:
    std::unordered_set<int64_t> hidden;  // search.rs:31-34 (kept; search_vector does not consult it)

    Searcher(Context& ctx, int dim, Metric metric = Metric::Dot) : dim_(dim) {
        check(pcv_searcher_create(ctx.handle(), dim, (int)metric, &h_));
    }
    ~Searcher() { pcv_searcher_destroy(h_); }
    Searcher(const Searcher&) = delete;
    Searcher& operator=(const Searcher&) = delete;

    // Searcher::build (search.rs:38-56) with the Database replaced by its row stream
    template <class Rows>
    static std::unique_ptr<Searcher> build(Context& ctx, const Rows& rows, int dim, Metric metric = Metric::Dot) {
        auto s = std::make_unique<Searcher>(ctx, dim, metric);
        for (const EmbeddingRow& r : rows) s->insert(r);
        check(pcv_searcher_finalize(s->h_));
        return s;
    }
    // Searcher::rebuild_source (search.rs:58-79)
    template <class Rows>
    void rebuild_source(const Rows& rows, int64_t source_id) {
        // the new index is built first and swapped in only once it exists (search.rs:57-79): staged under
        // PCV_STAGING_SOURCE, so a bad row leaves the source's old rows in place
        check(pcv_searcher_clear_source(h_, PCV_STAGING_SOURCE));
        try {
            for (const EmbeddingRow& r : rows)
                if (r.source_id == source_id) insert(r, PCV_STAGING_SOURCE);  // search.rs:106-109
            check(pcv_searcher_finalize(h_));
        } catch (...) {
            pcv_searcher_clear_source(h_, PCV_STAGING_SOURCE);
            pcv_searcher_finalize(h_);
            throw;
        }
        check(pcv_searcher_replace_source(h_, PCV_STAGING_SOURCE, source_id));
        check(pcv_searcher_finalize(h_));
    }
    // Searcher::search_vector (search.rs:157-182)
    std::vector<SearchItem> search_vector(const std::vector<int64_t>& sources, size_t num_results,
                                          const std::vector<float>& vector) const {
        if (sources.empty()) return {};  // `sources.contains(..)` matches nothing
        std::vector<int64_t> ids(num_results);
        std::vector<float> scores(num_results);
        int count = 0;
        check(pcv_searcher_search(h_, vector.data(), 1, sources.data(), (int)sources.size(), (int)num_results,
                                  ids.data(), scores.data(), &count));
        std::vector<SearchItem> out;
        for (int i = 0; i < count; ++i) out.push_back({ids[i], scores[i]});
        return out;
    }
    // search_vector over every rank's shard: a collective, same arguments on all ranks; this rank's rows
    // start at global position `set_shard_offset`
    void set_shard_offset(int64_t first_global_pos) { check(pcv_searcher_set_shard_offset(h_, first_global_pos)); }
    std::vector<SearchItem> search_vector_sharded(Comm& comm, const std::vector<int64_t>& sources, size_t num_results,
                                                  const std::vector<float>& vector) const {
        if (sources.empty()) return {};
        std::vector<int64_t> ids(num_results);
        std::vector<float> scores(num_results);
        int count = 0;
        check(pcv_searcher_search_sharded(h_, comm.handle(), vector.data(), 1, sources.data(), (int)sources.size(),
                                          (int)num_results, ids.data(), scores.data(), &count));
        std::vector<SearchItem> out;
        for (int i = 0; i < count; ++i) out.push_back({ids[i], scores[i]});
        return out;
    }
    int64_t num_rows() const {
        int64_t n = 0;
        check(pcv_searcher_num_rows(h_, &n));
        return n;
    }
    // Searcher::build / rebuild_source against the reference's SQLite file itself (search.rs:38-155): the SQL
    // runs inside the library.  `only_source` empty = every source of the database.
    int64_t load_sqlite(const std::string& db_path, uint32_t model_id, uint32_t model_version,
                        std::optional<int64_t> only_source = std::nullopt) {
        int64_t rows = 0;
        const int64_t src = only_source.value_or(0);
        check(pcv_searcher_load_sqlite(h_, db_path.c_str(), model_id, model_version, only_source ? &src : nullptr, &rows));
        return rows;
    }
    // capacity hint: the rows about to be added to `source_id` land in one device segment
    void reserve(int64_t source_id, int64_t n_rows) { check(pcv_searcher_reserve(h_, source_id, n_rows)); }
    // narrow screening copy of the rows next to the f32 rows (a half / a quarter of the bytes per scan, same results):
    // PCV_SCREEN_COPY_{OFF, BF16, INT8, AUTO}; built by the next rebuild / finalize
    void set_screening_copy(int mode) { check(pcv_searcher_set_screening_copy(h_, mode)); }
    pcv_scan_stats last_stats() const {
        pcv_scan_stats st;
        check(pcv_searcher_last_stats(h_, &st));
        return st;
    }
    pcv_searcher* handle() const { return h_; }

private:
    void insert(const EmbeddingRow& r) { insert(r, r.source_id); }
    void insert(const EmbeddingRow& r, int64_t into_source) {
        if ((int)r.embedding.size() != dim_ * 4)
            throw Error(PCV_ERR_INVALID, "embedding blob of item " + std::to_string(r.item_id) + " has the wrong size");
        check(pcv_searcher_add_blobs(h_, into_source, &r.item_id, r.embedding.data(), 1));
    }
    pcv_searcher* h_ = nullptr;
    int dim_;
};

// configs.rs:30-39, model_id() of configs.rs:72-83
enum class SentenceEmbeddingsModelType {
    AllMiniLmL6V2 = 0,
    AllMiniLmL12V2 = 1,
    DistiluseBaseMultilingualCased = 2,
    AllDistilrobertaV1 = 3,
    ParaphraseAlbertSmallV2 = 4,
    MsMarcoDistilbertDotV5 = 5,
    MsMarcoDistilbertBaseTasB = 6,
    MsMarcoBertBaseDotV5 = 7,
};
inline uint32_t model_id(SentenceEmbeddingsModelType t) { return (uint32_t)t; }

// One tensor of a checkpoint file (model.safetensors, rust_model.ot, pytorch_model.bin) as pcv_checkpoint_visit reports it
struct CheckpointTensor {
    std::string name;
    std::vector<int64_t> shape;
    int dtype = PCV_TENSOR_OTHER;
    std::vector<float> values;  // empty for integer tensors
};
inline std::vector<CheckpointTensor> read_checkpoint(const std::string& path) {
    std::vector<CheckpointTensor> out;
    auto visit = [](void* user, const char* name, const int64_t* shape, int rank, int dtype, const float* values, int64_t numel) -> int {
        auto* v = static_cast<std::vector<CheckpointTensor>*>(user);
        CheckpointTensor t;
        t.name = name;
        t.shape.assign(shape, shape + rank);
        t.dtype = dtype;
        if (values) t.values.assign(values, values + numel);
        v->push_back(std::move(t));
        return 0;
    };
    const pcv_status st = pcv_checkpoint_visit(path.c_str(), visit, &out);
    if (st != PCV_OK) throw ModelError(st, pcv_last_error());
    return out;
}

// SentenceEmbeddingsTokenizerOuput of tokenize.rs:9-57: right-padded ids, mask = id != pad
struct TokenTensors {
    std::vector<int64_t> tokens_ids, tokens_masks;
    int batch = 0, len = 0;
};
inline TokenTensors generate_token_tensors(const std::vector<std::vector<int64_t>>& token_ids, int64_t pad_token_id = 0) {
    TokenTensors t;
    t.batch = (int)token_ids.size();
    for (auto& v : token_ids) t.len = std::max<int>(t.len, (int)v.size());
    t.tokens_ids.assign((size_t)t.batch * t.len, pad_token_id);
    for (int b = 0; b < t.batch; ++b)
        for (size_t i = 0; i < token_ids[b].size(); ++i) t.tokens_ids[(size_t)b * t.len + i] = token_ids[b][i];
    t.tokens_masks.resize(t.tokens_ids.size());
    for (size_t i = 0; i < t.tokens_ids.size(); ++i) t.tokens_masks[i] = t.tokens_ids[i] != pad_token_id;
    return t;
}

// model.rs:56-191
class Model {
public:
    SentenceEmbeddingsModelType model_type;  // pub field, model.rs:57

    Model(Context& ctx, const pcv_model_desc& desc, const char* weights_path, uint64_t synthetic_seed = 0,
          SentenceEmbeddingsModelType type = SentenceEmbeddingsModelType::AllMiniLmL6V2)
        : model_type(type) {
        pcv_status s = pcv_model_create(ctx.handle(), &desc, weights_path, synthetic_seed, &h_);
        if (s != PCV_OK) throw ModelError(s, pcv_last_error());
        check(pcv_model_output_dim(h_, &dim_));
    }
    // Model::new_pretrained (model.rs:68-174) from a sentence-transformers directory: configs, tokenizer and
    // model.safetensors are read by the library; the model owns its tokenizer
    Model(Context& ctx, const std::string& model_dir, SentenceEmbeddingsModelType type = SentenceEmbeddingsModelType::AllMiniLmL6V2,
          int compute = PCV_COMPUTE_F32, bool load_weights = true)
        : model_type(type) {
        pcv_status s = pcv_model_create_from_dir(ctx.handle(), model_dir.c_str(), compute, load_weights ? 1 : 0, &h_);
        if (s != PCV_OK) throw ModelError(s, pcv_last_error());
        check(pcv_model_output_dim(h_, &dim_));
    }
    ~Model() { pcv_model_destroy(h_); }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    // Model::encode(&[S]) (model.rs:176-179): tokenize + forward in one call
    std::vector<std::vector<float>> encode(const std::vector<std::string>& inputs) const {
        std::vector<const char*> ptrs;
        std::vector<size_t> lens;
        for (const auto& t : inputs) {
            ptrs.push_back(t.data());
            lens.push_back(t.size());
        }
        std::vector<float> flat(inputs.size() * (size_t)dim_);
        pcv_status s = pcv_model_encode_text(h_, ptrs.data(), lens.data(), (int)inputs.size(), flat.data());
        if (s != PCV_OK) throw ModelError(s, pcv_last_error());
        std::vector<std::vector<float>> out(inputs.size());
        for (size_t b = 0; b < inputs.size(); ++b) out[b].assign(flat.begin() + b * dim_, flat.begin() + (b + 1) * dim_);
        return out;
    }
    // Model::highlight (highlight.rs:23-165): views into `documents` (nullopt: the document gave no chunk)
    std::vector<std::optional<std::string_view>> highlight(const std::string& query, const std::vector<std::string>& documents) const {
        std::vector<const char*> ptrs;
        std::vector<size_t> lens;
        for (const auto& d : documents) {
            ptrs.push_back(d.data());
            lens.push_back(d.size());
        }
        std::vector<int64_t> b(documents.size(), -1), e(documents.size(), -1);
        pcv_status s = pcv_model_highlight(h_, query.data(), query.size(), ptrs.data(), lens.data(), (int)documents.size(), 0, -1,
                                           b.data(), e.data());
        if (s != PCV_OK) throw ModelError(s, pcv_last_error());
        std::vector<std::optional<std::string_view>> out(documents.size());
        for (size_t i = 0; i < documents.size(); ++i)
            if (b[i] >= 0) out[i] = std::string_view(documents[i]).substr((size_t)b[i], (size_t)(e[i] - b[i]));
        return out;
    }

    // Model::encode_tokens (model.rs:181-190 -> worker.rs:78-106); result rows = Vec<Vec<f32>>::from(Tensor)
    std::vector<std::vector<float>> encode_tokens(const TokenTensors& t) const {
        std::vector<float> flat((size_t)t.batch * dim_);
        pcv_status s = pcv_model_encode_tokens(h_, t.tokens_ids.data(), t.tokens_masks.data(), t.batch, t.len, flat.data());
        if (s != PCV_OK) throw ModelError(s, pcv_last_error());
        std::vector<std::vector<float>> out(t.batch);
        for (int b = 0; b < t.batch; ++b) out[b].assign(flat.begin() + (size_t)b * dim_, flat.begin() + (size_t)(b + 1) * dim_);
        return out;
    }
    int output_dim() const { return dim_; }

private:
    pcv_model* h_ = nullptr;
    int dim_ = 0;
};

// lib.rs:63-77, row-major [B][N] result
inline std::vector<float> dot_product(Context& ctx, const std::vector<float>& set1, int B, const std::vector<float>& set2,
                                      int64_t N, int dim) {
    std::vector<float> out((size_t)B * N);
    check(pcv_dot_product(ctx.handle(), set1.data(), B, set2.data(), N, dim, out.data()));
    return out;
}
inline std::vector<float> cosine_similarity_multi_query(Context& ctx, const std::vector<float>& set1, int B,
                                                        const std::vector<float>& set2, int64_t N, int dim) {
    std::vector<float> out((size_t)B * N);
    check(pcv_cosine_similarity(ctx.handle(), set1.data(), B, set2.data(), N, dim, out.data()));
    return out;
}
inline std::vector<float> cosine_similarity_single_query(Context& ctx, const std::vector<float>& query,
                                                         const std::vector<float>& matches, int64_t N, int dim) {
    return cosine_similarity_multi_query(ctx, query, 1, matches, N, dim);
}

}  // namespace perceive

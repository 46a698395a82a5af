// Drives the C++ host mirror (include/perceive.hpp) end to end on the GPU: build a Searcher from
// a blob row stream, search_vector with the reference's distance convention, rebuild_source, and a
// Model::encode_tokens round.  Expected values are recomputed here with plain f64 loops (the same
// canonical definition as oracle/scan.c, restated so this binary links only the product library).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>

#include "perceive.hpp"

using namespace perceive;

static int failures = 0;
#define EXPECT(cond)                                                      \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            ++failures;                                                   \
        }                                                                 \
    } while (0)

int main(int argc, char** argv) {
    Context ctx(0);
    const int D = 384, N = 3000;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd;
    std::vector<std::vector<float>> emb(N, std::vector<float>(D));
    std::vector<EmbeddingRow> rows;
    for (int i = 0; i < N; ++i) {
        for (auto& v : emb[i]) v = nd(rng);
        rows.push_back({1000 + i, i % 2 == 0 ? 1 : 2, serialize_embedding(emb[i])});
    }
    EXPECT(deserialize_embedding(rows[5].embedding) == emb[5]);
    auto s = Searcher::build(ctx, rows, D, Metric::Dot);
    EXPECT(s->num_rows() == N);

    std::vector<float> q(D);
    for (auto& v : q) v = nd(rng);
    auto expect = [&](int64_t source) {
        std::vector<std::pair<double, int64_t>> sc;
        for (int i = 0; i < N; ++i) {
            if (rows[i].source_id != source) continue;
            double dot = 0;
            for (int k = 0; k < D; ++k) dot += (double)q[k] * (double)emb[i][k];
            sc.push_back({dot, rows[i].item_id});
        }
        std::stable_sort(sc.begin(), sc.end(), [](auto& a, auto& b) { return a.first > b.first; });
        return sc;
    };
    for (int64_t source : {1, 2}) {
        auto items = s->search_vector({source}, 10, q);
        auto ref = expect(source);
        EXPECT(items.size() == 10);
        for (size_t j = 0; j < items.size(); ++j) {
            EXPECT(items[j].id == ref[j].second);
            const float d = (float)std::max(0.0, 1.0 - ref[j].first / D);  // search.rs:275-277
            EXPECT(std::fabs(items[j].score - d) < 1e-6f);
            if (j) EXPECT(items[j].score >= items[j - 1].score);  // ascending, search.rs:179
        }
    }
    EXPECT(s->search_vector({}, 10, q).empty());
    {  // a screening copy (int8) is kept by default and changes nothing but the bytes streamed
        auto with = s->search_vector({1, 2}, 10, q);
        const int64_t streamed8 = s->last_stats().bytes_streamed;  // per 32-row block: the int8 pieces (padded to 128 features) + the block's scale
        EXPECT(s->last_stats().screening_copy == 2 && streamed8 >= (int64_t)N * D && streamed8 < (int64_t)N * D * 2);
        s->set_screening_copy(PCV_SCREEN_COPY_OFF);
        auto without = s->search_vector({1, 2}, 10, q);
        EXPECT(s->last_stats().screening_copy == 0 && s->last_stats().bytes_streamed >= (int64_t)N * D * 4 && s->last_stats().bytes_streamed > 3 * streamed8);
        EXPECT(with.size() == without.size());
        for (size_t j = 0; j < with.size() && j < without.size(); ++j) EXPECT(with[j].id == without[j].id && with[j].score == without[j].score);
        s->set_screening_copy(PCV_SCREEN_COPY_AUTO);
    }
    {  // the sharded form at world size 1 (RCCL communicator owned by the library) gives the same items
        Comm comm(ctx, 1, 0, Comm::unique_id());
        auto a = s->search_vector({1}, 10, q), b = s->search_vector_sharded(comm, {1}, 10, q);
        EXPECT(a.size() == b.size());
        for (size_t j = 0; j < a.size() && j < b.size(); ++j) EXPECT(a[j].id == b[j].id && a[j].score == b[j].score);
    }
    // rebuild_source: source 2 shrinks to 5 rows
    std::vector<EmbeddingRow> repl(rows.begin() + 1, rows.begin() + 11);
    s->rebuild_source(repl, 2);
    EXPECT(s->num_rows() == N / 2 + 5);
    s->hidden.insert(42);

    // Model::encode_tokens on a tiny synthetic-weight model: unit-norm rows, deterministic
    pcv_model_desc d;
    pcv_model_desc_minilm_l6(&d);
    d.vocab_size = 1000;
    d.layers = 2;
    Model m(ctx, d, nullptr, 3);
    auto t = generate_token_tensors({{5, 6, 7, 8}, {9, 10}});
    EXPECT(t.len == 4 && t.tokens_masks[6] == 0 && t.tokens_masks[5] == 1);
    auto e1 = m.encode_tokens(t), e2 = m.encode_tokens(t);
    EXPECT(e1.size() == 2 && (int)e1[0].size() == m.output_dim() && e1 == e2);
    for (auto& row : e1) {
        double n = 0;
        for (float v : row) n += (double)v * v;
        EXPECT(std::fabs(n - 1.0) < 1e-5);
    }
    EXPECT(model_id(m.model_type) == 0);
    try {
        TokenTensors bad = generate_token_tensors({std::vector<int64_t>(600, 5)});
        m.encode_tokens(bad);
        EXPECT(false);
    } catch (const ModelError& e) {
        EXPECT(e.status == PCV_ERR_INVALID);
    }
    auto sim = cosine_similarity_single_query(ctx, e1[0], e1[0], 1, m.output_dim());
    EXPECT(std::fabs(sim[0] - 1.0f) < 1e-5f);

    // Model::new_pretrained / encode / highlight from a model directory (argv[1], written by the Python test:
    // JSON configs + vocab.txt; weights stay the seeded synthetic ones, load_weights = false)
    if (argc > 1) {
        Model tm(ctx, argv[1], SentenceEmbeddingsModelType::AllMiniLmL6V2, PCV_COMPUTE_F32, /*load_weights=*/false);
        std::vector<std::string> texts = {"Hello world", "the search of embeddings, really?", "Hello world"};
        auto emb = tm.encode(texts);
        EXPECT(emb.size() == 3 && (int)emb[0].size() == tm.output_dim());
        EXPECT(emb[0] == emb[2] && emb[0] != emb[1]);
        std::string filler;
        for (int i = 0; i < 12; ++i) filler += "people work each day and the years go on. ";
        std::vector<std::string> docs = {filler + "how good is the search model in the world today " + filler, "short one", ""};
        auto hl = tm.highlight("how good is the search model", docs);
        EXPECT(hl.size() == 3 && hl[0].has_value() && !hl[1].has_value() && !hl[2].has_value());
        if (hl[0]) EXPECT(hl[0]->data() >= docs[0].data() && hl[0]->data() + hl[0]->size() <= docs[0].data() + docs[0].size());
        try {
            Model missing(ctx, std::string(argv[1]) + "/nope");
            EXPECT(false);
        } catch (const ModelError& e) {
            EXPECT(e.status == PCV_ERR_IO);
        }
    }

    // a checkpoint file walked through the C ABI (argv[2]: the rust_model.ot fixture written by libtorch)
    if (argc > 2) {
        auto tensors = read_checkpoint(argv[2]);
        EXPECT(tensors.size() == 2 && tensors[0].name == "linear.weight" && tensors[1].name == "linear.bias");
        if (tensors.size() == 2) {
            EXPECT(tensors[0].shape == (std::vector<int64_t>{64, 128}) && tensors[0].values.size() == 64 * 128 && tensors[0].dtype == PCV_TENSOR_F32);
            EXPECT(tensors[1].shape == (std::vector<int64_t>{64}) && tensors[1].values.size() == 64);
        }
        try {
            read_checkpoint(std::string(argv[2]) + ".missing");
            EXPECT(false);
        } catch (const ModelError& e) {
            EXPECT(e.status == PCV_ERR_IO);
        }
    }

    std::printf(failures ? "host_mirror_test: %d failure(s)\n" : "host_mirror_test: ok\n", failures);
    return failures ? 1 : 0;
}

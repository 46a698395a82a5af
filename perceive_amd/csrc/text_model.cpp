// text_model.cpp — the text side of the Model C ABI: model directories, text -> embedding, highlight.
// Replaces, of the reference (crates/perceive-core):
//   model.rs:68-174        Model::new_pretrained      -> pcv_model_create_from_dir (+ pcv_model_load_hf_tensor)
//   model.rs:176-179       Model::encode(&[S])        -> pcv_model_encode_text
//   model/highlight.rs     Model::highlight           -> pcv_model_highlight
// Host code (JSON, safetensors, chunk planning, offsets) is plain C++; the encoder forward and the chunk
// scoring run on the GPU (model.cpp, encoder_kernels.hip).
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <functional>
#include <memory>
#include <sstream>
#include <thread>

#include "common.h"
#include "encoder.h"
#include "model_internal.h"
#include "tokenizer.h"
#include "torch_archive.h"

using namespace pcv;

namespace {

// ---- a small JSON document model (configs and the safetensors header; no dependency to link) ----
struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    double n = 0.0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;

    const JVal* get(const char* key) const {
        if (t != Obj) return nullptr;
        for (const auto& kv : o)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool truthy(const char* key, bool dflt) const {
        const JVal* v = get(key);
        if (!v || v->t == Null) return dflt;
        return v->t == Bool ? v->b : (v->t == Num ? v->n != 0.0 : dflt);
    }
    double num(const char* key, double dflt) const {
        const JVal* v = get(key);
        return (v && v->t == Num) ? v->n : dflt;
    }
    std::string str(const char* key, const std::string& dflt) const {
        const JVal* v = get(key);
        return (v && v->t == Str) ? v->s : dflt;
    }
};

struct JParser {
    const std::string& j;
    size_t i = 0;
    const std::string& what;
    [[noreturn]] void fail(const char* msg) const { PCV_FAIL(PCV_ERR_IO, "%s: malformed JSON at byte %zu (%s)", what.c_str(), i, msg); }
    void ws() {
        while (i < j.size() && (j[i] == ' ' || j[i] == '\t' || j[i] == '\n' || j[i] == '\r')) ++i;
    }
    static void utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) {
            out += (char)cp;
        } else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
    uint32_t hex4() {
        if (i + 4 > j.size()) fail("short \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = j[i++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        if (j[i] != '"') fail("expected a string");
        ++i;
        std::string out;
        while (true) {
            if (i >= j.size()) fail("unterminated string");
            const char c = j[i++];
            if (c == '"') return out;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (i >= j.size()) fail("unterminated escape");
            const char e = j[i++];
            switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF && i + 6 <= j.size() && j[i] == '\\' && j[i + 1] == 'u') {
                        i += 2;
                        const uint32_t lo = hex4();
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, cp);
                    break;
                }
                default: out += e;  // \" \\ \/
            }
        }
    }
    JVal value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (i >= j.size()) fail("unexpected end");
        JVal v;
        const char c = j[i];
        if (c == '{') {
            v.t = JVal::Obj;
            ++i;
            ws();
            if (i < j.size() && j[i] == '}') {
                ++i;
                return v;
            }
            while (true) {
                ws();
                std::string k = string();
                ws();
                if (i >= j.size() || j[i] != ':') fail("expected ':'");
                ++i;
                v.o.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (i < j.size() && j[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < j.size() && j[i] == '}') {
                    ++i;
                    return v;
                }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.t = JVal::Arr;
            ++i;
            ws();
            if (i < j.size() && j[i] == ']') {
                ++i;
                return v;
            }
            while (true) {
                v.a.push_back(value(depth + 1));
                ws();
                if (i < j.size() && j[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < j.size() && j[i] == ']') {
                    ++i;
                    return v;
                }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.t = JVal::Str;
            v.s = string();
            return v;
        }
        if (j.compare(i, 4, "true") == 0) {
            v.t = JVal::Bool;
            v.b = true;
            i += 4;
            return v;
        }
        if (j.compare(i, 5, "false") == 0) {
            v.t = JVal::Bool;
            i += 5;
            return v;
        }
        if (j.compare(i, 4, "null") == 0) {
            i += 4;
            return v;
        }
        const size_t b0 = i;
        while (i < j.size() && (std::isdigit((unsigned char)j[i]) || j[i] == '-' || j[i] == '+' || j[i] == '.' || j[i] == 'e' || j[i] == 'E')) ++i;
        if (i == b0) fail("unexpected character");
        v.t = JVal::Num;
        v.n = std::strtod(j.substr(b0, i - b0).c_str(), nullptr);
        return v;
    }
};

JVal parse_json(const std::string& text, const std::string& what) {
    JParser p{text, 0, what};
    JVal v = p.value(0);
    p.ws();
    if (p.i != text.size()) p.fail("trailing characters");
    return v;
}

bool file_exists(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return f.good();
}

JVal read_json_file(const std::string& path, bool optional) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        if (optional) return JVal{};
        PCV_FAIL(PCV_ERR_IO, "missing model file %s", path.c_str());
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_json(ss.str(), path);
}

std::string last_component(const std::string& dotted) {
    const size_t p = dotted.rfind('.');
    return p == std::string::npos ? dotted : dotted.substr(p + 1);
}

// ---- checkpoint tensor names -> the names of the encoder graph (model.cpp) -----------------------------
enum Arch { ARCH_BERT = 0, ARCH_DISTILBERT = 1, ARCH_ROBERTA = 2, ARCH_ALBERT = 3 };

void replace_all(std::string& s, const char* a, const char* b) {
    const size_t la = std::strlen(a), lb = std::strlen(b);
    for (size_t p = 0; (p = s.find(a, p)) != std::string::npos; p += lb) s.replace(p, la, b);
}

std::string graph_name(int arch, std::string k) {
    for (const char* pre : {"bert.", "roberta.", "distilbert.", "albert."})
        if (k.compare(0, std::strlen(pre), pre) == 0) k = k.substr(std::strlen(pre));
    if (arch == ARCH_DISTILBERT) {  // the same post-LayerNorm encoder under DistilBertModel's names
        replace_all(k, "transformer.layer.", "encoder.layer.");
        replace_all(k, ".attention.q_lin.", ".attention.self.query.");
        replace_all(k, ".attention.k_lin.", ".attention.self.key.");
        replace_all(k, ".attention.v_lin.", ".attention.self.value.");
        replace_all(k, ".attention.out_lin.", ".attention.output.dense.");
        replace_all(k, ".sa_layer_norm.", ".attention.output.LayerNorm.");
        replace_all(k, ".ffn.lin1.", ".intermediate.dense.");
        replace_all(k, ".ffn.lin2.", ".output.dense.");
        replace_all(k, ".output_layer_norm.", ".output.LayerNorm.");
    }
    if (arch == ARCH_ALBERT) {  // one shared layer (group 0, inner layer 0) under AlbertModel's names
        replace_all(k, "encoder.albert_layer_groups.0.albert_layers.0.", "encoder.layer.0.");
        if (k.compare(0, 16, "encoder.layer.0.") == 0) {
            replace_all(k, ".attention.query.", ".attention.self.query.");
            replace_all(k, ".attention.key.", ".attention.self.key.");
            replace_all(k, ".attention.value.", ".attention.self.value.");
            replace_all(k, ".attention.dense.", ".attention.output.dense.");
            replace_all(k, ".attention.LayerNorm.", ".attention.output.LayerNorm.");
            replace_all(k, ".ffn_output.", ".output.dense.");
            replace_all(k, ".ffn.", ".intermediate.dense.");
            replace_all(k, ".full_layer_layer_norm.", ".output.LayerNorm.");
        }
    }
    return k;
}

// One checkpoint tensor into the model; names the graph does not use are ignored.  The caller holds m->mu.
void load_hf_tensor(pcv_model* m, const std::string& hf_name, const float* data, int64_t numel) {
    const std::string name = graph_name(m->arch, hf_name);
    auto it = m->table.find(name);
    if (it == m->table.end()) return;  // pooler, position_ids, ...
    const float* src = data;
    int64_t n = numel;
    if (name == "embeddings.position_embeddings.weight" && m->pos_shift > 0) {
        // RoBERTa numbers positions from padding_idx + 1: the table is used from that row on
        const int64_t skip = (int64_t)m->pos_shift * m->d.hidden;
        PCV_REQUIRE(n > skip, "checkpoint tensor %s is shorter than the position shift", hf_name.c_str());
        src += skip;
        n -= skip;
    }
    if (n != it->second.n)
        PCV_FAIL(PCV_ERR_IO, "checkpoint tensor %s has %lld elements, the model expects %lld", hf_name.c_str(), (long long)n,
                 (long long)it->second.n);
    PCV_HIP(hipMemcpy(it->second.p, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    m->loaded[name] = true;
    m->planes_dirty = true;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else {  // subnormal
            int e = -1;
            uint32_t mm = man;
            do {
                ++e;
                mm <<= 1;
            } while (!(mm & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

using WantFn = std::function<bool(const std::string&)>;
using TensorFn = std::function<void(const ArchiveTensor&)>;

// safetensors: u64 LE header length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian data.
// Nothing in the file is executed; F32 / F16 / BF16 / F64 tensors are converted, others reported as AR_OTHER.
void read_safetensors(const std::string& path, const WantFn& want, const TensorFn& fn) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) PCV_FAIL(PCV_ERR_IO, "cannot open %s", path.c_str());
    struct Closer {
        FILE* f;
        ~Closer() { std::fclose(f); }
    } closer{f};
    uint8_t lenb[8];
    if (std::fread(lenb, 1, 8, f) != 8) PCV_FAIL(PCV_ERR_IO, "%s: truncated header", path.c_str());
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | lenb[i];
    if (hlen == 0 || hlen > (1ull << 27)) PCV_FAIL(PCV_ERR_IO, "%s: implausible header length %llu", path.c_str(), (unsigned long long)hlen);
    std::string header(hlen, '\0');
    if (std::fread(&header[0], 1, hlen, f) != hlen) PCV_FAIL(PCV_ERR_IO, "%s: truncated header", path.c_str());
    std::fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)std::ftell(f), base = 8 + hlen;
    const JVal doc = parse_json(header, path);
    if (doc.t != JVal::Obj) PCV_FAIL(PCV_ERR_IO, "%s: header is not a JSON object", path.c_str());
    std::vector<uint8_t> raw;
    ArchiveTensor out;
    for (const auto& kv : doc.o) {
        if (kv.first == "__metadata__") continue;
        const JVal& t = kv.second;
        const std::string dtype = t.str("dtype", "");
        const JVal* offs = t.get("data_offsets");
        const JVal* shape = t.get("shape");
        if (!offs || offs->t != JVal::Arr || offs->a.size() != 2 || !shape || shape->t != JVal::Arr)
            PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has no shape / data_offsets", path.c_str(), kv.first.c_str());
        if (!want(kv.first)) continue;
        out.name = kv.first;
        out.dtype_name = dtype;
        out.shape.clear();
        out.numel = 1;
        for (const JVal& d : shape->a) {  // JSON numbers are doubles: whole, in range, and every product checked
            if (d.t != JVal::Num || !(d.n >= 0.0) || d.n > 1099511627776.0 /* 2^40 */ || d.n != std::floor(d.n) ||
                __builtin_mul_overflow(out.numel, (int64_t)d.n, &out.numel) || out.numel > (int64_t)1 << 40)
                PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has an implausible shape", path.c_str(), kv.first.c_str());
            out.shape.push_back((int64_t)d.n);
        }
        for (const JVal& o : offs->a)
            if (o.t != JVal::Num || !(o.n >= 0.0) || o.n > 9007199254740992.0 /* 2^53 */ || o.n != std::floor(o.n))
                PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has inconsistent offsets", path.c_str(), kv.first.c_str());
        const uint64_t a = (uint64_t)offs->a[0].n, b = (uint64_t)offs->a[1].n;
        const uint64_t esize = dtype == "F32" ? 4 : (dtype == "F16" || dtype == "BF16") ? 2 : dtype == "F64" ? 8 : 0;
        out.dtype = dtype == "F32" ? AR_F32 : dtype == "F16" ? AR_F16 : dtype == "BF16" ? AR_BF16 : dtype == "F64" ? AR_F64 : AR_OTHER;
        out.values.clear();
        if (esize != 0) {
            if (b < a || b - a != (uint64_t)out.numel * esize || base + b > fsize)
                PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has inconsistent offsets", path.c_str(), kv.first.c_str());
            raw.resize((size_t)(b - a));
            std::fseek(f, (long)(base + a), SEEK_SET);
            if (std::fread(raw.data(), 1, raw.size(), f) != raw.size()) PCV_FAIL(PCV_ERR_IO, "%s: truncated data of %s", path.c_str(), kv.first.c_str());
            out.values.resize((size_t)out.numel);
            if (esize == 4) {
                std::memcpy(out.values.data(), raw.data(), raw.size());  // little-endian host
            } else if (esize == 8) {
                for (int64_t i = 0; i < out.numel; ++i) {
                    double d;
                    std::memcpy(&d, raw.data() + 8 * i, 8);
                    out.values[(size_t)i] = (float)d;
                }
            } else {
                for (int64_t i = 0; i < out.numel; ++i) {
                    const uint16_t h = (uint16_t)raw[2 * i] | ((uint16_t)raw[2 * i + 1] << 8);
                    if (out.dtype == AR_BF16) {
                        const uint32_t u = (uint32_t)h << 16;
                        std::memcpy(&out.values[(size_t)i], &u, 4);
                    } else {
                        out.values[(size_t)i] = half_to_float(h);
                    }
                }
            }
        }
        fn(out);
    }
}

// A checkpoint in any of the three formats a model directory may hold: `model.safetensors`, the reference's
// `rust_model.ot` (configs.rs:109,112) or `pytorch_model.bin`; told apart by the zip signature.
void read_checkpoint(const std::string& path, const WantFn& want, const TensorFn& fn) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) PCV_FAIL(PCV_ERR_IO, "cannot open %s", path.c_str());
    uint8_t magic[4] = {0, 0, 0, 0};
    const size_t got = std::fread(magic, 1, 4, f);
    std::fclose(f);
    if (got == 4 && magic[0] == 'P' && magic[1] == 'K' && (magic[2] == 3 || magic[2] == 5) && (magic[3] == 4 || magic[3] == 6))
        read_torch_archive(path, want, fn);
    else
        read_safetensors(path, want, fn);
}

// Every tensor of `path` the graph has a place for, under `rename_prefix` + its checkpoint name.
void load_checkpoint(pcv_model* m, const std::string& path, const char* rename_prefix) {
    const std::string prefix(rename_prefix);
    read_checkpoint(
        path, [&](const std::string& name) { return m->table.count(graph_name(m->arch, prefix + name)) != 0; },
        [&](const ArchiveTensor& t) {
            if (t.dtype == AR_OTHER)
                PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: tensor %s has dtype %s (32/16-bit floats and doubles are read)", path.c_str(), t.name.c_str(),
                         t.dtype_name.c_str());
            PCV_REQUIRE(t.numel > 0, "%s: tensor %s is empty", path.c_str(), t.name.c_str());
            load_hf_tensor(m, prefix + t.name, t.values.data(), t.numel);
        });
}

// The weights file of a module directory, in the order: what the reference reads, then what Hugging Face ships.
std::string find_weights(const std::string& dir) {
    for (const char* name : {"rust_model.ot", "model.safetensors", "pytorch_model.bin"})
        if (file_exists(dir + name)) return dir + name;
    return std::string();
}

void require_all_loaded(pcv_model* m, const std::string& where) {
    std::string missing;
    int n = 0;
    for (const auto& kv : m->table)
        if (!m->loaded.count(kv.first)) {
            if (n < 3) missing += (n ? ", " : "") + kv.first;
            ++n;
        }
    if (n) PCV_FAIL(PCV_ERR_IO, "%s: the checkpoint lacks %d tensors the model needs, e.g. %s", where.c_str(), n, missing.c_str());
}

// CPUs this process may use: affinity mask capped by the cgroup quota (tokenization threads)
int host_threads() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) n = std::min<long long>(n, std::max<long long>(1, (std::atoll(quota.c_str()) + period / 2) / period));
    return std::max(1, std::min(n, 64));
}

// run fn(i) for i in [0, n) on up to `threads` host threads; the first exception is rethrown
template <class F>
void parallel_for(int n, int threads, F&& fn) {
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    std::vector<std::thread> pool;
    for (int w = 0; w < threads; ++w)
        pool.emplace_back([&] {
            try {
                for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!err) err = std::current_exception();
            }
        });
    for (auto& t : pool) t.join();
    if (err) std::rethrow_exception(err);
}

void require_tokenizer(const pcv_model* m, const char* who) {
    PCV_REQUIRE(m->tok != nullptr, "%s: this model has no tokenizer (pcv_model_create_from_dir or pcv_model_set_tokenizer)", who);
}

// encode_list(texts, max_len, LongestFirst, 0) + generate_token_tensors (tokenize.rs:60-77, 9-57) into a padded
// [n][L] id matrix, L = longest row; mask = id != pad.
void tokenize_batch(const pcv_model* m, const char* const* texts, const size_t* n_bytes, int n, std::vector<int64_t>& ids,
                    std::vector<int64_t>& mask, int* out_L) {
    const int max_len = std::max(2, m->d.max_seq_length);
    const TokSpecials sp = tokenizer_specials(m->tok);
    std::vector<std::vector<int64_t>> rows((size_t)n);
    parallel_for(n, host_threads(), [&](int i) {
        std::vector<TokPiece> pieces = tokenizer_pieces(m->tok, texts[i], n_bytes[i]);
        if (pieces.size() > (size_t)max_len - 2) pieces.resize((size_t)max_len - 2);  // LongestFirst on one sequence: drop the tail
        auto& r = rows[(size_t)i];
        r.reserve(pieces.size() + 2);
        r.push_back(sp.cls);
        for (const TokPiece& p : pieces) r.push_back(p.id);
        r.push_back(sp.sep);
    });
    size_t L = 0;
    for (const auto& r : rows) L = std::max(L, r.size());
    ids.assign((size_t)n * L, m->pad_id);
    for (int i = 0; i < n; ++i) std::copy(rows[(size_t)i].begin(), rows[(size_t)i].end(), ids.begin() + (size_t)i * L);
    mask.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) mask[i] = ids[i] != m->pad_id ? 1 : 0;  // tokenize.rs:36-46
    *out_L = (int)L;
}

int env_usize(const char* name, int dflt) {  // highlight.rs:7-18: unset / unparsable -> default
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    const long long x = std::strtoll(v, &end, 10);
    if (*end != '\0' || x < 0 || x > 1000000) return dflt;
    return (int)x;
}

struct Chunk {
    int doc;
    int lo, hi;  // token range inside the document's token list ([CLS] at 0)
};

// Chunk plan of one document (highlight.rs:53-100): windows of `size` tokens every `size - overlap` tokens
// while `start + overlap < n`; inside a window the longest run of non-special tokens is the chunk, kept
// when it is at least size/2 long.  Runs are found from the positions of the special tokens (normally just
// [CLS] and [SEP]) instead of a pass over every token of every window.  Two details of the reference are
// kept because they decide which tokens a chunk holds: a run that follows a special token is counted from
// that token's own index (so the chunk starts ON the special token and stops one short of the run's end),
// and of two equally long runs the first wins.
void plan_chunks(int doc, const std::vector<uint8_t>& special, int size, int overlap, std::vector<Chunk>& out) {
    const int n = (int)special.size();
    std::vector<int> sp;
    for (int i = 0; i < n; ++i)
        if (special[i]) sp.push_back(i);
    const int inc = size - overlap;
    size_t s0 = 0;  // first special position >= window start
    for (int w0 = 0; w0 + overlap < n; w0 += inc) {
        const int w1 = std::min(w0 + size, n);
        while (s0 < sp.size() && sp[s0] < w0) ++s0;
        int best_start = w0, best_len = 0;
        int run_start = w0, prev = w0 - 1;  // `prev`: index before the run's first token
        bool first = true;
        auto close_run = [&](int end_excl) {  // run of non-special tokens (prev, end_excl)
            const int len = end_excl - prev - 1;
            const int start = first ? w0 : run_start;
            if (len > best_len) {
                best_len = len;
                best_start = start;
            }
        };
        for (size_t k = s0; k < sp.size() && sp[k] < w1; ++k) {
            close_run(sp[k]);
            first = false;
            run_start = sp[k];  // the reference restarts its run counter AT the special token
            prev = sp[k];
        }
        close_run(w1);
        const int lo = best_start, hi = std::min(best_start + best_len, w1);
        if (hi - lo >= size / 2) out.push_back(Chunk{doc, lo, hi});
    }
}

// byte offset of char #idx of a UTF-8 string, or -1 if the string has no such char
int64_t byte_of_char(const char* s, size_t n, int64_t idx) {
    int64_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        if (((unsigned char)s[i] & 0xC0) == 0x80) continue;  // continuation byte
        if (c == idx) return (int64_t)i;
        ++c;
    }
    return -1;
}

// Everything model.rs:84-151 reads from a sentence-transformers model directory.
struct ParsedDir {
    pcv_model_desc d{};
    int arch = ARCH_BERT, pos_shift = 0;
    bool lower = true, add_prefix_space = false;
    int strip_accents = -1;
    std::string dense_path;  // "" = no Dense module
    bool dense_bias = true;
};

ParsedDir parse_model_dir(const char* model_dir) {
    ParsedDir pd;
    pcv_model_desc& d = pd.d;
    const std::string dir = std::string(model_dir) + "/";
    // modules.json (model.rs:84-86): Transformer first, then Pooling, optional Dense, optional Normalize
    const JVal modules = read_json_file(dir + "modules.json", false);
    if (modules.t != JVal::Arr || modules.a.empty()) PCV_FAIL(PCV_ERR_IO, "%smodules.json: expected a list of modules", dir.c_str());
    std::string pooling_path = "1_Pooling";
    bool normalize = false;
    for (size_t i = 0; i < modules.a.size(); ++i) {
        const std::string kind = last_component(modules.a[i].str("type", ""));
        if (i == 0 && kind != "Transformer")
            PCV_FAIL(PCV_ERR_IO, "%smodules.json: the first module must be a Transformer, got '%s'", dir.c_str(), kind.c_str());
        if (kind == "Pooling") pooling_path = modules.a[i].str("path", pooling_path);
        if (kind == "Dense") pd.dense_path = modules.a[i].str("path", "2_Dense");
        if (kind == "Normalize") normalize = true;  // has_normalization(), model.rs:151
    }
    // config.json (model.rs:118-121)
    const JVal cfg = read_json_file(dir + "config.json", false);
    const std::string mt = cfg.str("model_type", "bert");
    if (mt == "bert") pd.arch = ARCH_BERT;
    else if (mt == "distilbert") pd.arch = ARCH_DISTILBERT;
    else if (mt == "roberta") pd.arch = ARCH_ROBERTA;
    else if (mt == "albert") pd.arch = ARCH_ALBERT;
    else PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: transformer type '%s' is not supported (BERT, DistilBERT, RoBERTa and ALBERT are)", model_dir, mt.c_str());
    const std::string act = cfg.str("hidden_act", cfg.str("activation", pd.arch == ARCH_ALBERT ? "gelu_new" : "gelu"));
    if (act == "gelu") d.hidden_act = PCV_GELU_ERF;
    else if (act == "gelu_new" || act == "gelu_pytorch_tanh" || act == "gelu_fast") d.hidden_act = PCV_GELU_TANH;  // one formula, three names
    else PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: activation '%s' is not supported (gelu and gelu_new are)", model_dir, act.c_str());
    d.vocab_size = (int)cfg.num("vocab_size", 0);
    d.max_positions = (int)cfg.num("max_position_embeddings", 0);
    if (pd.arch == ARCH_DISTILBERT) {
        if (cfg.truthy("sinusoidal_pos_embds", false)) PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: sinusoidal position embeddings are not supported", model_dir);
        d.hidden = (int)cfg.num("dim", 0);
        d.layers = (int)cfg.num("n_layers", 0);
        d.heads = (int)cfg.num("n_heads", 0);
        d.intermediate = (int)cfg.num("hidden_dim", 0);
        d.type_vocab = 1;
        d.layer_norm_eps = 1e-12f;
    } else {
        d.hidden = (int)cfg.num("hidden_size", 0);
        d.layers = (int)cfg.num("num_hidden_layers", 0);
        d.heads = (int)cfg.num("num_attention_heads", 0);
        d.intermediate = (int)cfg.num("intermediate_size", 0);
        d.type_vocab = (int)cfg.num("type_vocab_size", 2);
        d.layer_norm_eps = (float)cfg.num("layer_norm_eps", 1e-12);
    }
    if (pd.arch == ARCH_ALBERT) {
        // factorised embeddings and one set of layer weights run num_hidden_layers times
        d.embedding_size = (int)cfg.num("embedding_size", 128);
        d.shared_layers = 1;
        const int groups = (int)cfg.num("num_hidden_groups", 1), inner = (int)cfg.num("inner_group_num", 1);
        if (groups != 1 || inner != 1)
            PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: ALBERT with %d layer groups of %d inner layers is not supported (1 x 1 is)", model_dir, groups, inner);
    }
    if (pd.arch == ARCH_ROBERTA) {
        // RoBERTa numbers positions from padding_idx + 1 (pad tokens sit at padding_idx): for right-padded
        // batches token l has position l + pad + 1, so the table is used from that row on
        pd.pos_shift = (int)cfg.num("pad_token_id", 1) + 1;
        d.max_positions -= pd.pos_shift;
    }
    // sentence_bert_config.json, tokenizer_config.json (model.rs:90-95)
    const JVal sbert = read_json_file(dir + "sentence_bert_config.json", true);
    const JVal tokc = read_json_file(dir + "tokenizer_config.json", true);
    d.max_seq_length = (int)sbert.num("max_seq_length", 128);
    const JVal* dl = tokc.get("do_lower_case");
    pd.lower = (dl && dl->t == JVal::Bool) ? dl->b : sbert.truthy("do_lower_case", true);  // model.rs:108-110
    const JVal* sa = tokc.get("strip_accents");
    pd.strip_accents = (sa && sa->t == JVal::Bool) ? (sa->b ? 1 : 0) : -1;
    pd.add_prefix_space = tokc.truthy("add_prefix_space", false);
    // pooling (model.rs:134-135)
    const JVal pool = read_json_file(dir + pooling_path + "/config.json", false);
    d.pooling = pool.truthy("pooling_mode_cls_token", false)              ? PCV_POOL_CLS
                : pool.truthy("pooling_mode_max_tokens", false)           ? PCV_POOL_MAX
                : pool.truthy("pooling_mode_mean_sqrt_len_tokens", false) ? PCV_POOL_MEAN_SQRT_LEN
                                                                          : PCV_POOL_MEAN;
    d.normalize = normalize ? 1 : 0;
    // Dense module (model.rs:139-149)
    if (!pd.dense_path.empty()) {
        const JVal dc = read_json_file(dir + pd.dense_path + "/config.json", false);
        std::string a = last_component(dc.str("activation_function", "torch.nn.modules.linear.Identity"));
        std::transform(a.begin(), a.end(), a.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        if (a == "tanh") d.dense_activation = PCV_ACT_TANH;
        else if (a == "identity") d.dense_activation = PCV_ACT_IDENTITY;
        else PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: Dense activation '%s' is not supported", model_dir, a.c_str());
        d.dense_out = (int)dc.num("out_features", 0);
        pd.dense_bias = dc.truthy("bias", true);
        PCV_REQUIRE(d.dense_out > 0, "%s: Dense module without out_features", model_dir);
    }
    d.compute = PCV_COMPUTE_F32;
    return pd;
}

}  // namespace

extern "C" {

const char* pcv_model_type_dir_name(int model_type) {
    // configs.rs:30-39 (variants) and 42-69, 121-141 (their sentence-transformers directory names)
    static const char* kNames[] = {"all-MiniLM-L6-v2",     "all-MiniLM-L12-v2",          "distiluse-base-multilingual-cased",
                                   "all-distilroberta-v1", "paraphrase-albert-small-v2", "msmarco-distilbert-dot-v5",
                                   "msmarco-distilbert-base-tas-b", "msmarco-bert-base-dot-v5"};
    return (model_type >= 0 && model_type < 8) ? kNames[model_type] : nullptr;
}

pcv_status pcv_model_dir_describe(const char* model_dir, pcv_model_desc* out_desc, int* out_arch, int* out_lower_case,
                                  int* out_strip_accents) {
    return guarded([&] {
        PCV_REQUIRE(model_dir != nullptr, "model_dir_describe: NULL argument");
        const ParsedDir pd = parse_model_dir(model_dir);
        if (out_desc) *out_desc = pd.d;
        if (out_arch) *out_arch = pd.arch;
        if (out_lower_case) *out_lower_case = pd.lower ? 1 : 0;
        if (out_strip_accents) *out_strip_accents = pd.strip_accents;
    });
}

pcv_status pcv_model_create_from_dir(pcv_ctx* ctx, const char* model_dir, int compute, int load_weights, pcv_model** out) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && model_dir != nullptr && out != nullptr, "model_create_from_dir: NULL argument");
        *out = nullptr;
        ParsedDir pd = parse_model_dir(model_dir);
        pd.d.compute = compute;
        const std::string dir = std::string(model_dir) + "/";
        // tokenizer (model.rs:96-113)
        pcv_tokenizer* tok = nullptr;
        pcv_status st = pd.arch == ARCH_ROBERTA
                            ? pcv_tokenizer_create_bpe((dir + "vocab.json").c_str(), (dir + "merges.txt").c_str(), pd.add_prefix_space ? 1 : 0, &tok)
                        : pd.arch == ARCH_ALBERT
                            ? pcv_tokenizer_create_sentencepiece((dir + "spiece.model").c_str(), pd.lower ? 1 : 0, pd.strip_accents, &tok)
                            : pcv_tokenizer_create((dir + "vocab.txt").c_str(), pd.lower ? 1 : 0, pd.strip_accents, &tok);
        if (st != PCV_OK) throw Error{st};
        pcv_model* m = nullptr;
        st = pcv_model_create(ctx, &pd.d, nullptr, 0, &m);
        if (st != PCV_OK) {
            pcv_tokenizer_destroy(tok);
            throw Error{st};
        }
        m->tok = tok;
        m->own_tok = true;
        const TokSpecials sp = tokenizer_specials(tok);
        m->pad_id = sp.pad >= 0 ? sp.pad : 0;  // tokenize.rs:19
        m->arch = pd.arch;
        m->pos_shift = pd.pos_shift;
        try {
            PCV_HIP(hipSetDevice(ctx->device));
            if (pd.arch == ARCH_DISTILBERT) {  // DistilBERT has no token-type embeddings: one zero row
                PCV_HIP(hipMemset(m->type.p, 0, (size_t)m->type.n * sizeof(float)));
                m->loaded["embeddings.token_type_embeddings.weight"] = true;
            }
            if (!pd.dense_path.empty() && !pd.dense_bias) {
                PCV_HIP(hipMemset(m->dense_b.p, 0, (size_t)m->dense_b.n * sizeof(float)));
                m->loaded["dense.linear.bias"] = true;
            }
            if (load_weights) {
                const std::string wpath = find_weights(dir);
                if (wpath.empty())
                    PCV_FAIL(PCV_ERR_IO, "%s: no rust_model.ot, model.safetensors or pytorch_model.bin", model_dir);
                std::lock_guard<std::mutex> lk(m->mu);
                load_checkpoint(m, wpath, "");
                if (!pd.dense_path.empty()) {
                    const std::string dpath = find_weights(dir + pd.dense_path + "/");
                    if (dpath.empty()) PCV_FAIL(PCV_ERR_IO, "%s: the Dense module %s holds no weights file", model_dir, pd.dense_path.c_str());
                    load_checkpoint(m, dpath, "dense.");
                }
                require_all_loaded(m, model_dir);
            }
        } catch (...) {
            pcv_model_destroy(m);
            throw;
        }
        *out = m;
    });
}

pcv_status pcv_checkpoint_visit(const char* path, pcv_tensor_visitor visit, void* user) {
    return guarded([&] {
        PCV_REQUIRE(path != nullptr && visit != nullptr, "checkpoint_visit: NULL argument");
        read_checkpoint(
            path, [](const std::string&) { return true; },
            [&](const ArchiveTensor& t) {
                const float* values = t.dtype == AR_OTHER || t.values.empty() ? nullptr : t.values.data();
                if (visit(user, t.name.c_str(), t.shape.data(), (int)t.shape.size(), t.dtype, values, t.numel) != 0)
                    PCV_FAIL(PCV_ERR_INVALID, "checkpoint_visit: stopped by the visitor at %s", t.name.c_str());
            });
    });
}

pcv_status pcv_model_load_hf_tensor(pcv_model* m, const char* hf_name, const float* data, int64_t numel) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && hf_name != nullptr && data != nullptr && numel > 0, "model_load_hf_tensor: bad argument");
        std::lock_guard<std::mutex> lk(m->mu);
        PCV_HIP(hipSetDevice(m->ctx->device));
        PCV_HIP(hipStreamSynchronize(m->ctx->stream));
        load_hf_tensor(m, hf_name, data, numel);
    });
}

pcv_status pcv_model_check_loaded(pcv_model* m) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr, "model_check_loaded: model is NULL");
        std::lock_guard<std::mutex> lk(m->mu);
        require_all_loaded(m, "model_check_loaded");
    });
}

pcv_status pcv_model_set_tokenizer(pcv_model* m, pcv_tokenizer* t, int take_ownership) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr, "model_set_tokenizer: model is NULL");
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->tok && m->own_tok && m->tok != t) pcv_tokenizer_destroy(m->tok);
        m->tok = t;
        m->own_tok = t != nullptr && take_ownership != 0;
        m->pad_id = 0;
        if (t) {
            const TokSpecials sp = tokenizer_specials(t);
            if (sp.pad >= 0) m->pad_id = sp.pad;  // tokenize.rs:19
        }
    });
}

pcv_status pcv_model_get_desc(pcv_model* m, pcv_model_desc* out_desc, int64_t* out_pad_id) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr, "model_get_desc: model is NULL");
        std::lock_guard<std::mutex> lk(m->mu);
        if (out_desc) *out_desc = m->d;
        if (out_pad_id) *out_pad_id = m->pad_id;
    });
}

pcv_status pcv_model_tokenizer(pcv_model* m, pcv_tokenizer** out_tok) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && out_tok != nullptr, "model_tokenizer: NULL argument");
        std::lock_guard<std::mutex> lk(m->mu);
        *out_tok = m->tok;
    });
}

pcv_status pcv_model_encode_text(pcv_model* m, const char* const* texts, const size_t* n_bytes, int n_texts, float* out) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && n_texts >= 0 && (n_texts == 0 || (texts && n_bytes && out)), "model_encode_text: bad argument");
        if (n_texts == 0) return;
        std::lock_guard<std::mutex> lk(m->mu);  // one forward at a time, like the worker channel (model.rs:161,187)
        require_tokenizer(m, "model_encode_text");
        PCV_HIP(hipSetDevice(m->ctx->device));
        const int OD = m->d.dense_out > 0 ? m->d.dense_out : m->d.hidden;
        constexpr int kStep = 1024;  // texts per forward (bounds the workspace; padding is per forward and changes nothing)
        std::vector<int64_t> ids, mask;
        for (int t0 = 0; t0 < n_texts; t0 += kStep) {
            const int B = std::min(kStep, n_texts - t0);
            int L = 0;
            tokenize_batch(m, texts + t0, n_bytes + t0, B, ids, mask, &L);
            model_check_tokens(m, ids.data(), mask.data(), B, L);
            model_forward(m, ids.data(), mask.data(), B, L);
            PCV_HIP(hipMemcpyAsync(out + (size_t)t0 * OD, m->out, (size_t)B * OD * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
            PCV_HIP(hipStreamSynchronize(m->ctx->stream));
            PCV_HIP(hipGetLastError());
            model_finish_stats(m);
        }
        model_check_f16_output(m, out, (size_t)n_texts * OD);
    });
}

pcv_status pcv_model_highlight(pcv_model* m, const char* query, size_t query_bytes, const char* const* docs, const size_t* doc_bytes,
                               int n_docs, int chunk_size, int chunk_overlap, int64_t* out_begin, int64_t* out_end) {
    return guarded([&] {
        PCV_REQUIRE(m != nullptr && query != nullptr && n_docs >= 0 && (n_docs == 0 || (docs && doc_bytes && out_begin && out_end)),
                    "model_highlight: bad argument");
        if (chunk_size <= 0) chunk_size = env_usize("CHUNK_SIZE", 20);       // highlight.rs:7-18
        if (chunk_overlap < 0) chunk_overlap = env_usize("CHUNK_OVERLAP", 4);
        PCV_REQUIRE(chunk_size > chunk_overlap, "model_highlight: chunk size %d must exceed the overlap %d", chunk_size, chunk_overlap);
        if (n_docs == 0) return;
        std::lock_guard<std::mutex> lk(m->mu);
        require_tokenizer(m, "model_highlight");
        PCV_HIP(hipSetDevice(m->ctx->device));
        hipStream_t st = m->ctx->stream;
        const int OD = m->d.dense_out > 0 ? m->d.dense_out : m->d.hidden;
        const TokSpecials sp = tokenizer_specials(m->tok);

        // documents: full token lists with offsets (no truncation: chunked below, highlight.rs:32-38), in parallel
        struct Doc {
            std::vector<int64_t> ids;
            std::vector<int32_t> begin, end;
            std::vector<uint8_t> special;
        };
        std::vector<Doc> td((size_t)n_docs);
        parallel_for(n_docs, host_threads(), [&](int i) {
            std::vector<TokPiece> pieces = tokenizer_pieces(m->tok, docs[i], doc_bytes[i]);
            if (pieces.size() > 999998) pieces.resize(999998);  // the reference's max_len of 1_000_000
            Doc& d = td[(size_t)i];
            const size_t n = pieces.size() + 2;
            d.ids.resize(n);
            d.begin.assign(n, -1);
            d.end.assign(n, -1);
            d.special.assign(n, 0);
            d.ids[0] = sp.cls;
            d.special[0] = 1;
            for (size_t k = 0; k < pieces.size(); ++k) {
                d.ids[k + 1] = pieces[k].id;
                d.begin[k + 1] = pieces[k].begin;
                d.end[k + 1] = pieces[k].end;
                d.special[k + 1] = pieces[k].special;
            }
            d.ids[n - 1] = sp.sep;
            d.special[n - 1] = 1;
        });
        std::vector<Chunk> chunks;
        std::vector<int32_t> bounds((size_t)n_docs + 1, 0);
        for (int i = 0; i < n_docs; ++i) {
            plan_chunks(i, td[(size_t)i].special, chunk_size, chunk_overlap, chunks);
            bounds[(size_t)i + 1] = (int32_t)chunks.size();
        }
        const size_t C = chunks.size();

        // the query (highlight.rs:29) and every chunk (highlight.rs:102-108) through the encoder; embeddings stay on the device
        if (m->hl_docs_cap < (size_t)n_docs) {
            if (m->hl_bounds) (void)hipFree(m->hl_bounds);
            if (m->hl_best) (void)hipFree(m->hl_best);
            m->hl_bounds = m->hl_best = nullptr;
            m->hl_docs_cap = 0;
            PCV_HIP(hipMalloc((void**)&m->hl_bounds, ((size_t)n_docs + 1) * sizeof(int32_t)));
            PCV_HIP(hipMalloc((void**)&m->hl_best, (size_t)n_docs * 2 * sizeof(int32_t)));
            m->hl_docs_cap = (size_t)n_docs;
        }
        if (!m->hl_query) PCV_HIP(hipMalloc((void**)&m->hl_query, (size_t)OD * sizeof(float)));
        if (m->hl_emb_cap < C) {
            if (m->hl_emb) (void)hipFree(m->hl_emb);
            m->hl_emb = nullptr;
            m->hl_emb_cap = 0;
            PCV_HIP(hipMalloc((void**)&m->hl_emb, std::max<size_t>(C, 64) * OD * sizeof(float)));
            m->hl_emb_cap = std::max<size_t>(C, 64);
        }
        std::vector<int64_t> ids, mask;
        {
            int L = 0;
            const char* qs[1] = {query};
            const size_t qb[1] = {query_bytes};
            tokenize_batch(m, qs, qb, 1, ids, mask, &L);
            model_check_tokens(m, ids.data(), mask.data(), 1, L);
            model_forward(m, ids.data(), mask.data(), 1, L);
            PCV_HIP(hipMemcpyAsync(m->hl_query, m->out, (size_t)OD * sizeof(float), hipMemcpyDeviceToDevice, st));
            PCV_HIP(hipStreamSynchronize(st));  // `ids` / `mask` are refilled below
        }
        constexpr size_t kStep = 2048;  // chunks per forward
        for (size_t c0 = 0; c0 < C; c0 += kStep) {
            const size_t B = std::min(kStep, C - c0);
            size_t L = 0;
            for (size_t c = c0; c < c0 + B; ++c) L = std::max<size_t>(L, (size_t)(chunks[c].hi - chunks[c].lo));
            ids.assign(B * L, m->pad_id);  // generate_token_tensors, tokenize.rs:9-57
            for (size_t c = 0; c < B; ++c) {
                const Chunk& ch = chunks[c0 + c];
                std::copy(td[(size_t)ch.doc].ids.begin() + ch.lo, td[(size_t)ch.doc].ids.begin() + ch.hi, ids.begin() + c * L);
            }
            mask.resize(ids.size());
            for (size_t i = 0; i < ids.size(); ++i) mask[i] = ids[i] != m->pad_id ? 1 : 0;
            model_check_tokens(m, ids.data(), mask.data(), (int)B, (int)L);
            model_forward(m, ids.data(), mask.data(), (int)B, (int)L);
            PCV_HIP(hipMemcpyAsync(m->hl_emb + c0 * OD, m->out, B * OD * sizeof(float), hipMemcpyDeviceToDevice, st));
            PCV_HIP(hipStreamSynchronize(st));
        }
        // dot_product(query, chunks) and the per-document best chunk, on the device (highlight.rs:109-127)
        std::vector<int32_t> best((size_t)n_docs * 2, -1);
        PCV_HIP(hipMemcpyAsync(m->hl_bounds, bounds.data(), bounds.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        launch_chunk_argmax(st, m->hl_query, m->hl_emb, OD, m->hl_bounds, n_docs, m->hl_best, m->hl_best + n_docs);
        PCV_HIP(hipMemcpyAsync(best.data(), m->hl_best, best.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PCV_HIP(hipStreamSynchronize(st));
        PCV_HIP(hipGetLastError());
        // back to text (highlight.rs:129-160)
        for (int i = 0; i < n_docs; ++i) {
            out_begin[i] = out_end[i] = -1;  // None: the document gave no chunk
            const int32_t bc = best[(size_t)i];
            if (bounds[(size_t)i] == bounds[(size_t)i + 1]) continue;
            if (best[(size_t)n_docs + i] != 0 || bc < 0)
                PCV_FAIL(PCV_ERR_INVALID, "model_highlight: NaN chunk score in document %d (the reference panics on it)", i);
            const Chunk& ch = chunks[(size_t)bc];
            const Doc& d = td[(size_t)i];
            // span of the chunk's tokens that have offsets: starts as the first one, then min / max
            int64_t tb = 0, te = 0;
            for (int k = ch.lo; k < ch.hi; ++k) {
                if (d.begin[(size_t)k] < 0) continue;  // special token: no offset
                if (tb == 0 && te == 0) {
                    tb = d.begin[(size_t)k];
                    te = d.end[(size_t)k];
                } else {
                    tb = std::min<int64_t>(tb, d.begin[(size_t)k]);
                    te = std::max<int64_t>(te, d.end[(size_t)k]);
                }
            }
            // char_indices().nth(tb), then .nth(te - tb) on the same iterator: chars #tb and #(te + 1) must both
            // exist, else the reference's Option::zip is None and the highlight is ""
            const int64_t b0 = byte_of_char(docs[i], doc_bytes[i], tb);
            const int64_t b1 = byte_of_char(docs[i], doc_bytes[i], te + 1);
            if (b0 >= 0 && b1 >= 0) {
                out_begin[i] = b0;
                out_end[i] = b1;
            } else {
                out_begin[i] = out_end[i] = 0;  // Some("")
            }
        }
    });
}

}  // extern "C"

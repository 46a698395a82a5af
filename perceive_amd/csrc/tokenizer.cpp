// tokenizer.cpp — BERT WordPiece tokenizer on the host (replaces rust_tokenizers' BertTokenizer as
// used by model/tokenize.rs:60-77 and model/highlight.rs:32-38).  The reference tokenizes on the
// CPU as well; nothing here touches the GPU.
//
// Algorithm (Google BERT tokenization.py, which rust_tokenizers and HF BertTokenizer both port):
//   BasicTokenizer: drop NUL / U+FFFD / control chars, whitespace -> ' ', spaces around CJK
//   ideographs, whitespace split, optional lower-casing and accent stripping (NFD, drop Mn),
//   split on punctuation; special tokens present in the text are kept whole.
//   WordPiece: greedy longest-match-first with "##" continuation pieces, words longer than 100
//   chars or without a match -> [UNK].
// Unicode data: unicode_tables.h, generated from Python's unicodedata.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <unordered_map>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "unicode_tables.h"

using namespace pcv;

namespace {

template <class T, size_t N>
bool in_ranges(const T (&tab)[N], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < tab[mid].lo)
            hi = mid;
        else if (cp > tab[mid].hi)
            lo = mid + 1;
        else
            return true;
    }
    return false;
}

uint32_t to_lower(uint32_t cp) {
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    size_t lo = 0, hi = sizeof(uni::kLower) / sizeof(uni::kLower[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < uni::kLower[mid].from)
            hi = mid;
        else if (cp > uni::kLower[mid].from)
            lo = mid + 1;
        else
            return uni::kLower[mid].to;
    }
    return cp;
}

// canonical decomposition (NFD) of one code point, appended to `out`
void decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp >= 0xAC00 && cp <= 0xD7A3) {  // Hangul syllables: algorithmic
        const uint32_t s = cp - 0xAC00;
        out.push_back(0x1100 + s / 588);
        out.push_back(0x1161 + (s % 588) / 28);
        if (s % 28) out.push_back(0x11A7 + s % 28);
        return;
    }
    if (cp >= 0xC0) {
        size_t lo = 0, hi = sizeof(uni::kDecomp) / sizeof(uni::kDecomp[0]);
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (cp < uni::kDecomp[mid].cp)
                hi = mid;
            else if (cp > uni::kDecomp[mid].cp)
                lo = mid + 1;
            else {
                for (int i = 0; i < uni::kDecomp[mid].n; ++i) out.push_back(uni::kDecomp[mid].to[i]);
                return;
            }
        }
    }
    out.push_back(cp);
}

bool is_whitespace(uint32_t cp) {
    return cp == ' ' || cp == '\t' || cp == '\n' || cp == '\r' || in_ranges(uni::kZs, cp);
}
bool is_control(uint32_t cp) {
    if (cp == '\t' || cp == '\n' || cp == '\r') return false;
    return in_ranges(uni::kControl, cp);
}
bool is_punctuation(uint32_t cp) {
    if ((cp >= 33 && cp <= 47) || (cp >= 58 && cp <= 64) || (cp >= 91 && cp <= 96) || (cp >= 123 && cp <= 126)) return true;
    return in_ranges(uni::kPunct, cp);
}
bool is_cjk(uint32_t cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
           (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
           (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}

void append_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) {
        s.push_back((char)cp);
    } else if (cp < 0x800) {
        s.push_back((char)(0xC0 | (cp >> 6)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18)));
        s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// lenient UTF-8 decode: malformed bytes become U+FFFD (and are then dropped by clean_text)
std::vector<uint32_t> decode_utf8(const char* s, size_t n) {
    std::vector<uint32_t> out;
    out.reserve(n);
    size_t i = 0;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp = 0xFFFD;
        int len = 1;
        if (c < 0x80) {
            cp = c;
        } else if ((c & 0xE0) == 0xC0 && i + 1 < n) {
            cp = ((c & 0x1F) << 6) | ((unsigned char)s[i + 1] & 0x3F);
            len = 2;
        } else if ((c & 0xF0) == 0xE0 && i + 2 < n) {
            cp = ((c & 0x0F) << 12) | (((unsigned char)s[i + 1] & 0x3F) << 6) | ((unsigned char)s[i + 2] & 0x3F);
            len = 3;
        } else if ((c & 0xF8) == 0xF0 && i + 3 < n) {
            cp = ((c & 0x07) << 18) | (((unsigned char)s[i + 1] & 0x3F) << 12) | (((unsigned char)s[i + 2] & 0x3F) << 6) |
                 ((unsigned char)s[i + 3] & 0x3F);
            len = 4;
        }
        out.push_back(cp);
        i += len;
    }
    return out;
}

struct Cp {
    uint32_t cp;
    int32_t pos;  // index of the originating char in the input text
};
struct Piece {
    int64_t id;
    int32_t begin, end;
    uint8_t special;
};

}  // namespace

struct pcv_tokenizer {
    std::unordered_map<std::string, int64_t> vocab;
    bool lower_case = true, strip_accents = true;
    int64_t pad = -1, unk = -1, cls = -1, sep = -1, mask = -1;
    std::vector<std::pair<std::vector<uint32_t>, int64_t>> specials;  // kept whole when found in text

    int64_t lookup(const std::string& s) const {
        auto it = vocab.find(s);
        return it == vocab.end() ? -1 : it->second;
    }

    // one whitespace-delimited word -> lower / strip accents -> punctuation split -> wordpiece
    void emit_word(const std::vector<Cp>& word, std::vector<Piece>& out) const {
        std::vector<Cp> w;
        std::vector<uint32_t> tmp;
        for (const Cp& c : word) {
            uint32_t cp = c.cp;
            if (lower_case) {
                if (cp == 0x130) {  // 'İ'.lower() is "i̇" (two code points)
                    w.push_back({'i', c.pos});
                    if (!strip_accents) w.push_back({0x307, c.pos});
                    continue;
                }
                cp = to_lower(cp);
            }
            if (strip_accents) {
                tmp.clear();
                decompose(cp, tmp);
                for (uint32_t d : tmp)
                    if (!in_ranges(uni::kMn, d)) w.push_back({d, c.pos});
            } else {
                w.push_back({cp, c.pos});
            }
        }
        // split on punctuation: every punctuation char is its own token
        size_t i = 0;
        while (i < w.size()) {
            size_t j = i;
            if (is_punctuation(w[i].cp)) {
                j = i + 1;
            } else {
                while (j < w.size() && !is_punctuation(w[j].cp)) ++j;
            }
            wordpiece(w, i, j, out);
            i = j;
        }
    }

    void wordpiece(const std::vector<Cp>& w, size_t a, size_t b, std::vector<Piece>& out) const {
        if (b <= a) return;
        const int32_t wb = w[a].pos, we = w[b - 1].pos + 1;
        if (b - a > 100) {  // max_input_chars_per_word
            out.push_back({unk, wb, we, 0});
            return;
        }
        std::vector<Piece> pieces;
        size_t start = a;
        std::string s;
        while (start < b) {
            size_t end = b;
            int64_t found = -1;
            while (end > start) {
                s.clear();
                if (start > a) s = "##";
                for (size_t k = start; k < end; ++k) append_utf8(s, w[k].cp);
                found = lookup(s);
                if (found >= 0) break;
                --end;
            }
            if (found < 0) {  // no piece matches: the whole word is unknown
                out.push_back({unk, wb, we, 0});
                return;
            }
            pieces.push_back({found, w[start].pos, w[end - 1].pos + 1, 0});
            start = end;
        }
        out.insert(out.end(), pieces.begin(), pieces.end());
    }

    std::vector<Piece> tokenize(const char* text, size_t n) const {
        const std::vector<uint32_t> raw = decode_utf8(text, n);
        std::vector<Piece> out;
        std::vector<Cp> word;
        auto flush = [&] {
            if (!word.empty()) emit_word(word, out);
            word.clear();
        };
        size_t i = 0;
        while (i < raw.size()) {
            // special tokens written out in the text ("[SEP]") stay whole
            bool matched = false;
            if (raw[i] == '[') {
                for (const auto& sp : specials) {
                    const auto& pat = sp.first;
                    if (i + pat.size() <= raw.size() && std::equal(pat.begin(), pat.end(), raw.begin() + i)) {
                        flush();
                        out.push_back({sp.second, (int32_t)i, (int32_t)(i + pat.size()), 0});
                        i += pat.size();
                        matched = true;
                        break;
                    }
                }
            }
            if (matched) continue;
            const uint32_t cp = raw[i];
            if (cp == 0 || cp == 0xFFFD || is_control(cp)) {
                // dropped by clean_text
            } else if (is_whitespace(cp)) {
                flush();
            } else if (is_cjk(cp)) {  // every CJK ideograph is its own word
                flush();
                word.push_back({cp, (int32_t)i});
                flush();
            } else {
                word.push_back({cp, (int32_t)i});
            }
            ++i;
        }
        flush();
        return out;
    }
};

extern "C" {

pcv_status pcv_tokenizer_create(const char* vocab_path, int lower_case, int strip_accents, pcv_tokenizer** out) {
    return guarded([&] {
        PCV_REQUIRE(vocab_path != nullptr && out != nullptr, "tokenizer_create: NULL argument");
        *out = nullptr;
        std::ifstream f(vocab_path, std::ios::binary);
        if (!f) PCV_FAIL(PCV_ERR_IO, "tokenizer_create: cannot open vocab file %s", vocab_path);
        auto t = std::make_unique<pcv_tokenizer>();
        std::string line;
        int64_t id = 0;
        while (std::getline(f, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            t->vocab.emplace(line, id);  // first occurrence wins, ids are line numbers
            ++id;
        }
        PCV_REQUIRE(id > 0, "tokenizer_create: vocab file %s is empty", vocab_path);
        t->lower_case = lower_case != 0;
        t->strip_accents = strip_accents < 0 ? t->lower_case : strip_accents != 0;
        t->pad = t->lookup("[PAD]");
        t->unk = t->lookup("[UNK]");
        t->cls = t->lookup("[CLS]");
        t->sep = t->lookup("[SEP]");
        t->mask = t->lookup("[MASK]");
        if (t->unk < 0 || t->cls < 0 || t->sep < 0)
            PCV_FAIL(PCV_ERR_IO, "tokenizer_create: vocab %s lacks [UNK]/[CLS]/[SEP]", vocab_path);
        for (const char* sp : {"[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]"}) {
            const int64_t sid = t->lookup(sp);
            if (sid >= 0) t->specials.push_back({decode_utf8(sp, std::strlen(sp)), sid});
        }
        *out = t.release();
    });
}

pcv_status pcv_tokenizer_destroy(pcv_tokenizer* t) {
    return guarded([&] { delete t; });
}

pcv_status pcv_tokenizer_vocab_size(pcv_tokenizer* t, int* out_n) {
    return guarded([&] {
        PCV_REQUIRE(t != nullptr && out_n != nullptr, "tokenizer_vocab_size: NULL argument");
        *out_n = (int)t->vocab.size();
    });
}

pcv_status pcv_tokenizer_special_ids(pcv_tokenizer* t, int64_t* pad, int64_t* unk, int64_t* cls, int64_t* sep) {
    return guarded([&] {
        PCV_REQUIRE(t != nullptr, "tokenizer_special_ids: tokenizer is NULL");
        if (pad) *pad = t->pad;
        if (unk) *unk = t->unk;
        if (cls) *cls = t->cls;
        if (sep) *sep = t->sep;
    });
}

pcv_status pcv_tokenizer_encode(pcv_tokenizer* t, const char* text, size_t n_bytes, int max_len, int64_t* out_ids,
                                int32_t* out_begin, int32_t* out_end, uint8_t* out_special, int cap, int* out_len) {
    return guarded([&] {
        PCV_REQUIRE(t != nullptr && (text != nullptr || n_bytes == 0) && out_ids != nullptr && out_len != nullptr,
                    "tokenizer_encode: NULL argument");
        PCV_REQUIRE(max_len >= 2, "tokenizer_encode: max_len %d leaves no room for [CLS] [SEP]", max_len);
        std::vector<Piece> pieces = t->tokenize(text, n_bytes);
        // truncate_sequences(LongestFirst, stride 0) of a single sequence: drop from the end
        const size_t room = (size_t)max_len - 2;
        if (pieces.size() > room) pieces.resize(room);
        const int total = (int)pieces.size() + 2;
        *out_len = total;
        PCV_REQUIRE(cap >= total, "tokenizer_encode: output capacity %d < %d tokens", cap, total);
        auto put = [&](int i, const Piece& p) {
            out_ids[i] = p.id;
            if (out_begin) out_begin[i] = p.begin;
            if (out_end) out_end[i] = p.end;
            if (out_special) out_special[i] = p.special;
        };
        put(0, Piece{t->cls, -1, -1, 1});
        for (size_t i = 0; i < pieces.size(); ++i) put((int)i + 1, pieces[i]);
        put(total - 1, Piece{t->sep, -1, -1, 1});
    });
}

// Model::tokenize for a whole batch (tokenize.rs:60-77 + generate_token_tensors, tokenize.rs:9-57): every text
// is encoded like pcv_tokenizer_encode, written right-padded with `pad_id` into row i of out_ids
// ([n_texts][max_len]); out_lens[i] = its token count.  The texts are independent, so they are spread over
// `n_threads` host threads (0 = hardware concurrency, capped by n_texts): the reference tokenizes on the CPU
// too, and at GPU encode rates a single thread (~70 us per 256-token document) would be the bottleneck.
pcv_status pcv_tokenizer_encode_batch(pcv_tokenizer* t, const char* const* texts, const size_t* n_bytes, int n_texts,
                                      int max_len, int64_t pad_id, int64_t* out_ids, int32_t* out_lens, int n_threads) {
    return guarded([&] {
        PCV_REQUIRE(t != nullptr && (n_texts == 0 || (texts != nullptr && n_bytes != nullptr && out_ids != nullptr && out_lens != nullptr)),
                    "tokenizer_encode_batch: NULL argument");
        PCV_REQUIRE(max_len >= 2, "tokenizer_encode_batch: max_len %d leaves no room for [CLS] [SEP]", max_len);
        PCV_REQUIRE(n_texts >= 0, "tokenizer_encode_batch: negative count");
        auto one = [&](int i) {
            std::vector<Piece> pieces = t->tokenize(texts[i], n_bytes[i]);
            const size_t room = (size_t)max_len - 2;
            if (pieces.size() > room) pieces.resize(room);
            int64_t* row = out_ids + (size_t)i * max_len;
            row[0] = t->cls;
            for (size_t j = 0; j < pieces.size(); ++j) row[j + 1] = pieces[j].id;
            row[pieces.size() + 1] = t->sep;
            for (size_t j = pieces.size() + 2; j < (size_t)max_len; ++j) row[j] = pad_id;
            out_lens[i] = (int32_t)pieces.size() + 2;
        };
        int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
        nt = std::max(1, std::min(nt, n_texts));
        if (nt == 1) {
            for (int i = 0; i < n_texts; ++i) one(i);
            return;
        }
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        std::exception_ptr err;
        std::mutex err_mu;
        for (int w = 0; w < nt; ++w)
            pool.emplace_back([&] {
                try {
                    for (int i = next.fetch_add(1); i < n_texts; i = next.fetch_add(1)) one(i);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!err) err = std::current_exception();
                }
            });
        for (auto& th : pool) th.join();
        if (err) std::rethrow_exception(err);
    });
}

}  // extern "C"

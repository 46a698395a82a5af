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
#include <climits>
#include <iterator>
#include <vector>

#include "common.h"
#include "tokenizer.h"
#include "unicode_nfkc_tables.h"
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

// ---- NFKC (UAX #15): compatibility decomposition, canonical ordering, canonical composition ----------------
uint32_t combining_class(uint32_t cp) {
    if (cp < 0x300) return 0;
    size_t lo = 0, hi = sizeof(uni::kCcc) / sizeof(uni::kCcc[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < uni::kCcc[mid].lo)
            hi = mid;
        else if (cp > uni::kCcc[mid].hi)
            lo = mid + 1;
        else
            return uni::kCcc[mid].ccc;
    }
    return 0;
}

uint32_t compose_pair(uint32_t a, uint32_t b) {  // 0 = no primary composite
    if (a >= 0x1100 && a < 0x1113 && b >= 0x1161 && b < 0x1176) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;  // L + V
    if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b > 0x11A7 && b < 0x11C3) return a + (b - 0x11A7);        // LV + T
    size_t lo = 0, hi = sizeof(uni::kCompose) / sizeof(uni::kCompose[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const uni::Compose& c = uni::kCompose[mid];
        if (a < c.a || (a == c.a && b < c.b))
            hi = mid;
        else if (a > c.a || (a == c.a && b > c.b))
            lo = mid + 1;
        else
            return c.c;
    }
    return 0;
}

struct Cp {
    uint32_t cp;
    int32_t pos;  // index of the originating char in the input text
};

void decompose_compat(const Cp& c, std::vector<Cp>& out) {
    const uint32_t cp = c.cp;
    if (cp >= 0xAC00 && cp <= 0xD7A3) {
        const uint32_t s = cp - 0xAC00;
        out.push_back({0x1100 + s / 588, c.pos});
        out.push_back({0x1161 + (s % 588) / 28, c.pos});
        if (s % 28) out.push_back({0x11A7 + s % 28, c.pos});
        return;
    }
    if (cp >= 0xA0) {
        size_t lo = 0, hi = sizeof(uni::kCompatIdx) / sizeof(uni::kCompatIdx[0]);
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (cp < uni::kCompatIdx[mid].cp)
                hi = mid;
            else if (cp > uni::kCompatIdx[mid].cp)
                lo = mid + 1;
            else {
                for (uint32_t i = 0; i < uni::kCompatIdx[mid].n; ++i) out.push_back({uni::kCompatPool[uni::kCompatIdx[mid].off + i], c.pos});
                return;
            }
        }
    }
    out.push_back(c);
}

// NFKC of a code-point sequence; every output code point keeps the input position it came from (a composite:
// the position of its starter)
std::vector<Cp> nfkc(const std::vector<Cp>& in) {
    bool plain = true;
    for (const Cp& c : in)
        if (c.cp >= 0xA0) {
            plain = false;
            break;
        }
    if (plain) return in;
    std::vector<Cp> d;
    d.reserve(in.size() + 8);
    for (const Cp& c : in) decompose_compat(c, d);
    // canonical ordering: stable sort of every run of non-starters by combining class
    for (size_t i = 0; i < d.size();) {
        if (combining_class(d[i].cp) == 0) {
            ++i;
            continue;
        }
        size_t j = i;
        while (j < d.size() && combining_class(d[j].cp) != 0) ++j;
        std::stable_sort(d.begin() + (long)i, d.begin() + (long)j,
                         [](const Cp& x, const Cp& y) { return combining_class(x.cp) < combining_class(y.cp); });
        i = j;
    }
    if (d.empty()) return d;
    // canonical composition (the reference algorithm of UAX #15)
    size_t starter = 0, comp = 1;
    uint32_t starter_cp = d[0].cp;
    int last_class = (int)combining_class(starter_cp);
    if (last_class != 0) last_class = 256;  // a leading non-starter never composes
    for (size_t i = 1; i < d.size(); ++i) {
        const Cp ch = d[i];
        const int cls = (int)combining_class(ch.cp);
        const uint32_t composite = compose_pair(starter_cp, ch.cp);
        if (composite != 0 && (last_class < cls || last_class == 0)) {
            d[starter].cp = composite;
            starter_cp = composite;
        } else {
            if (cls == 0) {
                starter = comp;
                starter_cp = ch.cp;
            }
            last_class = cls;
            d[comp++] = ch;
        }
    }
    d.resize(comp);
    return d;
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

using Piece = pcv::TokPiece;

}  // namespace

struct pcv_tokenizer {
    std::unordered_map<std::string, int64_t> vocab;
    bool lower_case = true, strip_accents = true;
    // byte-level BPE (RoBERTa family): merge ranks keyed by "left\x01right", the 256 byte symbols of GPT-2's
    // bytes_to_unicode, and a cache of finished words (shared by the batch threads)
    bool bpe = false, add_prefix_space = false;
    std::unordered_map<std::string, int> merges;
    std::string byte_sym[256];
    struct BpeTok {
        int64_t id;
        int32_t b0, b1;  // byte range inside the word
    };
    mutable std::unordered_map<std::string, std::vector<BpeTok>> bpe_cache;
    mutable std::mutex bpe_mu;
    int64_t pad = -1, unk = -1, cls = -1, sep = -1, mask = -1;
    std::vector<std::pair<std::vector<uint32_t>, int64_t>> specials;  // kept whole when found in text
    // SentencePiece unigram model (ALBERT): a trie over the code points of the NORMAL / USER_DEFINED pieces
    bool spm = false;
    struct TrieNode {
        std::vector<std::pair<uint32_t, int32_t>> next;  // sorted by code point
        int64_t id = -1;
        float score = 0.f;
    };
    std::vector<TrieNode> trie;

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

    // ---- byte-level BPE -------------------------------------------------------------------------------
    // GPT-2's pre-tokenizer pattern, by hand (no regex engine):
    //   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
    static bool bpe_space(uint32_t c) {  // \s: Unicode White_Space
        return (c >= 9 && c <= 13) || c == 32 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
               c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
    }
    static int bpe_class(uint32_t c) {  // 0 letter, 1 number, 2 other, 3 whitespace
        if (bpe_space(c)) return 3;
        if (in_ranges(uni::kLetter, c)) return 0;
        if (in_ranges(uni::kNumber, c)) return 1;
        return 2;
    }
    static size_t bpe_next_pretoken(const std::vector<uint32_t>& cp, size_t i) {
        const size_t n = cp.size();
        if (cp[i] == '\'' && i + 1 < n) {
            const uint32_t a = cp[i + 1], b = i + 2 < n ? cp[i + 2] : 0;
            if (a == 's' || a == 't' || a == 'm' || a == 'd') return i + 2;
            if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) return i + 3;
        }
        size_t j = i;
        if (cp[j] == ' ' && j + 1 < n && bpe_class(cp[j + 1]) != 3) ++j;  // " ?" in front of a non-space run
        const int cls = bpe_class(cp[j]);
        if (cls != 3) {
            size_t k = j + 1;
            while (k < n && bpe_class(cp[k]) == cls) ++k;
            return k;
        }
        size_t k = i;  // whitespace run: all of it at the end of the text, else all but its last char (>= 1)
        while (k < n && bpe_space(cp[k])) ++k;
        if (k == n || k - i == 1) return k;
        return k - 1;
    }

    // merges of one pre-token (bytes w[0..len)): ids + byte ranges
    std::vector<BpeTok> bpe_word(const unsigned char* w, size_t len) const {
        const std::string key((const char*)w, len);
        {
            std::lock_guard<std::mutex> lk(bpe_mu);
            auto it = bpe_cache.find(key);
            if (it != bpe_cache.end()) return it->second;
        }
        struct Sym {
            std::string s;
            int32_t b0, b1;
        };
        std::vector<Sym> syms;
        syms.reserve(len);
        for (size_t i = 0; i < len; ++i) syms.push_back({byte_sym[w[i]], (int32_t)i, (int32_t)i + 1});
        while (syms.size() > 1) {
            int best = INT32_MAX;
            std::string bl, br;
            for (size_t i = 0; i + 1 < syms.size(); ++i) {
                auto it = merges.find(syms[i].s + '\x01' + syms[i + 1].s);
                if (it != merges.end() && it->second < best) {
                    best = it->second;
                    bl = syms[i].s;
                    br = syms[i + 1].s;
                }
            }
            if (best == INT32_MAX) break;
            std::vector<Sym> next;
            next.reserve(syms.size());
            for (size_t i = 0; i < syms.size();) {
                if (i + 1 < syms.size() && syms[i].s == bl && syms[i + 1].s == br) {
                    next.push_back({bl + br, syms[i].b0, syms[i + 1].b1});
                    i += 2;
                } else {
                    next.push_back(syms[i]);
                    ++i;
                }
            }
            syms.swap(next);
        }
        std::vector<BpeTok> out;
        out.reserve(syms.size());
        for (const Sym& y : syms) {
            const int64_t id = lookup(y.s);
            out.push_back({id >= 0 ? id : unk, y.b0, y.b1});
        }
        if (len <= 64) {
            std::lock_guard<std::mutex> lk(bpe_mu);
            if (bpe_cache.size() < (1u << 20)) bpe_cache.emplace(key, out);
        }
        return out;
    }

    std::vector<Piece> tokenize_bpe(const char* text, size_t n) const {
        std::string buf;
        const bool prefixed = add_prefix_space && n > 0 && text[0] != ' ';
        if (prefixed) {
            buf.reserve(n + 1);
            buf.push_back(' ');
            buf.append(text, n);
            text = buf.data();
            n = buf.size();
        }
        const std::vector<uint32_t> cp = decode_utf8(text, n);
        std::vector<uint32_t> boff(cp.size() + 1);  // byte offset of every char
        {
            size_t b = 0, i = 0, ci = 0;
            while (i < n) {  // same walk as decode_utf8
                const unsigned char c = (unsigned char)text[i];
                int len = 1;
                if (c >= 0x80) {
                    if ((c & 0xE0) == 0xC0 && i + 1 < n) len = 2;
                    else if ((c & 0xF0) == 0xE0 && i + 2 < n) len = 3;
                    else if ((c & 0xF8) == 0xF0 && i + 3 < n) len = 4;
                }
                boff[ci++] = (uint32_t)b;
                b += len;
                i += len;
            }
            boff[ci] = (uint32_t)b;
        }
        std::vector<int32_t> char_of(n + 1);  // char index of every byte
        for (size_t ci = 0; ci < cp.size(); ++ci)
            for (uint32_t b = boff[ci]; b < boff[ci + 1]; ++b) char_of[b] = (int32_t)ci;
        std::vector<Piece> out;
        size_t i = 0;
        while (i < cp.size()) {
            const size_t k = bpe_next_pretoken(cp, i);
            const uint32_t b0 = boff[i], b1 = boff[k];
            for (const BpeTok& tk : bpe_word((const unsigned char*)text + b0, b1 - b0)) {
                int32_t cb = char_of[b0 + tk.b0], ce = char_of[b0 + tk.b1 - 1] + 1;
                if (prefixed) {  // offsets are reported against the caller's text (the tokenizers library's convention)
                    cb = cb > 0 ? cb - 1 : 0;
                    ce = ce > 1 ? ce - 1 : ce;
                }
                out.push_back({tk.id, cb, ce, 0});
            }
            i = k;
        }
        return out;
    }

    // ---- SentencePiece unigram (AlbertTokenizer of rust_tokenizers, what rust-bert builds for ModelType::Albert) ----
    void trie_insert(const std::vector<uint32_t>& cps, int64_t id, float score) {
        int32_t node = 0;
        for (uint32_t c : cps) {
            auto& nx = trie[(size_t)node].next;
            auto it = std::lower_bound(nx.begin(), nx.end(), std::make_pair(c, (int32_t)INT32_MIN));
            if (it != nx.end() && it->first == c) {
                node = it->second;
            } else {
                const int32_t fresh = (int32_t)trie.size();
                nx.insert(it, {c, fresh});
                trie.emplace_back();
                node = fresh;
            }
        }
        if (trie[(size_t)node].id < 0) {  // first occurrence wins
            trie[(size_t)node].id = id;
            trie[(size_t)node].score = score;
        }
    }

    // id of the piece spelled w[a..b), or <unk>
    int64_t trie_lookup(const std::vector<Cp>& w, size_t a, size_t b) const {
        int32_t node = 0;
        for (size_t j = a; j < b; ++j) {
            const auto& nx = trie[(size_t)node].next;
            auto it = std::lower_bound(nx.begin(), nx.end(), std::make_pair(w[j].cp, (int32_t)INT32_MIN));
            if (it == nx.end() || it->first != w[j].cp) return unk;
            node = it->second;
        }
        return trie[(size_t)node].id >= 0 ? trie[(size_t)node].id : unk;
    }

    // character span of w[a..b) in the caller's text; the inserted leading "▁" (pos -1) covers nothing
    static Piece piece_over(const std::vector<Cp>& w, size_t a, size_t b, int64_t id) {
        int32_t lo = INT32_MAX, hi = -1;
        for (size_t k = a; k < b; ++k)
            if (w[k].pos >= 0) {
                lo = std::min(lo, w[k].pos);
                hi = std::max(hi, w[k].pos + 1);
            }
        if (hi < 0) lo = hi = (b < w.size() && w[b].pos >= 0) ? w[b].pos : 0;
        return Piece{id, lo, hi, 0};
    }

    struct Span {
        size_t a, b;
        int64_t id;
    };

    // Viterbi over the code points of w: the best-scoring segmentation into pieces (f32 sums, a later candidate
    // replaces an earlier one only when strictly better); a char no piece covers becomes <unk> on its own and the
    // running score restarts behind it (rust_tokenizers' SentencePieceModel::decode_forward)
    std::vector<Span> unigram(const std::vector<Cp>& w) const {
        const size_t n = w.size();
        struct Best {
            float score;
            int32_t from;
            int64_t id;
        };
        std::vector<Best> best(n + 1, Best{-INFINITY, -1, -1});
        best[0].score = 0.f;
        for (size_t i = 0; i < n; ++i) {
            int32_t node = 0;
            for (size_t j = i; j < n; ++j) {
                const auto& nx = trie[(size_t)node].next;
                auto it = std::lower_bound(nx.begin(), nx.end(), std::make_pair(w[j].cp, (int32_t)INT32_MIN));
                if (it == nx.end() || it->first != w[j].cp) break;
                node = it->second;
                const TrieNode& tn = trie[(size_t)node];
                if (tn.id >= 0) {
                    const float sc = best[i].score + tn.score;
                    if (sc > best[j + 1].score) best[j + 1] = Best{sc, (int32_t)i, tn.id};
                }
            }
            if (best[i + 1].from < 0) best[i + 1] = Best{0.f, (int32_t)i, unk};
        }
        std::vector<Span> out;
        for (size_t e = n; e > 0; e = (size_t)best[e].from) out.push_back(Span{(size_t)best[e].from, e, best[e].id});
        std::reverse(out.begin(), out.end());
        return out;
    }

    std::vector<Piece> tokenize_spm(const char* text, size_t n) const {
        const std::vector<uint32_t> raw = decode_utf8(text, n);
        std::vector<Cp> w;
        w.reserve(raw.size() + 1);
        for (size_t i = 0; i < raw.size(); ++i) {  // clean_text: control characters go, whitespace becomes ' '
            const uint32_t cp = raw[i];
            if (cp == 0 || cp == 0xFFFD || is_control(cp)) continue;
            w.push_back({is_whitespace(cp) ? (uint32_t)' ' : cp, (int32_t)i});
        }
        w = nfkc(w);
        if (lower_case || strip_accents) {
            std::vector<Cp> v;
            v.reserve(w.size());
            std::vector<uint32_t> tmp;
            for (const Cp& c : w) {
                uint32_t cp = c.cp;
                if (lower_case) {
                    if (cp == 0x130) {  // the full lowercase mapping of U+0130 is two code points
                        v.push_back({'i', c.pos});
                        if (!strip_accents) v.push_back({0x307, c.pos});
                        continue;
                    }
                    cp = to_lower(cp);
                }
                if (strip_accents) {
                    tmp.clear();
                    decompose(cp, tmp);
                    for (uint32_t d : tmp)
                        if (!in_ranges(uni::kMn, d)) v.push_back({d, c.pos});
                } else {
                    v.push_back({cp, c.pos});
                }
            }
            w.swap(v);
        }
        for (Cp& c : w)
            if (is_whitespace(c.cp)) c.cp = 0x2581;
        if (w.empty() || w[0].cp != 0x2581) w.insert(w.begin(), Cp{0x2581, -1});
        std::vector<Piece> out;
        for (const Span& sp : unigram(w)) {
            const size_t len = sp.b - sp.a;
            // ALBERT's digit rule: a piece "<...digit>," is segmented again without the comma ("▁2000," -> "▁2000" ",")
            if (len > 1 && w[sp.b - 1].cp == ',' && w[sp.b - 2].cp >= '0' && w[sp.b - 2].cp <= '9') {
                const bool had_prefix = w[sp.a].cp == 0x2581;
                std::vector<Cp> sub;
                sub.push_back(Cp{0x2581, had_prefix ? w[sp.a].pos : -1});  // the dummy prefix of a fresh encode
                for (size_t k = sp.a; k + 1 < sp.b; ++k)
                    if (w[k].cp != 0x2581) sub.push_back(w[k]);
                std::vector<Span> again = unigram(sub);
                for (size_t q = 0; q < again.size(); ++q) {
                    Span s2 = again[q];
                    if (q == 0 && !had_prefix && sub[s2.a].cp == 0x2581) {  // the prefix was only there for the re-run
                        if (s2.b - s2.a == 1) continue;
                        s2.a += 1;
                        s2.id = trie_lookup(sub, s2.a, s2.b);
                    }
                    out.push_back(piece_over(sub, s2.a, s2.b, s2.id));
                }
                out.push_back(piece_over(w, sp.b - 1, sp.b, trie_lookup(w, sp.b - 1, sp.b)));
            } else {
                out.push_back(piece_over(w, sp.a, sp.b, sp.id));
            }
        }
        return out;
    }

    std::vector<Piece> tokenize(const char* text, size_t n) const {
        if (spm) return tokenize_spm(text, n);
        if (bpe) return tokenize_bpe(text, n);
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

namespace {
// {"token": id, ...} — the only JSON the byte-level BPE vocabulary needs
void parse_vocab_json(const std::string& j, std::unordered_map<std::string, int64_t>& vocab) {
    size_t i = 0;
    auto ws = [&] { while (i < j.size() && (j[i] == ' ' || j[i] == '\n' || j[i] == '\r' || j[i] == '\t')) ++i; };
    auto hex4 = [&](size_t at) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = at + k < j.size() ? j[at + k] : '0';
            v = v * 16 + (uint32_t)(c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : 0);
        }
        return v;
    };
    ws();
    PCV_REQUIRE(i < j.size() && j[i] == '{', "vocab.json: expected an object");
    ++i;
    while (true) {
        ws();
        if (i < j.size() && j[i] == '}') break;
        PCV_REQUIRE(i < j.size() && j[i] == '"', "vocab.json: expected a string key at byte %zu", i);
        ++i;
        std::string key;
        while (i < j.size() && j[i] != '"') {
            if (j[i] == '\\' && i + 1 < j.size()) {
                const char e = j[i + 1];
                i += 2;
                switch (e) {
                    case 'n': key.push_back('\n'); break;
                    case 't': key.push_back('\t'); break;
                    case 'r': key.push_back('\r'); break;
                    case 'b': key.push_back('\b'); break;
                    case 'f': key.push_back('\f'); break;
                    case 'u': {
                        uint32_t cp = hex4(i);
                        i += 4;
                        if (cp >= 0xD800 && cp <= 0xDBFF && i + 5 < j.size() && j[i] == '\\' && j[i + 1] == 'u') {
                            const uint32_t lo = hex4(i + 2);
                            if (lo >= 0xDC00 && lo <= 0xDFFF) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                                i += 6;
                            }
                        }
                        append_utf8(key, cp);
                        break;
                    }
                    default: key.push_back(e); break;  // \" \\ \/
                }
            } else {
                key.push_back(j[i++]);
            }
        }
        PCV_REQUIRE(i < j.size(), "vocab.json: unterminated string");
        ++i;
        ws();
        PCV_REQUIRE(i < j.size() && j[i] == ':', "vocab.json: expected ':' at byte %zu", i);
        ++i;
        ws();
        int64_t v = 0;
        bool any = false;
        while (i < j.size() && j[i] >= '0' && j[i] <= '9') {
            v = v * 10 + (j[i++] - '0');
            any = true;
        }
        PCV_REQUIRE(any, "vocab.json: expected an integer id for \"%s\"", key.c_str());
        vocab.emplace(key, v);
        ws();
        if (i < j.size() && j[i] == ',') ++i;
    }
}
}  // namespace

// RobertaTokenizer::from_file(vocab.json, merges.txt, .., add_prefix_space) of rust_tokenizers (what rust-bert
// builds for ModelType::Roberta): byte-level BPE with <s> ... </s> framing.
pcv_status pcv_tokenizer_create_bpe(const char* vocab_json_path, const char* merges_path, int add_prefix_space,
                                    pcv_tokenizer** out) {
    return guarded([&] {
        PCV_REQUIRE(vocab_json_path != nullptr && merges_path != nullptr && out != nullptr, "tokenizer_create_bpe: NULL argument");
        *out = nullptr;
        std::ifstream vf(vocab_json_path, std::ios::binary);
        if (!vf) PCV_FAIL(PCV_ERR_IO, "tokenizer_create_bpe: cannot open vocab file %s", vocab_json_path);
        const std::string json((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
        auto t = std::make_unique<pcv_tokenizer>();
        t->bpe = true;
        t->add_prefix_space = add_prefix_space != 0;
        parse_vocab_json(json, t->vocab);
        PCV_REQUIRE(!t->vocab.empty(), "tokenizer_create_bpe: vocab file %s is empty", vocab_json_path);
        std::ifstream mf(merges_path, std::ios::binary);
        if (!mf) PCV_FAIL(PCV_ERR_IO, "tokenizer_create_bpe: cannot open merges file %s", merges_path);
        std::string line;
        int rank = 0;
        while (std::getline(mf, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            if (line.empty() || line.rfind("#version", 0) == 0) continue;
            const size_t sp = line.find(' ');
            if (sp == std::string::npos) continue;
            t->merges.emplace(line.substr(0, sp) + '\x01' + line.substr(sp + 1), rank++);
        }
        {  // bytes_to_unicode of GPT-2: printable bytes map to themselves, the rest to U+0100...
            int extra = 0;
            for (int b = 0; b < 256; ++b) {
                const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
                append_utf8(t->byte_sym[b], keep ? (uint32_t)b : (uint32_t)(256 + extra++));
            }
        }
        t->pad = t->lookup("<pad>");
        t->unk = t->lookup("<unk>");
        t->cls = t->lookup("<s>");
        t->sep = t->lookup("</s>");
        t->mask = t->lookup("<mask>");
        if (t->unk < 0 || t->cls < 0 || t->sep < 0)
            PCV_FAIL(PCV_ERR_IO, "tokenizer_create_bpe: vocab %s lacks <unk>/<s>/</s>", vocab_json_path);
        *out = t.release();
    });
}

namespace {
// protobuf wire format, as much as sentencepiece_model.proto needs:
//   ModelProto { repeated SentencePiece pieces = 1; ... }   SentencePiece { string piece = 1; float score = 2; Type type = 3; }
struct Wire {
    const uint8_t* p;
    size_t n, i = 0;
    const char* what;
    bool more() const { return i < n; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (i >= n) PCV_FAIL(PCV_ERR_IO, "%s: truncated varint", what);
            const uint8_t b = p[i++];
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        PCV_FAIL(PCV_ERR_IO, "%s: varint too long", what);
    }
    Wire bytes() {
        const uint64_t len = varint();
        if (len > n - i) PCV_FAIL(PCV_ERR_IO, "%s: field runs past the end", what);
        Wire w{p + i, (size_t)len, 0, what};
        i += (size_t)len;
        return w;
    }
    void skip(int wire_type) {
        switch (wire_type) {
            case 0: (void)varint(); break;
            case 1: if (n - i < 8) PCV_FAIL(PCV_ERR_IO, "%s: truncated", what); i += 8; break;
            case 2: (void)bytes(); break;
            case 5: if (n - i < 4) PCV_FAIL(PCV_ERR_IO, "%s: truncated", what); i += 4; break;
            default: PCV_FAIL(PCV_ERR_IO, "%s: wire type %d is not supported", what, wire_type);
        }
    }
};
}  // namespace

// AlbertTokenizer::from_file(spiece.model, lower_case, strip_accents) of rust_tokenizers (what rust-bert builds for
// ModelType::Albert): the unigram pieces and scores of a SentencePiece model file.  The model's own normaliser
// spec (precompiled character map) is not applied: like rust_tokenizers the text is cleaned, NFKC-normalised,
// lower-cased, stripped of accents and its whitespace turned into "▁" before the Viterbi segmentation.
pcv_status pcv_tokenizer_create_sentencepiece(const char* model_path, int lower_case, int strip_accents, pcv_tokenizer** out) {
    return guarded([&] {
        PCV_REQUIRE(model_path != nullptr && out != nullptr, "tokenizer_create_sentencepiece: NULL argument");
        *out = nullptr;
        std::ifstream f(model_path, std::ios::binary);
        if (!f) PCV_FAIL(PCV_ERR_IO, "tokenizer_create_sentencepiece: cannot open %s", model_path);
        const std::string blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto t = std::make_unique<pcv_tokenizer>();
        t->spm = true;
        t->lower_case = lower_case != 0;
        t->strip_accents = strip_accents < 0 ? t->lower_case : strip_accents != 0;
        t->trie.emplace_back();
        Wire top{(const uint8_t*)blob.data(), blob.size(), 0, model_path};
        int64_t id = 0;
        while (top.more()) {
            const uint64_t tag = top.varint();
            if ((tag >> 3) != 1 || (tag & 7) != 2) {  // trainer / normaliser specs ...
                top.skip((int)(tag & 7));
                continue;
            }
            Wire pc = top.bytes();
            std::string piece;
            float score = 0.f;
            int type = 1;  // NORMAL
            while (pc.more()) {
                const uint64_t ptag = pc.varint();
                if ((ptag >> 3) == 1 && (ptag & 7) == 2) {
                    Wire sv = pc.bytes();
                    piece.assign((const char*)sv.p, sv.n);
                } else if ((ptag >> 3) == 2 && (ptag & 7) == 5) {
                    if (pc.n - pc.i < 4) PCV_FAIL(PCV_ERR_IO, "%s: truncated score", model_path);
                    std::memcpy(&score, pc.p + pc.i, 4);  // little-endian float
                    pc.i += 4;
                } else if ((ptag >> 3) == 3 && (ptag & 7) == 0) {
                    type = (int)pc.varint();
                } else {
                    pc.skip((int)(ptag & 7));
                }
            }
            t->vocab.emplace(piece, id);
            // NORMAL = 1 and USER_DEFINED = 4 pieces can be matched in text; <unk>, control and unused pieces cannot
            if ((type == 1 || type == 4) && !piece.empty()) t->trie_insert(decode_utf8(piece.data(), piece.size()), id, score);
            ++id;
        }
        PCV_REQUIRE(id > 0, "tokenizer_create_sentencepiece: %s holds no pieces (not a SentencePiece model?)", model_path);
        t->pad = t->lookup("<pad>");
        t->unk = t->lookup("<unk>");
        t->cls = t->lookup("[CLS]");
        t->sep = t->lookup("[SEP]");
        t->mask = t->lookup("[MASK]");
        if (t->unk < 0 || t->cls < 0 || t->sep < 0)
            PCV_FAIL(PCV_ERR_IO, "tokenizer_create_sentencepiece: %s lacks <unk>/[CLS]/[SEP]", model_path);
        *out = t.release();
    });
}

// Unicode NFKC of a UTF-8 string (the normal form the SentencePiece path applies); out_n = bytes needed.
pcv_status pcv_unicode_nfkc(const char* text, size_t n_bytes, char* out, size_t cap, size_t* out_n) {
    return guarded([&] {
        PCV_REQUIRE((text != nullptr || n_bytes == 0) && out_n != nullptr, "unicode_nfkc: NULL argument");
        const std::vector<uint32_t> raw = decode_utf8(text, n_bytes);
        std::vector<Cp> w(raw.size());
        for (size_t i = 0; i < raw.size(); ++i) w[i] = Cp{raw[i], (int32_t)i};
        std::string res;
        for (const Cp& c : nfkc(w)) append_utf8(res, c.cp);
        *out_n = res.size();
        if (out != nullptr && cap >= res.size()) std::memcpy(out, res.data(), res.size());
        else if (out != nullptr) PCV_FAIL(PCV_ERR_INVALID, "unicode_nfkc: %zu bytes needed, room for %zu", res.size(), cap);
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

namespace pcv {
std::vector<TokPiece> tokenizer_pieces(const pcv_tokenizer* t, const char* text, size_t n_bytes) { return t->tokenize(text, n_bytes); }
TokSpecials tokenizer_specials(const pcv_tokenizer* t) { return TokSpecials{t->pad, t->unk, t->cls, t->sep}; }
}  // namespace pcv

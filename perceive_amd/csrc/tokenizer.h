// tokenizer.h — the tokenizer as the text side of the Model uses it from C++ (text_model.cpp).  Not part of
// the C ABI (that is pcv_tokenizer_* in include/perceive_hip.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

struct pcv_tokenizer;

namespace pcv {

struct TokPiece {
    int64_t id;
    int32_t begin, end;  // char (Unicode scalar) offsets of the token in the text
    uint8_t special;
};

// The tokens of `text` without [CLS] / [SEP] framing and without truncation.
std::vector<TokPiece> tokenizer_pieces(const pcv_tokenizer* t, const char* text, size_t n_bytes);

struct TokSpecials {
    int64_t pad, unk, cls, sep;  // -1 when the vocabulary lacks one
};
TokSpecials tokenizer_specials(const pcv_tokenizer* t);

}  // namespace pcv

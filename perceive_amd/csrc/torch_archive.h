// Reader for libtorch's zip archives: the `rust_model.ot` files the reference loads its weights from
// (model.rs:117-124 VarStore::load; written by tch's Tensor::save_multi = torch::serialize::OutputArchive,
// scripts/install_models.sh:36) and `pytorch_model.bin` state dicts (torch.save).
//
// Both are zip files of STORED entries: `<root>/data.pkl` (a pickle program naming the tensors) and
// `<root>/data/<key>` (raw little-endian storages).  The pickle is interpreted by a closed machine that
// knows the handful of constructors those writers emit (tensor rebuild, OrderedDict, the module object);
// any other global ends the read with an error — nothing from the file is ever executed.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace pcv {

enum ArchiveDtype { AR_F32 = 0, AR_F16 = 1, AR_BF16 = 2, AR_F64 = 3, AR_OTHER = 4 };

struct ArchiveTensor {
    std::string name;             // the key the writer used, e.g. "bert.embeddings.word_embeddings.weight"
    int dtype = AR_OTHER;         // AR_OTHER: integer / bool tensors (position_ids ...), data not converted
    std::string dtype_name;       // storage class as written ("FloatStorage", ...)
    std::vector<int64_t> shape;
    int64_t numel = 0;
    std::vector<float> values;    // row-major f32 copy (empty for AR_OTHER)
};

// Calls `fn` once per tensor, in file order.  `want(name)` = false skips the conversion of a tensor nobody
// asked for (fn is not called for it).  Throws pcv::Error (PCV_ERR_IO / PCV_ERR_UNSUPPORTED) with the
// message in pcv_last_error().
void read_torch_archive(const std::string& path, const std::function<bool(const std::string&)>& want,
                        const std::function<void(const ArchiveTensor&)>& fn);

}  // namespace pcv

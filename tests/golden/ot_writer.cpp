// Writes a `rust_model.ot` exactly the way the reference's converter does: tch's Tensor::save_multi is
//     torch::serialize::OutputArchive archive;  archive.write(name, tensor) ...;  archive.save_to(path);
// (rust-bert utils/convert_model.py -> `convert-tensor` -> tch, scripts/install_models.sh:36).
// Test-fixture tool only: compiled against the libtorch inside the PyTorch wheel by gen_ot_fixture.py, never
// shipped or linked into the library.
//
//   ot_writer <in.raw> <out.ot>      in.raw: per tensor  u32 name_len | name | u32 rank | i64 dims[rank] | f32 data
#include <torch/torch.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    torch::serialize::OutputArchive archive;
    uint32_t name_len;
    int count = 0;
    while (std::fread(&name_len, 4, 1, f) == 1) {
        std::string name(name_len, '\0');
        uint32_t rank;
        if (std::fread(&name[0], 1, name_len, f) != name_len || std::fread(&rank, 4, 1, f) != 1) return 4;
        std::vector<int64_t> dims(rank);
        if (rank && std::fread(dims.data(), 8, rank, f) != rank) return 4;
        torch::Tensor t = torch::empty(dims, torch::kFloat32);
        if (std::fread(t.data_ptr<float>(), 4, (size_t)t.numel(), f) != (size_t)t.numel()) return 4;
        archive.write(name, t, /*is_buffer=*/false);
        ++count;
    }
    std::fclose(f);
    archive.save_to(argv[2]);
    std::printf("%d tensors -> %s\n", count, argv[2]);
    return 0;
}

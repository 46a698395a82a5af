// scan.h — device data layout + launcher interface of the similarity scan.
//
// HBM layout of a corpus segment ("row-block interleaved", DESIGN.md §HBM layout):
//   rows are grouped in blocks of 32; features are padded to Dp (multiple of 64) and cut in
//   16-byte pieces f4 = feature/4; the float4 of (block b, piece f4, row r) lives at
//       blk[(b * D4 + f4) * 32 + r]            D4 = Dp/4
//   so one wave instruction (lanes r=0..31 | h=0,1) reads two contiguous 512-byte runs and each
//   lane receives exactly the 8 consecutive features an MFMA 32x32x16 A-fragment wants.
//   Same bytes as row-major f32 (plus zero padding), permuted at 16-byte granularity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/perceive_hip.h"

namespace pcv {

constexpr int kBlockRows = 32;      // rows per corpus block
constexpr int kMaxSeg = 8;          // corpus segments one scan launch can walk
constexpr int kMaxK = 128;          // largest num_results the running top-k slots hold
constexpr int kSeedPartRows = 512;   // rows one seed workgroup ranks
constexpr int kSeedParts = 32;       // seed workgroups per query group -> up to 16384 seed rows
constexpr int kMaxWaveQueries = 4;  // wave-reduction kernel handles 1..4 queries per pass
constexpr int kHot = 256;            // uint32 words between per-query hot words (tau, cand_cnt): 1 KB apart,
                                     // so the device-wide atomics on them do not queue on one HBM channel
constexpr int kMfmaQueries = 128;   // MFMA kernel handles up to 128 queries per pass

struct SegDesc {
    const float4* blk;   // blocked matrix
    const float* scale;  // [nblocks*32] 1/|x| (cosine) or 1 (dot); 0 = row not searchable
    const int64_t* ids;  // [nblocks*32] item ids, or nullptr -> id = id0 + row
    int64_t id0;
    int64_t pos0;        // global position of row 0
    uint32_t nrows;
    uint32_t nblocks;
    uint32_t blk0;       // first block index of this segment in the launch's block numbering
    uint32_t pad;
};

struct ScanParams {
    SegDesc seg[kMaxSeg];
    int nseg;
    uint32_t total_blocks;
    int D4;              // Dp / 4
    int B;               // queries in this pass
    int k;
    int metric;
    const float* qf32;       // [B][Dp]   scan-side query (normalised for cosine), zero padded
    const uint16_t* qbf16;   // [64][Dp]  same, rounded to bf16; rows >= B are zero
    const float* qraw;       // [B][Dp]   original query values (exact rescoring)
    const double* qnorm2;    // [B]       f64 |q|^2
    const float* margin;     // [B]       2*eps in score units: rows with s < tau - margin are dropped
    uint32_t* tau;           // [B*kHot]  ordered key of the running k-th best approximate score (word q*kHot)
    uint32_t* slots;         // [B][kMaxK] ordered keys of k distinct rows' approximate scores
    uint32_t* cand_cnt;      // [B*kHot]  survivors emitted per query (word q*kHot)
    uint32_t* cand_cnt_out;  // [B]       compact copy written by select_kernel for the host
    uint64_t* cand;          // [B][cand_cap]  (segment index << 32) | row
    float* cand_s;           // [B][cand_cap]  screening score the row was emitted with
    double* cand_score;      // [B][cand_cap]  canonical score, filled by the rescoring kernel
    uint32_t* seed_part;     // [B][kSeedParts][kMaxK] per-part seed keys
    uint32_t cand_cap;
    uint32_t seed_blocks;    // leading blocks of segment 0 ranked by the seed kernel
    uint32_t flags;          // bit 0: plain (temporal) corpus loads instead of nt; bits 8..15: workgroups per CU override (tuning)
};

// float <-> order-preserving uint32 key (for atomicMax / CAS on scores)
__host__ __device__ static inline uint32_t f32_key(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float key_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}
constexpr uint32_t kKeyNegInf = 0x007fffffu;  // f32_key(-inf)

struct pcv_hit_dev {
    double score;
    int64_t pos;
    int64_t id;
};

// ---- launchers (scan_kernels.hip) ----
void launch_pack_rows(hipStream_t st, const float* rows_rowmajor, int64_t n, int D, int D4, float4* blk,
                      uint32_t nblocks, uint32_t row0);
void launch_row_scales(hipStream_t st, const float4* blk, uint32_t nblocks, uint32_t nrows, int D4, int metric,
                       float* scale, uint32_t* max_norm_bits);
void launch_synth_fill(hipStream_t st, float4* blk, uint32_t nblocks, uint32_t nrows, uint32_t row0, int D, int D4,
                       uint64_t seed, int64_t first_row, int normalize);
void launch_gather_rows(hipStream_t st, const SegDesc* d_segs, int nseg, const int64_t* d_pos, int64_t n, int D,
                        int D4, float* out_rows, int64_t* out_ids);
void launch_prep_queries(hipStream_t st, const float* d_queries, int B, int D, int Dp, int metric, float eps_rel,
                         float max_norm, int k, float* qf32, uint16_t* qbf16, float* qraw, double* qnorm2,
                         float* margin, uint32_t* tau, uint32_t* slots, uint32_t* cand_cnt);
// `p` is the host copy (shapes for the launch geometry), `dp` the same struct resident in device
// memory: the kernels index p.seg[] at run time, which a by-value kernel argument would force
// through scratch memory.
void launch_seed(hipStream_t st, const ScanParams& p, const ScanParams* dp);  // seed_partial + seed_merge
void launch_scan_wave(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus);
void launch_scan_mfma(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus);
int mfma_pass_queries(int Dp);  // queries one MFMA pass can take at this padded dim (LDS-limited), 0 = none
void launch_rescore(hipStream_t st, const ScanParams& p, const ScanParams* dp);
void launch_select(hipStream_t st, const ScanParams& p, const ScanParams* dp, pcv_hit_dev* out);
void launch_merge(hipStream_t st, const pcv_hit_dev* lists, int n_shards, int B, int k, pcv_hit_dev* out,
                  int flagged = 0);
void launch_overflow_flag(hipStream_t st, const uint32_t* cnt, int B, uint32_t cap, pcv_hit_dev* rec);
void launch_similarity_matrix(hipStream_t st, const float* a, int B, const float* m, int64_t N, int D, int cosine,
                              float* out);

}  // namespace pcv

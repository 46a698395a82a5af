// scan.h — device data layout + launcher interface of the similarity scan.
//
// HBM layout of a corpus segment ("row-block interleaved", DESIGN.md §HBM layout):
//   rows are grouped in blocks of 32; features are padded to Dp (multiple of 64) and cut in
//   16-byte pieces f4 = feature/4; the float4 of (block b, piece f4, row r) lives at
//       blk[(b * D4 + f4) * 32 + r]            D4 = Dp/4
//   so one wave instruction (lanes r=0..31 | h=0,1) reads two contiguous 512-byte runs and each
//   lane receives exactly the 8 consecutive features an MFMA 32x32x16 A-fragment wants.
//   Same bytes as row-major f32 (plus zero padding), permuted at 16-byte granularity.
//
// Screening copy (optional, one per segment): the same rows, already multiplied by their scale and rounded to
// bf16 — exactly the A operand the MFMA screen builds from the f32 rows — in 16-byte pieces of 8 features:
//       blk16[(b * D8 + f8) * 32 + r]          D8 = Dp/8
// The coarse screen then streams 2 bytes per feature instead of 4; the f32 rows are read only for the coarse
// survivors (fine screen) and the finalists (exact rescoring), so results are unchanged.
//
// Int8 screening copy (PCV_SCREEN_COPY_INT8): y = row x scale quantised per 32-row block, x^_i = rint(y_i * s_row) in [-127, 127],
// s_row = s_blk = 127 / max|y_i| over the block's searchable rows, in 16-byte pieces of 16 features (feature dimension padded
// to a multiple of 128):
//       blk8[(b * D16 + f16) * 32 + r]         D16 = roundup(Dp, 128) / 16;      scale8[b] = s_blk
// The screen is then an exact integer dot product (v_mfma_i32_32x32x32_i8) of quantised row and quantised query,
// 384 B per 384-d vector, with the certified bound
//       |c - acc / (s_row s_q)| <= |q'|_1 * 0.5 / s_row  +  |x^|_1 / s_row * 0.5 / s_q        (+ the f32 term eps32)
// and |x^|_1 <= sqrt(D) (s_row |y|_2 + 0.5 sqrt(D)), so one float per block is all the test needs (any s_row <= 127 / max|y_i|
// of the row keeps both |x^_i| <= 127 and |y_i - x^_i / s_row| <= 0.5 / s_row).
//
// Mid copy (optional, built when the coarse screen of a corpus lets many rows through: clustered embeddings): y = row x scale
// quantised per row to int16, Y_i = rint(y_i * s2), s2 = 32766 / max|y_i|, stored ROW-MAJOR, Dp int16 per row:
//       mid16[row * (Dp / 8) + piece]          scale16[row] = s2  (NaN = row not searchable)
// A coarse survivor's exact-f32 check reads its f32 row out of the blocked layout — 96 pieces of 16 bytes, 512 bytes apart:
// 96 cache lines, 12 KB of traffic for 1536 bytes.  With the mid copy the fine screen first reads the row's 768 contiguous bytes
// (6 lines) and drops it unless   q'.Y / s2  >=  tau - (|q'|_1 * 0.5002 / s2 + margin32 * 1.5)   — |y_i - Y_i / s2| <= 0.5002 / s2,
// so the bound is ~1e-4 in cosine, the size of the f32 margin itself: what passes goes on to the f32 row as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/perceive_hip.h"

namespace pcv {

constexpr int kBlockRows = 32;      // rows per corpus block
constexpr int kMaxK = PCV_MAX_RESULTS;  // largest num_results the running top-k slots hold (128)
constexpr int kSeedPartRows = 256;   // rows one seed workgroup ranks (one per thread)
constexpr int kSeedParts = 64;       // seed workgroups per query group -> up to 16384 seed rows
constexpr int kMaxWaveQueries = 4;  // wave-reduction kernel handles 1..4 queries per pass
constexpr int kHot = 256;            // uint32 words between per-query hot words (tau, cand_cnt): 1 KB apart,
                                     // so the device-wide atomics on them do not queue on one HBM channel
constexpr int kMfmaQueries = 256;   // most queries of one pass (the int8 scan up to 384-d: 256; the other MFMA scans 128)
constexpr int kScale8Stride = 1;    // floats per block in SegDesc::scale8: the block's quantisation scale

// One corpus segment as a scan launch sees it.  A launch walks any number of them: the table lives in
// device memory next to the ScanParams (one source of the reference = one or more segments,
// search.rs:24-27; every incremental add can append one).
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
    const uint4* blk16;  // bf16 screening copy (see below), or nullptr
    const uint4* blk8;   // int8 screening copy, or nullptr
    const uint4* mid16;  // row-major 16-bit copy (see below), or nullptr
    const float* scale16;
    const float* scale8; // [nblocks] quantisation scale of the int8 copy's blocks (NaN = no searchable row in the block)
};

struct pcv_hit_dev {
    double score;
    int64_t pos;
    int64_t id;
};

// More results than a pass ranks (num_results > kMaxK): pcv_searcher_search goes over the rows again for the next kMaxK, and
// again, each pass counting only the rows that rank strictly AFTER the last hit of the pass before it — the pass's ceiling, one
// per query (the reference has no limit on num_results: search.rs:157-182; perceive-cli's --num-results is user input).
//   score, pos : the canonical score and position of that hit; +inf / -1: no ceiling for this query; -inf / INT64_MAX: nothing
//                ranks after it (the query has all the rows there are)
//   lo, hi     : the same boundary in units of the f32 screening score, one fine margin below and above it.  A row with
//                screening score s < lo certainly ranks after the boundary: it counts and may raise the running thresholds;
//                lo <= s <= hi: it counts (the exact ranking decides) but must not raise them; s > hi: it ranks at or before
//                the boundary and is not listed.  The thresholds thus rest on k rows that count, and the argument for the
//                screens (scan_kernels.hip) holds among the rows that count.
struct CeilRec {
    double score;
    int64_t pos;
    float lo, hi;
};

// Everything one pass needs, resident in device memory (uploaded with the segment table and the
// queries in ONE copy): the kernels index p.seg[] at run time, which a by-value kernel argument would
// force through scratch memory.
struct ScanParams {
    const SegDesc* seg;      // [nseg] device table, blk0 ascending
    int nseg;
    uint32_t total_blocks;
    int D;                   // embedding width
    int D4;                  // Dp / 4
    int B;                   // queries in this pass
    int k;
    int metric;
    uint32_t tile_rows;      // rows of the bf16 query tile the scan kernel stages (rows >= B are zeroed)
    const CeilRec* ceil;     // [B] the pass's ceilings, or nullptr (the first kMaxK results: every row counts)
    const float* queries;    // [B][D]    raw queries as the caller passed them
    float* qf32;             // [B][Dp]   scan-side query (normalised for cosine), zero padded
    uint16_t* qbf16;         // [128][Dp] same, rounded to bf16
    int8_t* q8;              // [128][Dp8] same, quantised per query to int8 (int8 screen; Dp8 = Dp rounded up to 128)
    float* q8c;              // [256][4]  s_q (quantisation scale, 0 = dead query), V_q of the int8 test (scan_mfma8_kernel), |q'|_1 (mid screen), -
    float* qraw;             // [B][Dp]   original query values, zero padded (exact rescoring)
    float* margin;           // [B]  coarse screen: rows with s16 < tau - margin are dropped       (eps16 + eps32)
    float* margin32;         // [B]  fine screen:   rows with s32 < tau - margin32 are dropped     (2 * eps32)
    uint32_t* tau;           // [B*kHot]  ordered key of the running k-th best f32 score (word q*kHot)
    uint32_t* tau_c;         // [256]     the same keys side by side (raised right after tau, so never above it): one load instead of a
                             //           64-line gather for a wave that wants all thresholds of the pass — for FEW readers (the one drain
                             //           wave per CU of scan_mfma8_kernel's DRAIN form, about once a microsecond).  Read by every wave at
                             //           every block these 256 bytes are one hot spot in one memory channel: the 4-waves-a-workgroup form
                             //           went from 0.94 to 1.04 ms at 12.5M rows with it, which is what the 1 KB spacing of `tau` is for
    uint32_t* slots;         // [B][kMaxK] ordered keys of k distinct rows' f32 scores
    uint32_t* cand_cnt;      // [B*kHot]  survivors emitted per query (word q*kHot); word q*kHot + 32: rows that passed the COARSE screen
                             //           (statistics only: pcv_scan_stats.coarse_survivors)
    uint64_t* cand;          // [B][cand_cap]  (segment index << 32) | row
    float* cand_s;           // [B][cand_cap]  f32 screening score the row was emitted with
    pcv_hit_dev* out;        // [B][k] device results
    pcv_hit_dev* out_host;   // pinned host mirror of `out`, or nullptr
    uint32_t* cnt_host;      // [B] pinned host: survivors per query, uncapped (the host sizes a rerun from it)
    uint32_t* coarse_host;   // [2][256] pinned host: coarse survivors per query (MFMA scans), then the pairs the mid screen let through
    pcv_hit_dev* flag_rec;   // overflow record behind a shard's hit list (device), or nullptr
    uint32_t cand_cap;
    uint32_t seed_blocks;    // blocks of segment 0 ranked by the seed kernel: blocks i << seed_shift, i < seed_blocks — spread
    uint32_t seed_shift;     // over the whole segment, so that rows stored in an order that goes with their content (by topic, by
                             // date) still give a sample of all of it
    uint32_t flags;          // set by the searcher: bit 4: every segment has its bf16 screening copy, stream that; bit 6: every segment
                             // has its int8 screening copy, stream that.  Tuning / comparison (PCV_SCAN_FLAGS at searcher creation):
                             // bit 0: plain (temporal) corpus loads instead of nt; bit 1: 16 queries per seed workgroup (VALU seed);
                             // bit 2: the VALU seed kernel; bit 3: the 128-query tile instead of the block-holding int8 scan;
                             // bit 5: no speculative start threshold; bit 7: no learned part of it; bits 8..15: workgroups per CU;
                             // bits 16..23: seed workgroups; bits 24..27: chunk buffers
    unsigned long long* stamps;  // diagnostic build only (-DPCV_STAMPS, tools/build_stamps.sh): 8 words per wave of the scan launch, else nullptr
    float eps16, eps32;      // |s - c| bounds of the bf16 / f32 screening scores, relative to |q||x|
    float max_norm;          // upper bound of |x| over the corpus (dot metric margins)
    // Speculative start threshold (MFMA scans; 0 = off).  The k slots the seed kernel fills are the best scores of k
    // disjoint groups of seed rows; min(slots) is a certified k-th best, but a weak one, and every wave screens its
    // first block against it.  set_guess (quantize_queries_kernel / set_guess_kernel) therefore raises tau to the spec_rank-th LARGEST slot — a guess
    // that at least k rows of the whole pass score that high — and rescore_select_kernel checks the guess: it counts the
    // survivors whose f32 score is above it by the fine margin; fewer than k and the query reports 0xffffffff survivors,
    // the pass is repeated without the guess (searcher.cpp: finish_pass).  A checked guess keeps the result exact:
    // with k rows at or above it the true k-th best is too, and no screen drops a row at or above tau - its margin.
    // The host picks spec_rank so that, for rows in an order unrelated to the query, a guess fails with probability
    // < 1e-6 (C(k-1, j) (seed rows / rows)^j: j of the k-1 best rows of the pass would have to be seed rows).
    //
    // On top of that distribution-free guess the searcher learns one from its own passes (spec_gap, NaN = none): how far
    // the k-th best score of a pass ended up above the MEDIAN seed slot of its query (both come back through pinned
    // memory: kth_host, spec_base_host); a fraction of the smallest gap of the recent passes is added to the median
    // slot of the next queries.  Self-calibrating, and checked like the other: a corpus whose queries differ a lot
    // simply learns a small gap.
    uint32_t* spec;          // [128] key of the guess per query (kKeyNegInf: none), written by set_guess
    int spec_rank;
    float spec_gap;
    float spec_spread;       // learned: mean (best - median) seed slot; a query takes the learned gap only if its own is within 50 %
    float* spec_base_host;   // [128] pinned: median seed slot per query (NaN: none)
    float* spec_top_host;    // [128] pinned: best seed slot per query
    float* kth_host;         // [128] pinned: k-th best exact score per query (NaN: fewer than k hits)
};
constexpr uint32_t kSpecFailed = 0xffffffffu;  // cnt_host value of a query whose speculative threshold did not hold

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

// ---- launchers (scan_kernels.hip); they throw pcv::Error on a bad shape or a failed HIP call ----
void launch_pack_rows(hipStream_t st, const float* rows_rowmajor, int64_t n, int D, int D4, float4* blk, uint32_t row0);
void launch_iota_ids(hipStream_t st, int64_t* ids, int64_t first, int64_t n);
// scales of the rows in blocks [first_block, nblocks) of a segment
void launch_row_scales(hipStream_t st, const float4* blk, uint32_t first_block, uint32_t nblocks, uint32_t nrows, int D4,
                       int metric, float* scale, uint32_t* max_norm_bits);
// screening copy of the rows in blocks [first_block, nblocks): bf16(row * scale), zeros where scale == 0
void launch_coarse_pack(hipStream_t st, const float4* blk, const float* scale, uint4* blk16, uint32_t first_block, uint32_t nblocks,
                        int D4);
// int8 screening copy + quantisation scales of the rows in blocks [first_block, nblocks)
void launch_coarse_pack8(hipStream_t st, const float4* blk, const float* scale, uint4* blk8, float* scale8, uint32_t first_block,
                         uint32_t nblocks, int D4);
void launch_scan_mfma8(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus);  // quantises the queries first
int mfma8_pass_queries(int Dp);  // queries one int8 MFMA pass can take (LDS-limited)
// mid copy + its scales of rows [first_row, nrows) of a segment
// (scale8: the quantisation scales of the segment's int8 copy if it covers these rows — the copy is then made block by block with
// the blocks' scales — or nullptr: row by row, each with its own)
void launch_mid_pack(hipStream_t st, const float4* blk, const float* scale, const float* scale8, uint4* mid16, float* scale16,
                     uint32_t first_row, uint32_t nrows, int D4);
void launch_synth_fill(hipStream_t st, float4* blk, uint32_t nrows, uint32_t row0, int D, int D4, uint64_t seed,
                       int64_t first_row, int normalize, uint32_t n_clusters, float noise, float amp_lo = 0.0f, float amp_hi = 0.0f);
void launch_gather_rows(hipStream_t st, const SegDesc* d_segs, int nseg, const int64_t* d_pos, int64_t n, int D,
                        int D4, float* out_rows, int64_t* out_ids);
// `p` is the host copy (shapes for the launch geometry), `dp` the same struct resident in device memory.
void launch_upload(hipStream_t st, const void* src_pinned, void* dst, size_t bytes);  // pinned host -> device, on the compute queue
void launch_prep_seed(hipStream_t st, const ScanParams& p, const ScanParams* dp, const SegDesc& seg0);
void launch_scan_wave(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus);
void launch_scan_mfma(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus);
int mfma_pass_queries(int Dp);   // queries one MFMA pass can take at this padded dim (LDS-limited), 0 = none
uint32_t mfma_tile_rows(int B);  // rows of the bf16 query tile the MFMA kernel stages for B queries
void launch_rescore_select(hipStream_t st, const ScanParams& p, const ScanParams* dp);
void launch_reset_scan_state(hipStream_t st, uint32_t* tau, uint32_t* slots, uint32_t* cand_cnt);
void launch_merge(hipStream_t st, const pcv_hit_dev* lists, int n_shards, int B, int k, pcv_hit_dev* out,
                  int flagged = 0);
void launch_similarity_matrix(hipStream_t st, const float* a, int B, const float* m, int64_t N, int D, int cosine,
                              float* out);

}  // namespace pcv

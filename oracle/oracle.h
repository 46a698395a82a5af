/*
 * oracle.h — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (perceive_amd/, libperceive_hip.so) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED BY THE REFERENCE: dimfeld/perceive is Rust-only (no cargo/rustc in this image),
 * its arithmetic lives in un-vendored crates (tch 0.10.1 -> libtorch, rust-bert 0.19.0 @3bf86331,
 * hnsw_rs 0.1.17, ndarray 0.15.6) and its own tests hold no vector for this path (SURVEY.md §4,
 * §8c).  The restatement is cross-checked instead against PyTorch-CPU executing the same ATen
 * operator sequence tch binds (tests/golden/, generator scripts committed beside the vectors).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#ifndef PCV_ORACLE_H
#define PCV_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- lib.rs:63-77 --------------------------------------------------------------------------- */
/* dot_product (lib.rs:63-65): out[b][n] = sum_i a[b][i]*m[n][i], f32 accumulate in index order. */
void orc_dot_product(const float* a, int B, const float* m, int64_t N, int D, float* out);
/* cosine_similarity_multi_query (lib.rs:73-77): both sides divided by their f32 L2 norm over
 * dim 1 (no epsilon: a zero row yields NaN exactly like the reference), then dot_product. */
void orc_cosine_similarity_multi_query(const float* a, int B, const float* m, int64_t N, int D, float* out);
/* cosine_similarity_single_query (lib.rs:67-71): query [D] normalised over dim 0. out [N]. */
void orc_cosine_similarity_single_query(const float* q, const float* m, int64_t N, int D, float* out);

/* ---- canonical exact ranking (DESIGN.md §canonical ranking) --------------------------------- */
/* Canonical f64 score of one (query,row) pair: products of the f32 inputs are exact in f64 and are
 * added in index order.  metric 0: dot/(sqrt(|q|^2)*sqrt(|x|^2)); metric 1: dot.
 * Returns NaN when the score is undefined (zero / non-finite norm). */
double orc_canonical_score(const float* q, const float* x, int D, int metric);
/* Exact top-k of one query over N rows: descending canonical score, ties -> lower position;
 * undefined scores never returned.  out_* have room for k.  Returns the number filled. */
int orc_topk(const float* q, const float* m, int64_t N, int D, int metric, int k, int64_t* out_pos,
             double* out_score);

/* ---- search.rs --------------------------------------------------------------------------------*/
/* NdArrayDistance::eval (search.rs:269-278): max(0, 1 - dot/len) in f32. */
float orc_ndarray_distance(const float* a, const float* b, int D);
/* Searcher::search_vector (search.rs:157-182) with an exact scan in place of hnsw.search:
 * rows carry (id, source_id); only rows whose source is in `sources` take part (search.rs:166);
 * result sorted ascending by distance (search.rs:179), ties -> lower position; truncated to k.
 * Returns the number filled. */
int orc_search_vector(const float* q, const float* m, const int64_t* ids, const int64_t* source_of_row,
                      int64_t N, int D, const int64_t* sources, int n_sources, int k, int64_t* out_ids,
                      float* out_dist);
/* serialize_embedding / deserialize_embedding (search.rs:281-294). */
void orc_serialize_embedding(const float* v, size_t n, uint8_t* out);
size_t orc_deserialize_embedding(const uint8_t* blob, size_t n_bytes, float* out);

/* ---- synthetic data (twin of perceive_amd/csrc/synth.h; not part of the reference) ---------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Row `row` of the synthetic corpus `seed`: D values (D % 4 == 0), approx N(0,1); normalize != 0
 * divides by the f64-accumulated L2 norm. */
void orc_synth_row(uint64_t seed, int64_t row, int D, int normalize, float* out);
void orc_synth_rows(uint64_t seed, int64_t first_row, int64_t n, int D, int normalize, float* out);
/* Clustered rows: centroid(cluster(row))/sqrt(D) + noise * u(row) * synth_row(seed,row), u in [0.5,1.5) (bench.py --clustered). */
void orc_synth_rows_clustered(uint64_t seed, int64_t first_row, int64_t n, int D, int normalize, int n_clusters,
                              float noise, float* out);
/* Scaled rows: a(row) * synth_row(seed,row), a uniform in [amp_lo, amp_hi) (bench.py's 768-d dot-metric legs). */
void orc_synth_rows_scaled(uint64_t seed, int64_t first_row, int64_t n, int D, float amp_lo, float amp_hi, float* out);

/* ---- timed CPU baseline (bench.py cpu_baseline leg) ----------------------------------------- */
/* Fused single pass of lib.rs:67-77 + top-k over rows [0,N) of `m` for B queries on `threads`
 * host threads; f32 arithmetic.  out_pos/out_score: [B][k].  Returns seconds of wall time. */
double orc_baseline_scan_fused(const float* queries, int B, const float* m, int64_t N, int D, int k,
                               int threads, int64_t* out_pos, float* out_score);
/* "Reference-shaped": materialise the normalised corpus, then [B,N] score matrix, then select
 * (what libtorch does for lib.rs:73-77; >= 3x the algorithmic traffic). */
double orc_baseline_scan_reference_shaped(const float* queries, int B, const float* m, int64_t N, int D,
                                          int k, int threads, int64_t* out_pos, float* out_score);
int orc_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif

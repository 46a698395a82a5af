/*
 * perceive_hip.h — C ABI of the MI355X-native replacement for perceive-core's hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: the path
 * sits behind ordinary Rust `pub` items of crate `perceive-core`.  Each entry point below names
 * the reference item it replaces (paths relative to the reference checkout).  A Rust shim
 * (`extern "C"` block + `Model`/`Searcher` wrappers) binds exactly these symbols.
 *
 * Conventions
 *   - every function returns a pcv_status (0 = ok); no C++ exception or abort crosses the boundary;
 *   - pcv_last_error() returns a thread-local, NUL-terminated description of the last failure;
 *   - all handles are opaque; outputs are caller-allocated; the library never frees caller memory;
 *   - host pointers unless a parameter is documented as a device pointer;
 *   - a searcher handle may be searched from several host threads (calls serialise on an internal
 *     mutex, like the reference's model worker channel `model.rs:161,187`).
 */
#ifndef PERCEIVE_HIP_H
#define PERCEIVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pcv_status;
enum {
    PCV_OK = 0,
    PCV_ERR_INVALID = 1,      /* bad argument / bad handle state                                */
    PCV_ERR_DEVICE = 2,       /* a HIP call failed (no GPU, out of memory, launch failure)      */
    PCV_ERR_UNSUPPORTED = 3,  /* shape outside what the kernels implement                       */
    PCV_ERR_IO = 4,           /* weight / vocab file problems                                   */
    PCV_ERR_INTERNAL = 5
};

/* Ranking metric of a searcher.
 * PCV_METRIC_COSINE : score = cos(q, x), the brute-force path `lib.rs:67-77`
 *                     (cosine_similarity_single_query / _multi_query); results best-first.
 * PCV_METRIC_DOT    : ranks by the raw dot product and reports the reference Searcher's distance
 *                     `max(0, 1 - dot/len)` (`search.rs:266-279`), ascending like `search.rs:179`. */
enum { PCV_METRIC_COSINE = 0, PCV_METRIC_DOT = 1 };

typedef struct pcv_ctx pcv_ctx;
typedef struct pcv_searcher pcv_searcher;
typedef struct pcv_model pcv_model;
typedef struct pcv_tokenizer pcv_tokenizer;

/* ---- library / device context ------------------------------------------------------------- */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* pcv_last_error(void);

/* Library version string, e.g. "perceive-hip 0.1 (gfx950)". */
const char* pcv_version(void);

/* Number of visible HIP devices (0 when there is no GPU; never fails). */
int pcv_device_count(void);

/* Bind a context to one GPU.  One process drives one GPU (one rank per device); the context owns
 * the HIP stream every kernel of its handles is launched on.
 * Replaces: `tch::Device::cuda_if_available()` at model.rs:117 / configs.rs:117. */
pcv_status pcv_init(int device_index, pcv_ctx** out_ctx);
pcv_status pcv_shutdown(pcv_ctx* ctx);
/* Block until everything queued on the context's stream has finished. */
pcv_status pcv_synchronize(pcv_ctx* ctx);
/* The context's hipStream_t (as void*), for callers that order their own work against it. */
void* pcv_stream(pcv_ctx* ctx);
/* adopt != 0: queue all further work of this context on the caller's hipStream_t `hip_stream` (NULL = the
 * device's default stream) — e.g. the stream a host framework runs its collectives against — so that
 * both are ordered without host synchronisation.  adopt == 0: back to the context's own stream.  The
 * handle must come from the HIP runtime this library is bound to.  Drains the stream in use first. */
pcv_status pcv_set_stream(pcv_ctx* ctx, void* hip_stream, int adopt);

/* Plain device buffers for hosts that have no HIP binding of their own (the per-shard hit lists that
 * an RCCL all-gather exchanges live in such buffers). */
pcv_status pcv_device_alloc(pcv_ctx* ctx, size_t n_bytes, void** out_dptr);
pcv_status pcv_device_free(pcv_ctx* ctx, void* dptr);
pcv_status pcv_copy_to_host(pcv_ctx* ctx, void* dst_host, const void* src_dev, size_t n_bytes);
pcv_status pcv_copy_to_device(pcv_ctx* ctx, void* dst_dev, const void* src_host, size_t n_bytes);

/* ---- embedding blob codec (search.rs:281-294) ---------------------------------------------- */

/* deserialize_embedding: `n_bytes/4` little-endian f32 (trailing bytes that do not fill a chunk
 * are an error here; the reference would panic on the short chunk). */
pcv_status pcv_deserialize_embedding(const uint8_t* blob, size_t n_bytes, float* out, size_t out_cap,
                                     size_t* out_len);
/* serialize_embedding: writes 4*n bytes, little-endian. */
pcv_status pcv_serialize_embedding(const float* v, size_t n, uint8_t* out, size_t out_cap);

/* ---- Searcher (search.rs:29-260) ------------------------------------------------------------ */

/* Searcher::build, part 1 (search.rs:38-56): an empty index for `dim`-wide f32 embeddings. */
pcv_status pcv_searcher_create(pcv_ctx* ctx, int dim, int metric, pcv_searcher** out);
pcv_status pcv_searcher_destroy(pcv_searcher* s);

/* build_sources row insert (search.rs:87-113,146-148): append `n` rows (row-major [n][dim] f32,
 * host memory) with their item ids to source `source_id`.  The rows are uploaded and packed into the HBM
 * layout before the call returns (staged in bounded steps: the library keeps no host copy, so a corpus can
 * be streamed through a buffer of any size); they become searchable after pcv_searcher_finalize.
 * `ids` may be NULL: ids are then the row's running index in the source. */
pcv_status pcv_searcher_add_rows(pcv_searcher* s, int64_t source_id, const int64_t* ids,
                                 const float* rows, int64_t n);
/* Same, from the on-disk form: `n` blobs of dim*4 bytes each, back to back (search.rs:99,281). */
pcv_status pcv_searcher_add_blobs(pcv_searcher* s, int64_t source_id, const int64_t* ids,
                                  const uint8_t* blobs, int64_t n);
/* Synthetic rows generated on the device (never cross PCIe): row r (0-based within this call) of
 * the source is synth_row(seed, first_row + r); ids are first_row + r.  See DESIGN.md §synthetic
 * data; oracle/synth.c is the CPU twin.  normalize != 0 stores x/|x| instead of x. */
pcv_status pcv_searcher_add_synthetic(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                      int64_t first_row, int normalize);
/* Clustered synthetic rows (what a fixed-width screen has to survive on real sentence embeddings):
 * row = centroid(cluster(row)) / sqrt(dim) + noise * u(row) * synth_row(seed, row), u uniform in [0.5, 1.5), with
 * n_clusters seeded centroids; the cosines inside a cluster spread over ~noise^2 * dim.  n_clusters = 0: plain rows. */
pcv_status pcv_searcher_add_synthetic_clustered(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                                int64_t first_row, int normalize, int n_clusters, float noise);
/* Un-normalised synthetic rows whose norms spread — what a dot-product model (the reference's default
 * MsMarcoBertBaseDotV5, perceive-cli/state.rs:24) stores: row = a(row) * synth_row(seed, row), a(row) uniform in
 * [amp_lo, amp_hi) from a hash of the row number; 0 < amp_lo < amp_hi.  Bit-identical CPU twin: oracle/synth.c. */
pcv_status pcv_searcher_add_synthetic_scaled(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                             int64_t first_row, float amp_lo, float amp_hi);
/* Capacity hint (search.rs:138-140 sizes each source's index from its row count before inserting): the
 * host is about to add `n_rows` more rows to `source_id`, e.g. the COUNT(*) of the build query.  The rows
 * then land in one device segment instead of a chain of growing ones.  Optional; no-op for n_rows = 0. */
pcv_status pcv_searcher_reserve(pcv_searcher* s, int64_t source_id, int64_t n_rows);
/* Searcher::rebuild_source (search.rs:58-79): drop every row of `source_id`; follow with
 * add_* + finalize to install the replacement.  Unknown source: no-op. */
pcv_status pcv_searcher_clear_source(pcv_searcher* s, int64_t source_id);
/* The swap of Searcher::rebuild_source (search.rs:57-79: the new SourceSearch is built first and takes the old one's
 * place only once it exists): the rows of source `from_source_id` become the rows of `to_source_id`, whose old rows
 * are dropped; `from_source_id` disappears.  `from` unknown or empty: `to` is left absent (search.rs:67-69).  Build
 * the replacement under a staging id (add_* + finalize: a failure on the way leaves `to` untouched — clear the
 * staging id then), swap, finalize.  PCV_STAGING_SOURCE is the id this library's own loaders stage under: rows under it are
 * not counted (pcv_searcher_num_rows / num_sources / source_ids) and a search of "every source" (source_ids = NULL) does not
 * see them; between pcv_searcher_add_* and the next pcv_searcher_finalize every search fails (rows without scales), as after
 * any add — a rebuild has to be serialised against searches by the caller, like Searcher::rebuild_source's &mut self. */
#define PCV_STAGING_SOURCE INT64_MIN
pcv_status pcv_searcher_replace_source(pcv_searcher* s, int64_t from_source_id, int64_t to_source_id);
/* Pack pending rows into the HBM layout, compute row norms (set_searching_mode, search.rs:150). */
pcv_status pcv_searcher_finalize(pcv_searcher* s);

/* Searcher::build (search.rs:38-56) / rebuild_source (search.rs:58-79) straight from the reference's SQLite
 * file: runs `SELECT id FROM sources` and the items / item_embeddings join of search.rs:87-93 for
 * (model_id, model_version) — rows with `skipped` or `hidden_at` set never enter the index — streams the
 * embedding blobs into the device (one segment per source, sized from a COUNT), and finalizes.
 *   only_source   NULL: every source of the database; else that source is reloaded: its new rows are read, checked and
 *                 packed under PCV_STAGING_SOURCE first and replace the old ones only when all of them are in
 *                 (search.rs:57-79) — after a failure (a blob of the wrong size, an SQLite error) the old rows are
 *                 still there and searchable
 *   out_rows      rows loaded (may be NULL)
 * SQLite is bound at run time (libsqlite3.so.0); PCV_ERR_UNSUPPORTED if it is not installed. */
pcv_status pcv_searcher_load_sqlite(pcv_searcher* s, const char* db_path, uint32_t model_id, uint32_t model_version,
                                    const int64_t* only_source, int64_t* out_rows);

pcv_status pcv_searcher_dim(pcv_searcher* s, int* out_dim);
pcv_status pcv_searcher_num_rows(pcv_searcher* s, int64_t* out_rows);
/* Device segments the rows currently occupy (diagnostics: adds append in place while a segment has room). */
pcv_status pcv_searcher_num_segments(pcv_searcher* s, int* out_n);
pcv_status pcv_searcher_num_sources(pcv_searcher* s, int* out_n);
pcv_status pcv_searcher_source_ids(pcv_searcher* s, int64_t* out_ids, int cap);
/* Rows held for one source (0 if the source is unknown). */
pcv_status pcv_searcher_source_num_rows(pcv_searcher* s, int64_t source_id, int64_t* out_rows);
/* Read rows back (row-major) by global position (sources in insertion order, rows in insertion
 * order inside a source).  Test/diagnostic path. */
pcv_status pcv_searcher_get_rows(pcv_searcher* s, const int64_t* positions, int64_t n, float* out_rows,
                                 int64_t* out_ids);

/* Which scan kernel pcv_searcher_search uses. AUTO: wave-reduction kernel for n_queries <= 4,
 * MFMA tile kernel otherwise (up to 128 queries per corpus pass at dim <= 640; 256 with the int8 screening copy at
 * dim <= 384; among the ranks of a sharded search a pass is 128 queries on every rank, whatever copies each holds).
 * More queries than one pass takes are searched in several passes. */
enum { PCV_KERNEL_AUTO = 0, PCV_KERNEL_WAVE = 1, PCV_KERNEL_MFMA = 2 };
pcv_status pcv_searcher_set_kernel(pcv_searcher* s, int kernel);

/* Tuning: rows per query the candidate lists of a pass hold at first (default 8192; 20 bytes each, 128
 * lists).  A pass that needs more repeats itself with larger lists (pcv_scan_stats.overflow_reruns). */
pcv_status pcv_searcher_set_candidate_capacity(pcv_searcher* s, uint32_t n_candidates);

/* Diagnostic / comparison switches of one searcher; results never depend on them.  `flags` replaces what the
 * environment variable PCV_SCAN_FLAGS gave the searcher at creation (bit meanings: csrc/scan.h, ScanParams::flags —
 * bit 0 plain instead of non-temporal corpus loads, bit 3 the 128-query tile instead of the block-holding int8 scan,
 * bit 5 no speculative start threshold, bit 7 no learned part of it, bits 8..15 workgroups per CU, ...), plus
 *   PCV_TUNE_FAIL_COPY_ALLOC : while set, every allocation of a screening copy is treated as failed (how the tests
 *                              reach the out-of-memory branches of pcv_searcher_finalize). */
enum { PCV_TUNE_FAIL_COPY_ALLOC = 1073741824 }; /* bit 30 */
pcv_status pcv_searcher_set_tuning(pcv_searcher* s, uint32_t flags);

/* Screening copy: next to the f32 rows a segment can hold the same rows, already scaled, in a narrow form that
 * only the coarse screen reads; the f32 rows are then read for the rows that pass it (fine screen) and for the
 * finalists (exact ranking), so results are identical — the screens are certified bounds, not approximations.
 *   PCV_SCREEN_COPY_BF16 : the operand of the bf16 MFMA screen ready-made, 2 bytes per feature (+50 % HBM);
 *                          coarse margin 2^-8 relative
 *   PCV_SCREEN_COPY_INT8 : rows quantised per row to int8, 1 byte per feature + 4 bytes per row (+25 % HBM), screened
 *                          by an exact integer dot product with the quantised query; coarse margin ~0.024 in cosine
 *                          for 384-d unit rows (certified per row and query from the quantisation steps): more rows
 *                          reach the fine screen, a quarter of the bytes are streamed; dimensions up to 1024
 *   PCV_SCREEN_COPY_AUTO (default): INT8 (BF16 for rows wider than 1024 features), built at finalize; given up — for good, on this searcher — when an
 *                          allocation for rows or for a copy fails (the f32 rows are scanned then)
 *   PCV_SCREEN_COPY_OFF  : never built; existing copies are freed
 * BF16 / INT8 asked for explicitly: a failed copy allocation is an error at finalize.  Takes effect at the next
 * finalize (OFF: at once).  100M x 384: 153.6 GB of rows + 38.4 GB (INT8) or 76.8 GB (BF16). */
enum { PCV_SCREEN_COPY_OFF = 0, PCV_SCREEN_COPY_BF16 = 1, PCV_SCREEN_COPY_AUTO = 2, PCV_SCREEN_COPY_INT8 = 3 };
pcv_status pcv_searcher_set_screening_copy(pcv_searcher* s, int mode);

/* Mid copy: a third, optional representation of the rows — 16-bit fixed point per row, ROW-MAJOR (2 bytes per feature + 4 per
 * row: 76.8 GB for 100M x 384) — read by the fine screen in front of the f32 rows.  Worth its memory where the f32 rows of the
 * coarse (int8) screen's survivors are a visible share of a pass: corpora that let thousands of rows per query through
 * (clustered embeddings), wide rows, small shards.  A coarse survivor then costs 2 contiguous bytes per feature instead of
 * one 128-byte cache line per 4 features of the blocked f32 layout, and only rows within ~1e-4 of the running threshold go on
 * to their f32 row.  Results are identical with and without it (a certified bound, like the other screens).
 *   PCV_MID_COPY_AUTO (default): built by a search call after 2 passes in a row over int8 copies in which the coarse screen let
 *                        more than 4096 rows per query through, or in which the survivors' f32 rows came to more than 1/25 of
 *                        the bytes streamed — if the memory is there (4 GB stay free); dropped again, before the screening
 *                        copies, when an allocation for rows or for a screening copy fails
 *   PCV_MID_COPY_ON    : built at the next finalize (an allocation failure is an error)
 *   PCV_MID_COPY_OFF   : never built; an existing one is freed */
enum { PCV_MID_COPY_OFF = 0, PCV_MID_COPY_AUTO = 1, PCV_MID_COPY_ON = 2 };
pcv_status pcv_searcher_set_mid_copy(pcv_searcher* s, int mode);
/* AUTO builds the mid copy BESIDE the searches (a stream and a helper thread of its own): the search call that decides to
 * build it returns like any other, and so do the calls after it, without the copy, until it is complete — while it is being
 * built they share the memory system with the build.  This call waits for such a build to finish (a benchmark that wants
 * the steady state; nothing needs it for correctness).  Returns at once when nothing is under way. */
pcv_status pcv_searcher_wait_background(pcv_searcher* s);

/* Most hits ONE PASS over the rows ranks per query.  pcv_searcher_search takes any num_results (search.rs:157-182 has no limit;
 * the reference's callers ask for 10 and 20, perceive-cli's --num-results is user input): beyond this many it goes over the rows
 * again for the next PCV_MAX_RESULTS below the last hit of the pass before, and so on — every pass exact, results as from one
 * ranking.  The entry points that exchange fixed-size hit lists (pcv_searcher_search_device*, pcv_searcher_search_sharded,
 * pcv_merge_topk*) take at most this many. */
enum { PCV_MAX_RESULTS = 128 };

/* Searcher::search_vector (search.rs:157-182), batched over `n_queries` query vectors.
 *   queries      [n_queries][dim] f32
 *   source_ids   sources to search (search.rs:166 filter): NULL = all sources (n_sources ignored);
 *                non-NULL = exactly the n_sources listed ones, so n_sources = 0 matches nothing,
 *                like `sources.contains(..)` on an empty slice
 *   k            num_results
 *   out_ids      [n_queries][k] item ids, best first
 *   out_scores   [n_queries][k] cosine (COSINE) or reference distance (DOT)
 *   out_counts   [n_queries] entries actually filled (< k when fewer valid rows exist)
 * Result order: COSINE descending cosine, DOT ascending distance; ties -> lower global position.
 * Rows whose norm is 0 or not finite (cosine undefined; the reference would yield NaN and panic at
 * search.rs:179) are never returned.  Exactness: the returned set is the exact top-k under the
 * canonical f64 score (DESIGN.md §canonical ranking), not an approximation. */
pcv_status pcv_searcher_search(pcv_searcher* s, const float* queries, int n_queries,
                               const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                               float* out_scores, int* out_counts);

/* One entry of a per-shard result list, the unit exchanged between GPUs (all-gather payload). */
typedef struct pcv_hit {
    double score;  /* canonical f64 score (cosine or dot)                 */
    int64_t pos;   /* global row position (shard offset already applied)  */
    int64_t id;    /* item id                                             */
} pcv_hit;

/* Row position offset of this searcher's shard inside the whole (multi-GPU) corpus. */
pcv_status pcv_searcher_set_shard_offset(pcv_searcher* s, int64_t first_global_pos);

/* Local (per-shard) exact top-k, results left on the device: `d_out` is a DEVICE pointer to
 * [n_queries][k] pcv_hit; unfilled entries have pos = -1.  Runs on the context stream; the call
 * returns after every pass has been collected (`async` is accepted for compatibility and ignored: use
 * the begin/end pair below to overlap an exchange with the host). */
pcv_status pcv_searcher_search_device(pcv_searcher* s, const float* queries, int n_queries,
                                      const int64_t* source_ids, int n_sources, int k, void* d_out,
                                      int async);

/* The same per-shard search split in two so that the exchange can be queued behind it without a host
 * round trip.  `begin` queues the whole pass on the context stream and returns at once; `d_out` then
 * holds n_queries*k hits followed by ONE extra pcv_hit whose `pos` is 1 if the pass must be repeated —
 * a candidate list overflowed, or a speculative start threshold did not hold (pcv_scan_stats) — and the
 * hits are then incomplete; else 0.  `end` waits for the stream, books the statistics and prepares the
 * repeat (larger lists; no guess).
 * No other call may use the searcher between the two.  PCV_ERR_UNSUPPORTED only when n_queries exceeds
 * one pass (use pcv_searcher_search_device then) — a condition every rank of a sharded search evaluates
 * alike, so all ranks exchange the same payload; a shard that holds none of the selected sources delivers
 * the same layout (empty lists, clear overflow record).  One launch takes any number of segments.
 * Typical step:
 *   begin -> all-gather of (n*k+1)*24 bytes -> pcv_merge_topk_flagged -> end -> repeat if any_overflow. */
pcv_status pcv_searcher_search_device_begin(pcv_searcher* s, const float* queries, int n_queries,
                                            const int64_t* source_ids, int n_sources, int k, void* d_out);
/* `begin` with the queries in DEVICE memory ([n_queries][dim] f32 at d_queries, valid until `end`): embeddings that
 * pcv_model_encode_tokens_device left on the GPU (and an all-gather put together) go into the scan without a host hop —
 * the chain of BASELINE configs[4], model.rs:176 -> search.rs:157 with nothing in between. */
pcv_status pcv_searcher_search_device_begin_dq(pcv_searcher* s, const void* d_queries, int n_queries,
                                               const int64_t* source_ids, int n_sources, int k, void* d_out);
pcv_status pcv_searcher_search_device_end(pcv_searcher* s, int* out_overflowed);
/* How many queries a pass takes is part of a sharded search's protocol (every rank must split a batch alike), so among ranks a pass
 * is what every searcher can take whatever copies it holds: 128 queries.  A host that KNOWS every rank's searcher keeps the int8
 * screening copy of all its rows (pcv_scan_stats.screening_copy == 2 on every rank, e.g. agreed with one all-reduce) may say so on
 * every rank: a pass among ranks then takes what the int8 scan takes — 256 queries up to 384-d, the 256 embeddings of BASELINE
 * configs[4] in one pass instead of two.  A rank for which it is not true fails pcv_searcher_search_device_begin* /
 * pcv_searcher_search_sharded* with PCV_ERR_UNSUPPORTED before queueing anything (the other ranks' exchange would wait for it:
 * check first). */
pcv_status pcv_searcher_allow_wide_sharded_pass(pcv_searcher* s, int on);
/* A step of a sharded search is about to be repeated because SOME rank's pass was incomplete (any_overflow of
 * pcv_merge_topk_flagged): the repeat on THIS rank runs without a speculative start threshold as well.  Every rank
 * keeps its own guess statistics; without this call guesses could fail on different ranks in different attempts and
 * every such failure would make all ranks repeat.  Call it on every rank between `end` and the next `begin`. */
pcv_status pcv_searcher_repeat_without_guess(pcv_searcher* s);

/* Cross-shard merge (replaces the rayon flat_map + sort + truncate of search.rs:163-181):
 * `d_lists` is a DEVICE pointer to [n_shards][n_queries][k] pcv_hit (the all-gather result),
 * written to host arrays shaped like pcv_searcher_search's outputs. */
pcv_status pcv_merge_topk(pcv_ctx* ctx, int metric, int dim, const void* d_lists, int n_shards, int n_queries,
                          int k, int64_t* out_ids, float* out_scores, int* out_counts);

/* pcv_merge_topk for lists produced by pcv_searcher_search_device_begin: shards are n_queries*k+1
 * records apart; *out_any_overflow = OR of the shards' overflow records (identical on every rank). */
pcv_status pcv_merge_topk_flagged(pcv_ctx* ctx, int metric, int dim, const void* d_lists, int n_shards,
                                  int n_queries, int k, int64_t* out_ids, float* out_scores, int* out_counts,
                                  int* out_any_overflow);

/* The same merge on host memory (`lists` = host pointer, same shape): the reference's own merge is
 * host code (search.rs:179-180 sort + truncate); used when the lists were gathered on the host and
 * by the CPU-side (gloo) tests of the multi-GPU protocol.  Needs no GPU. */
pcv_status pcv_merge_topk_host(int metric, int dim, const pcv_hit* lists, int n_shards, int n_queries, int k,
                               int64_t* out_ids, float* out_scores, int* out_counts);

/* ---- native RCCL exchange (no PyTorch in the data path) ----------------------------------------
 * One communicator per process/GPU.  RCCL (librccl.so.1) is loaded on first use; PCV_ERR_UNSUPPORTED if
 * it is not installed.  Bootstrap: rank 0 calls pcv_comm_unique_id and hands the 128 bytes to every rank
 * by whatever channel the host has (torchrun's store, MPI, a file); all ranks then call pcv_comm_create. */
typedef struct pcv_comm pcv_comm;
pcv_status pcv_comm_unique_id(uint8_t out_id[128]);
pcv_status pcv_comm_create(pcv_ctx* ctx, int world_size, int rank, const uint8_t id[128], pcv_comm** out);
pcv_status pcv_comm_destroy(pcv_comm* c);
/* Sharded Searcher::search_vector: local exact top-k on this rank's shard, ncclAllGather of the
 * [n_queries][k] pcv_hit lists over xGMI (on the context stream), merge on every rank.  Collective: every
 * rank of the communicator must call it with the same queries / k.  Outputs as pcv_searcher_search. */
pcv_status pcv_searcher_search_sharded(pcv_searcher* s, pcv_comm* c, const float* queries, int n_queries,
                                       const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                                       float* out_scores, int* out_counts);
/* The same with the queries in DEVICE memory (the same on every rank), and the all-gather that puts them there: every rank
 * contributes bytes_per_rank bytes at d_send and receives world_size x bytes_per_rank at d_recv (rank order; d_send may be
 * this rank's slot of d_recv), on the context stream — data-parallel pcv_model_encode_tokens_device output -> all ranks hold
 * all embeddings -> pcv_searcher_search_sharded_dq, nothing passing through host memory. */
pcv_status pcv_searcher_search_sharded_dq(pcv_searcher* s, pcv_comm* c, const void* d_queries, int n_queries,
                                          const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                                          float* out_scores, int* out_counts);
pcv_status pcv_comm_all_gather(pcv_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank);

/* Brute-force similarity matrices of lib.rs:63-77 for small inputs (tests, highlight.rs:109):
 *   out[b][n] = dot(a_b, m_n)                       pcv_dot_product            (lib.rs:63-65)
 *   out[b][n] = cos(a_b, m_n)                       pcv_cosine_similarity      (lib.rs:67-77)
 * a: [B][dim], m: [N][dim], out: [B][N], all host f32.  Computed on the GPU in f32. */
pcv_status pcv_dot_product(pcv_ctx* ctx, const float* a, int B, const float* m, int64_t N, int dim,
                           float* out);
pcv_status pcv_cosine_similarity(pcv_ctx* ctx, const float* a, int B, const float* m, int64_t N, int dim,
                                 float* out);

/* Counters of the most recent search on this handle (diagnostics + bench roofline). */
typedef struct pcv_scan_stats {
    int64_t rows_scanned;        /* rows streamed by the scan kernel(s), padding excluded        */
    int64_t bytes_algorithmic;   /* rows_scanned * dim * 4                                       */
    float scan_ms;               /* hipEvent time of the scan kernel launches only (a pass over <= 4M rows replayed as a
                                    hipGraph is timed as a whole: then total_ms times the share last measured) */
    float total_ms;              /* hipEvent time of the whole device pipeline                   */
    int64_t candidates;          /* rows rescored exactly, summed over queries                   */
    int32_t scan_launches;       /* scan kernel launches (reruns after overflow included)        */
    int32_t overflow_reruns;     /* passes repeated because a candidate list overflowed          */
    int32_t kernel_used;         /* PCV_KERNEL_WAVE or PCV_KERNEL_MFMA                           */
    int32_t screening_copy;      /* what the scan streamed (last pass): 0 f32 rows, 1 bf16 copy, 2 int8 copy */
    float host_enqueue_ms;       /* host time spent queueing the passes (copies + launches)      */
    float host_wait_ms;          /* host time blocked until the stream had drained               */
    int64_t bytes_streamed;      /* bytes the scan kernel(s) had to read from HBM, layout padding included: per 32-row block
                                    the f32 pieces + 32 row scales, or the bf16 pieces, or the int8 pieces + the block's scale */
    int32_t speculation_reruns;  /* passes repeated because a speculative start threshold (a guess taken from the seed
                                    rows and checked at the end of the pass) did not hold; results are exact either way */
    int32_t mid_copy;            /* 1 if the last pass had the mid copy of every selected segment (pcv_searcher_set_mid_copy) */
    int64_t coarse_survivors;    /* MFMA scans: (row, query) pairs that passed the coarse screen and had their f32 row read
                                    by the fine screen, summed over queries and launches */
    int64_t mid_survivors;       /* ... and those of them that also passed the mid screen (= f32 rows read), when a mid copy exists */
} pcv_scan_stats;
pcv_status pcv_searcher_last_stats(pcv_searcher* s, pcv_scan_stats* out);

/* ---- Model (model.rs:56-191, model/worker.rs:78-106) ---------------------------------------- */

/* Transformer description: what rust-bert reads from config.json / modules.json /
 * 1_Pooling/config.json (model.rs:84-151). */
typedef struct pcv_model_desc {
    int32_t vocab_size;          /* 30522 for all-MiniLM-L6-v2                                  */
    int32_t hidden;              /* 384                                                         */
    int32_t layers;              /* 6                                                           */
    int32_t heads;               /* 12                                                          */
    int32_t intermediate;        /* 1536                                                        */
    int32_t max_positions;       /* 512                                                         */
    int32_t type_vocab;          /* 2                                                           */
    float layer_norm_eps;        /* 1e-12                                                       */
    int32_t pooling;             /* PCV_POOL_*                                                  */
    int32_t normalize;           /* modules.has_normalization(), model.rs:151                   */
    int32_t dense_out;           /* 0 = no Dense module; else output width (model.rs:139-149)   */
    int32_t dense_activation;    /* PCV_ACT_* of the Dense module                               */
    int32_t max_seq_length;      /* sentence_bert_config.max_seq_length (tokenize.rs:66)        */
    int32_t compute;             /* PCV_COMPUTE_*                                               */
    /* ALBERT (ParaphraseAlbertSmallV2, configs.rs:35); all three 0 for the BERT-family models */
    int32_t embedding_size;      /* width of the embedding tables when they are factorised (0 = hidden): a Linear
                                    embedding_size -> hidden follows the embedding LayerNorm       */
    int32_t shared_layers;       /* != 0: every layer runs with the weights of layer 0              */
    int32_t hidden_act;          /* PCV_GELU_ERF (BERT's "gelu") or PCV_GELU_TANH ("gelu_new")      */
} pcv_model_desc;
enum { PCV_GELU_ERF = 0, PCV_GELU_TANH = 1 };
enum { PCV_POOL_MEAN = 0, PCV_POOL_CLS = 1, PCV_POOL_MAX = 2, PCV_POOL_MEAN_SQRT_LEN = 3 };
enum { PCV_ACT_IDENTITY = 0, PCV_ACT_TANH = 1 };
/* F32   : every GEMM on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), the reference's dtype.
 * BF16X3: each f32 operand is split into three bf16 terms (hi + mid + lo = 24 significand bits) and
 *         a product is six bf16 MFMAs accumulated in f32 (the lo*mid, mid*lo, lo*lo terms, < 2^-24
 *         relative, are dropped): f32-level accuracy at 6/16 of the f32-MFMA cost.
 * F16X2 : each f32 operand is split into two f16 terms (11 + 11 significand bits, |x - hi - lo| <= 2^-24 |x|)
 *         and a product is three f16 MFMAs (lo*lo dropped): the same accuracy at half the matrix time of
 *         BF16X3.  Operands are rescaled by exact powers of two so the low terms stay normal; it needs
 *         |activation| < 4094 and |weight| < 255 (checked for weights when the model is built). */
enum { PCV_COMPUTE_F32 = 0, PCV_COMPUTE_BF16X3 = 1, PCV_COMPUTE_F16X2 = 2 };

/* Fill `d` with the all-MiniLM-L6-v2 shape. */
void pcv_model_desc_minilm_l6(pcv_model_desc* d);

/* Model::new_pretrained, transformer part (model.rs:117-151).  `weights_path` is a flat weight
 * file in this library's own format; NULL = seeded synthetic weights
 * (synth_weight(seed, tensor, index)), the only form available offline. */
pcv_status pcv_model_create(pcv_ctx* ctx, const pcv_model_desc* desc, const char* weights_path,
                            uint64_t synthetic_seed, pcv_model** out);
pcv_status pcv_model_destroy(pcv_model* m);
pcv_status pcv_model_output_dim(pcv_model* m, int* out_dim);
/* Overwrite one named tensor from host f32 data (tests load oracle weights through this). */
pcv_status pcv_model_set_tensor(pcv_model* m, const char* name, const float* data, int64_t n);
/* Copy one named tensor to the host (for oracles that must share the synthetic weights). */
pcv_status pcv_model_get_tensor(pcv_model* m, const char* name, float* out, int64_t cap, int64_t* out_n);

/* WorkerData::encode_tokens (worker.rs:78-106): ids/mask are [B][L] int64 row-major exactly as
 * generate_token_tensors lays them out (tokenize.rs:13-51: right-padded, mask = id != pad).
 * out: [B][output_dim] f32. */
pcv_status pcv_model_encode_tokens(pcv_model* m, const int64_t* ids, const int64_t* mask, int B, int L,
                                   float* out);
/* Same with the output left on the device (DEVICE pointer d_out, [B][output_dim] f32). */
pcv_status pcv_model_encode_tokens_device(pcv_model* m, const int64_t* ids, const int64_t* mask, int B,
                                          int L, void* d_out, int async);
/* Per-layer hidden states of the last encode (test hook): layer 0 = embedding output,
 * 1..layers = encoder layer outputs, [B][L][hidden] f32. */
pcv_status pcv_model_debug_hidden(pcv_model* m, int layer, float* out, int64_t cap);

typedef struct pcv_encode_stats {
    float total_ms;      /* hipEvent time of the whole forward                                  */
    double flops;        /* algorithmic FLOPs of the forward (SURVEY.md §8 row D formula)       */
    int32_t batch, seq_len;
} pcv_encode_stats;
pcv_status pcv_model_last_stats(pcv_model* m, pcv_encode_stats* out);

/* ---- Model, text side (model.rs:68-179, model/highlight.rs) -------------------------------------------
 * The host logic that sits between text and the device forward — model directories, tokenisation of a
 * batch, chunk planning and offset mapping of highlight — in the library, so that a Rust / C++ host gets
 * the whole of Model::{new_pretrained, encode, highlight} from this ABI. */

/* Directory name of a SentenceEmbeddingsModelType variant under model_data/ (configs.rs:30-69,121-141):
 * 0 AllMiniLmL6V2 .. 7 MsMarcoBertBaseDotV5, the enum order, which is also model_id() (configs.rs:72-83).
 * NULL for an unknown value. */
const char* pcv_model_type_dir_name(int model_type);

/* Model::new_pretrained (model.rs:68-174) from a sentence-transformers model directory: modules.json,
 * config.json, sentence_bert_config.json, tokenizer_config.json, vocab.txt (or vocab.json + merges.txt, or spiece.model),
 * <n>_Pooling/config.json, optional <n>_Dense/{config.json, weights}.  BERT, DistilBERT, RoBERTa and ALBERT
 * (spiece.model; one layer group) transformers: all eight variants of configs.rs:30-39.  The model owns its tokenizer.
 *   compute       PCV_COMPUTE_*
 *   load_weights  != 0: read the weights file of the directory — rust_model.ot (what the reference loads,
 *                 configs.rs:109,112), else model.safetensors, else pytorch_model.bin; see pcv_checkpoint_visit;
 *                 0: leave the weights to the caller — hand every checkpoint tensor to pcv_model_load_hf_tensor,
 *                 then pcv_model_check_loaded. */
pcv_status pcv_model_create_from_dir(pcv_ctx* ctx, const char* model_dir, int compute, int load_weights, pcv_model** out);
/* What pcv_model_create_from_dir would build, from the directory's JSON files alone (needs no GPU): the model
 * description, the transformer family (0 BERT, 1 DistilBERT, 2 RoBERTa, 3 ALBERT) and the tokenizer options
 * (strip_accents: -1 = follow lower_case).  Any output pointer may be NULL. */
pcv_status pcv_model_dir_describe(const char* model_dir, pcv_model_desc* out_desc, int* out_arch, int* out_lower_case,
                                  int* out_strip_accents);
/* Walk a checkpoint file: model.safetensors, or a libtorch zip archive — rust_model.ot as tch's
 * Tensor::save_multi writes it (VarStore::load, model.rs:117-124) or a torch.save state dict (pytorch_model.bin).
 * The archive's pickle is interpreted by a closed machine that knows only the tensor / dict / module records those
 * writers emit; nothing from the file is executed, an unknown record is PCV_ERR_UNSUPPORTED.
 * `visit` is called once per tensor in file order: shape[rank], dtype PCV_TENSOR_*, values = numel row-major f32
 * converted from the stored dtype (NULL for PCV_TENSOR_OTHER: integer tensors such as position_ids).  A non-zero
 * return from `visit` stops the walk with PCV_ERR_INVALID.  Needs no GPU. */
enum { PCV_TENSOR_F32 = 0, PCV_TENSOR_F16 = 1, PCV_TENSOR_BF16 = 2, PCV_TENSOR_F64 = 3, PCV_TENSOR_OTHER = 4 };
typedef int (*pcv_tensor_visitor)(void* user, const char* name, const int64_t* shape, int rank, int dtype, const float* values,
                                  int64_t numel);
pcv_status pcv_checkpoint_visit(const char* path, pcv_tensor_visitor visit, void* user);
/* One checkpoint tensor under its Hugging Face / rust-bert name ("bert.encoder.layer.0...", DistilBERT and
 * RoBERTa names included: they are mapped onto the encoder graph, the RoBERTa position table is shifted);
 * names the graph does not use are ignored.  `data`: numel f32, row-major. */
pcv_status pcv_model_load_hf_tensor(pcv_model* m, const char* hf_name, const float* data, int64_t numel);
/* PCV_ERR_IO naming a missing tensor unless every tensor of the graph has been provided. */
pcv_status pcv_model_check_loaded(pcv_model* m);
/* Give a model built by pcv_model_create its tokenizer (Model::tokenizer, model.rs:61).  take_ownership != 0:
 * the model destroys it. */
pcv_status pcv_model_set_tokenizer(pcv_model* m, pcv_tokenizer* t, int take_ownership);
/* The model's tokenizer (NULL if none); still owned as before. */
pcv_status pcv_model_tokenizer(pcv_model* m, pcv_tokenizer** out_tok);
/* The description the model was built from and the pad id its token tensors use (tokenize.rs:19). */
pcv_status pcv_model_get_desc(pcv_model* m, pcv_model_desc* out_desc, int64_t* out_pad_id);

/* Model::encode(&[S]) (model.rs:176-179): tokenize (encode_list, max_seq_length, LongestFirst; host threads),
 * pad to the batch maximum, forward.  texts[i] = n_bytes[i] bytes of UTF-8.  out: [n_texts][output_dim] f32. */
pcv_status pcv_model_encode_text(pcv_model* m, const char* const* texts, const size_t* n_bytes, int n_texts, float* out);

/* Model::highlight (model/highlight.rs:23-165): for every document the text of the chunk whose embedding has
 * the largest dot product with the query's.  Documents are tokenized without truncation, cut into chunks of
 * `chunk_size` tokens overlapping by `chunk_overlap` (<= 0 / < 0: the CHUNK_SIZE / CHUNK_OVERLAP environment
 * variables, defaults 20 / 4, highlight.rs:7-18), all chunks are encoded in batches and scored on the device.
 *   out_begin/out_end [n_docs]  byte range of the highlight inside docs[i];
 *                               -1/-1 = None (no chunk: the document is too short), begin == end = Some(""). */
pcv_status pcv_model_highlight(pcv_model* m, const char* query, size_t query_bytes, const char* const* docs, const size_t* doc_bytes,
                               int n_docs, int chunk_size, int chunk_overlap, int64_t* out_begin, int64_t* out_end);

/* ---- Tokenizer (model/tokenize.rs:60-77; rust_tokenizers BertTokenizer) -------------------------
 * Host code, like the reference's tokenizer: BERT BasicTokenizer (clean text, CJK spacing, lower-casing,
 * accent stripping, punctuation split) + greedy WordPiece over `vocab.txt` (one token per line, id =
 * line number).  Needs no GPU. */
/* TokenizerOption::from_file (model.rs:96-113).  strip_accents < 0: follow lower_case (the default of
 * rust_tokenizers / HF when tokenizer_config.strip_accents is absent). */
pcv_status pcv_tokenizer_create(const char* vocab_path, int lower_case, int strip_accents, pcv_tokenizer** out);
/* Byte-level BPE tokenizer of the RoBERTa-family models in the reference's list (AllDistilrobertaV1:
 * `RobertaTokenizer::from_file(vocab.json, merges.txt, lower_case, add_prefix_space)` in rust_tokenizers): GPT-2
 * pre-tokenization, bytes_to_unicode symbols, ranked merges, <s> ... </s> framing.  The handle works with
 * pcv_tokenizer_encode / _encode_batch / _special_ids (pad = <pad>, cls = <s>, sep = </s>). */
pcv_status pcv_tokenizer_create_bpe(const char* vocab_json_path, const char* merges_path, int add_prefix_space,
                                    pcv_tokenizer** out);
/* AlbertTokenizer::from_file(spiece.model, lower_case, strip_accents) of rust_tokenizers (what rust-bert builds for
 * ModelType::Albert, the ParaphraseAlbertSmallV2 variant of configs.rs:35): a SentencePiece unigram model file.
 * Text is cleaned, NFKC-normalised, lower-cased / stripped of accents as asked, whitespace becomes U+2581 and the
 * best-scoring segmentation is found by Viterbi; ALBERT's "<digit>," pieces are split again.  Framing is
 * [CLS] ... [SEP], padding <pad>.  strip_accents < 0 = follow lower_case. */
pcv_status pcv_tokenizer_create_sentencepiece(const char* model_path, int lower_case, int strip_accents, pcv_tokenizer** out);
/* Unicode NFKC (Unicode 13 tables) of UTF-8 `text`: what the SentencePiece path normalises with.  *out_n = bytes
 * of the result; out may be NULL to ask for the size. */
pcv_status pcv_unicode_nfkc(const char* text, size_t n_bytes, char* out, size_t cap, size_t* out_n);
pcv_status pcv_tokenizer_destroy(pcv_tokenizer* t);
pcv_status pcv_tokenizer_vocab_size(pcv_tokenizer* t, int* out_n);
/* ids of [PAD] (get_pad_id, tokenize.rs:19), [UNK], [CLS], [SEP]; -1 when the vocab lacks one */
pcv_status pcv_tokenizer_special_ids(pcv_tokenizer* t, int64_t* pad, int64_t* unk, int64_t* cls, int64_t* sep);

/* One element of encode_list(inputs, max_len, TruncationStrategy::LongestFirst, stride 0)
 * (tokenize.rs:64-75): [CLS] pieces... [SEP], truncated to max_len tokens in all.
 *   out_ids           token ids
 *   out_begin/out_end char (Unicode scalar) offsets of each token in `text`, -1 for special tokens
 *                     (TokenIdsWithOffsets::token_offsets, used by highlight.rs:129-147); may be NULL
 *   out_special       special_tokens_mask (highlight.rs:58-84); may be NULL
 *   cap               capacity of the output arrays; out_len receives the token count */
pcv_status pcv_tokenizer_encode(pcv_tokenizer* t, const char* text, size_t n_bytes, int max_len, int64_t* out_ids,
                                int32_t* out_begin, int32_t* out_end, uint8_t* out_special, int cap, int* out_len);
/* The batch form behind Model::tokenize (tokenize.rs:60-77) + generate_token_tensors (tokenize.rs:9-57):
 * text i -> row i of out_ids ([n_texts][max_len], right-padded with pad_id), out_lens[i] = token count
 * (truncated to max_len like pcv_tokenizer_encode).  Texts are spread over n_threads host threads
 * (0 = all hardware threads). */
pcv_status pcv_tokenizer_encode_batch(pcv_tokenizer* t, const char* const* texts, const size_t* n_bytes, int n_texts,
                                      int max_len, int64_t pad_id, int64_t* out_ids, int32_t* out_lens, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* PERCEIVE_HIP_H */

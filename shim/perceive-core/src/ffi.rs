//! `extern "C"` binding of libperceive_hip.so — GENERATED from include/perceive_hip.h by
//! tools/gen_rust_ffi.py; do not edit.  One declaration per symbol of the header, same order.
//! NOT COMPILED in the build environment of this repository (it has no Rust toolchain):
//! tests/test_rust_shim.py checks it mechanically against the header instead (symbol set, argument
//! counts, integer widths, struct fields).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

/// opaque handle
#[repr(C)]
pub struct pcv_ctx {
    _private: [u8; 0],
}
/// opaque handle
#[repr(C)]
pub struct pcv_searcher {
    _private: [u8; 0],
}
/// opaque handle
#[repr(C)]
pub struct pcv_model {
    _private: [u8; 0],
}
/// opaque handle
#[repr(C)]
pub struct pcv_tokenizer {
    _private: [u8; 0],
}
/// opaque handle
#[repr(C)]
pub struct pcv_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct pcv_hit {
    pub score: f64,
    pub pos: i64,
    pub id: i64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct pcv_scan_stats {
    pub rows_scanned: i64,
    pub bytes_algorithmic: i64,
    pub scan_ms: f32,
    pub total_ms: f32,
    pub candidates: i64,
    pub scan_launches: i32,
    pub overflow_reruns: i32,
    pub kernel_used: i32,
    pub screening_copy: i32,
    pub host_enqueue_ms: f32,
    pub host_wait_ms: f32,
    pub bytes_streamed: i64,
    pub speculation_reruns: i32,
    pub mid_copy: i32,
    pub coarse_survivors: i64,
    pub mid_survivors: i64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct pcv_model_desc {
    pub vocab_size: i32,
    pub hidden: i32,
    pub layers: i32,
    pub heads: i32,
    pub intermediate: i32,
    pub max_positions: i32,
    pub type_vocab: i32,
    pub layer_norm_eps: f32,
    pub pooling: i32,
    pub normalize: i32,
    pub dense_out: i32,
    pub dense_activation: i32,
    pub max_seq_length: i32,
    pub compute: i32,
    pub embedding_size: i32,
    pub shared_layers: i32,
    pub hidden_act: i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct pcv_encode_stats {
    pub total_ms: f32,
    pub flops: f64,
    pub batch: i32,
    pub seq_len: i32,
}

pub const PCV_OK: c_int = 0;
pub const PCV_ERR_INVALID: c_int = 1;
pub const PCV_ERR_DEVICE: c_int = 2;
pub const PCV_ERR_UNSUPPORTED: c_int = 3;
pub const PCV_ERR_IO: c_int = 4;
pub const PCV_ERR_INTERNAL: c_int = 5;
pub const PCV_METRIC_COSINE: c_int = 0;
pub const PCV_METRIC_DOT: c_int = 1;
pub const PCV_KERNEL_AUTO: c_int = 0;
pub const PCV_KERNEL_WAVE: c_int = 1;
pub const PCV_KERNEL_MFMA: c_int = 2;
pub const PCV_TUNE_FAIL_COPY_ALLOC: c_int = 1073741824;
pub const PCV_SCREEN_COPY_OFF: c_int = 0;
pub const PCV_SCREEN_COPY_BF16: c_int = 1;
pub const PCV_SCREEN_COPY_AUTO: c_int = 2;
pub const PCV_SCREEN_COPY_INT8: c_int = 3;
pub const PCV_MID_COPY_OFF: c_int = 0;
pub const PCV_MID_COPY_AUTO: c_int = 1;
pub const PCV_MID_COPY_ON: c_int = 2;
pub const PCV_MAX_RESULTS: c_int = 128;
pub const PCV_GELU_ERF: c_int = 0;
pub const PCV_GELU_TANH: c_int = 1;
pub const PCV_POOL_MEAN: c_int = 0;
pub const PCV_POOL_CLS: c_int = 1;
pub const PCV_POOL_MAX: c_int = 2;
pub const PCV_POOL_MEAN_SQRT_LEN: c_int = 3;
pub const PCV_ACT_IDENTITY: c_int = 0;
pub const PCV_ACT_TANH: c_int = 1;
pub const PCV_COMPUTE_F32: c_int = 0;
pub const PCV_COMPUTE_BF16X3: c_int = 1;
pub const PCV_COMPUTE_F16X2: c_int = 2;
pub const PCV_TENSOR_F32: c_int = 0;
pub const PCV_TENSOR_F16: c_int = 1;
pub const PCV_TENSOR_BF16: c_int = 2;
pub const PCV_TENSOR_F64: c_int = 3;
pub const PCV_TENSOR_OTHER: c_int = 4;
pub const PCV_STAGING_SOURCE: i64 = i64::MIN;

pub type pcv_tensor_visitor = Option<unsafe extern "C" fn(user: *mut c_void, name: *const c_char, shape: *const i64, rank: c_int, dtype: c_int, values: *const f32, numel: i64) -> c_int>;

#[link(name = "perceive_hip")]
extern "C" {
    pub fn pcv_last_error() -> *const c_char;
    pub fn pcv_version() -> *const c_char;
    pub fn pcv_device_count() -> c_int;
    pub fn pcv_init(device_index: c_int, out_ctx: *mut *mut pcv_ctx) -> c_int;
    pub fn pcv_shutdown(ctx: *mut pcv_ctx) -> c_int;
    pub fn pcv_synchronize(ctx: *mut pcv_ctx) -> c_int;
    pub fn pcv_stream(ctx: *mut pcv_ctx) -> *mut c_void;
    pub fn pcv_set_stream(ctx: *mut pcv_ctx, hip_stream: *mut c_void, adopt: c_int) -> c_int;
    pub fn pcv_device_alloc(ctx: *mut pcv_ctx, n_bytes: usize, out_dptr: *mut *mut c_void) -> c_int;
    pub fn pcv_device_free(ctx: *mut pcv_ctx, dptr: *mut c_void) -> c_int;
    pub fn pcv_copy_to_host(ctx: *mut pcv_ctx, dst_host: *mut c_void, src_dev: *const c_void, n_bytes: usize) -> c_int;
    pub fn pcv_copy_to_device(ctx: *mut pcv_ctx, dst_dev: *mut c_void, src_host: *const c_void, n_bytes: usize) -> c_int;
    pub fn pcv_deserialize_embedding(blob: *const u8, n_bytes: usize, out: *mut f32, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn pcv_serialize_embedding(v: *const f32, n: usize, out: *mut u8, out_cap: usize) -> c_int;
    pub fn pcv_searcher_create(ctx: *mut pcv_ctx, dim: c_int, metric: c_int, out: *mut *mut pcv_searcher) -> c_int;
    pub fn pcv_searcher_destroy(s: *mut pcv_searcher) -> c_int;
    pub fn pcv_searcher_add_rows(s: *mut pcv_searcher, source_id: i64, ids: *const i64, rows: *const f32, n: i64) -> c_int;
    pub fn pcv_searcher_add_blobs(s: *mut pcv_searcher, source_id: i64, ids: *const i64, blobs: *const u8, n: i64) -> c_int;
    pub fn pcv_searcher_add_synthetic(s: *mut pcv_searcher, source_id: i64, n: i64, seed: u64, first_row: i64, normalize: c_int) -> c_int;
    pub fn pcv_searcher_add_synthetic_clustered(s: *mut pcv_searcher, source_id: i64, n: i64, seed: u64, first_row: i64, normalize: c_int, n_clusters: c_int, noise: f32) -> c_int;
    pub fn pcv_searcher_add_synthetic_scaled(s: *mut pcv_searcher, source_id: i64, n: i64, seed: u64, first_row: i64, amp_lo: f32, amp_hi: f32) -> c_int;
    pub fn pcv_searcher_reserve(s: *mut pcv_searcher, source_id: i64, n_rows: i64) -> c_int;
    pub fn pcv_searcher_clear_source(s: *mut pcv_searcher, source_id: i64) -> c_int;
    pub fn pcv_searcher_replace_source(s: *mut pcv_searcher, from_source_id: i64, to_source_id: i64) -> c_int;
    pub fn pcv_searcher_finalize(s: *mut pcv_searcher) -> c_int;
    pub fn pcv_searcher_load_sqlite(s: *mut pcv_searcher, db_path: *const c_char, model_id: u32, model_version: u32, only_source: *const i64, out_rows: *mut i64) -> c_int;
    pub fn pcv_searcher_dim(s: *mut pcv_searcher, out_dim: *mut c_int) -> c_int;
    pub fn pcv_searcher_num_rows(s: *mut pcv_searcher, out_rows: *mut i64) -> c_int;
    pub fn pcv_searcher_num_segments(s: *mut pcv_searcher, out_n: *mut c_int) -> c_int;
    pub fn pcv_searcher_num_sources(s: *mut pcv_searcher, out_n: *mut c_int) -> c_int;
    pub fn pcv_searcher_source_ids(s: *mut pcv_searcher, out_ids: *mut i64, cap: c_int) -> c_int;
    pub fn pcv_searcher_source_num_rows(s: *mut pcv_searcher, source_id: i64, out_rows: *mut i64) -> c_int;
    pub fn pcv_searcher_get_rows(s: *mut pcv_searcher, positions: *const i64, n: i64, out_rows: *mut f32, out_ids: *mut i64) -> c_int;
    pub fn pcv_searcher_set_kernel(s: *mut pcv_searcher, kernel: c_int) -> c_int;
    pub fn pcv_searcher_set_candidate_capacity(s: *mut pcv_searcher, n_candidates: u32) -> c_int;
    pub fn pcv_searcher_set_tuning(s: *mut pcv_searcher, flags: u32) -> c_int;
    pub fn pcv_searcher_set_screening_copy(s: *mut pcv_searcher, mode: c_int) -> c_int;
    pub fn pcv_searcher_set_mid_copy(s: *mut pcv_searcher, mode: c_int) -> c_int;
    pub fn pcv_searcher_wait_background(s: *mut pcv_searcher) -> c_int;
    pub fn pcv_searcher_search(s: *mut pcv_searcher, queries: *const f32, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int) -> c_int;
    pub fn pcv_searcher_set_shard_offset(s: *mut pcv_searcher, first_global_pos: i64) -> c_int;
    pub fn pcv_searcher_search_device(s: *mut pcv_searcher, queries: *const f32, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, d_out: *mut c_void, async_: c_int) -> c_int;
    pub fn pcv_searcher_search_device_begin(s: *mut pcv_searcher, queries: *const f32, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, d_out: *mut c_void) -> c_int;
    pub fn pcv_searcher_search_device_begin_dq(s: *mut pcv_searcher, d_queries: *const c_void, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, d_out: *mut c_void) -> c_int;
    pub fn pcv_searcher_search_device_end(s: *mut pcv_searcher, out_overflowed: *mut c_int) -> c_int;
    pub fn pcv_searcher_allow_wide_sharded_pass(s: *mut pcv_searcher, on: c_int) -> c_int;
    pub fn pcv_searcher_repeat_without_guess(s: *mut pcv_searcher) -> c_int;
    pub fn pcv_merge_topk(ctx: *mut pcv_ctx, metric: c_int, dim: c_int, d_lists: *const c_void, n_shards: c_int, n_queries: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int) -> c_int;
    pub fn pcv_merge_topk_flagged(ctx: *mut pcv_ctx, metric: c_int, dim: c_int, d_lists: *const c_void, n_shards: c_int, n_queries: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int, out_any_overflow: *mut c_int) -> c_int;
    pub fn pcv_merge_topk_host(metric: c_int, dim: c_int, lists: *const pcv_hit, n_shards: c_int, n_queries: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int) -> c_int;
    pub fn pcv_comm_unique_id(out_id: *mut u8) -> c_int;
    pub fn pcv_comm_create(ctx: *mut pcv_ctx, world_size: c_int, rank: c_int, id: *const u8, out: *mut *mut pcv_comm) -> c_int;
    pub fn pcv_comm_destroy(c: *mut pcv_comm) -> c_int;
    pub fn pcv_searcher_search_sharded(s: *mut pcv_searcher, c: *mut pcv_comm, queries: *const f32, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int) -> c_int;
    pub fn pcv_searcher_search_sharded_dq(s: *mut pcv_searcher, c: *mut pcv_comm, d_queries: *const c_void, n_queries: c_int, source_ids: *const i64, n_sources: c_int, k: c_int, out_ids: *mut i64, out_scores: *mut f32, out_counts: *mut c_int) -> c_int;
    pub fn pcv_comm_all_gather(c: *mut pcv_comm, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: usize) -> c_int;
    pub fn pcv_dot_product(ctx: *mut pcv_ctx, a: *const f32, B: c_int, m: *const f32, N: i64, dim: c_int, out: *mut f32) -> c_int;
    pub fn pcv_cosine_similarity(ctx: *mut pcv_ctx, a: *const f32, B: c_int, m: *const f32, N: i64, dim: c_int, out: *mut f32) -> c_int;
    pub fn pcv_searcher_last_stats(s: *mut pcv_searcher, out: *mut pcv_scan_stats) -> c_int;
    pub fn pcv_model_desc_minilm_l6(d: *mut pcv_model_desc);
    pub fn pcv_model_create(ctx: *mut pcv_ctx, desc: *const pcv_model_desc, weights_path: *const c_char, synthetic_seed: u64, out: *mut *mut pcv_model) -> c_int;
    pub fn pcv_model_destroy(m: *mut pcv_model) -> c_int;
    pub fn pcv_model_output_dim(m: *mut pcv_model, out_dim: *mut c_int) -> c_int;
    pub fn pcv_model_set_tensor(m: *mut pcv_model, name: *const c_char, data: *const f32, n: i64) -> c_int;
    pub fn pcv_model_get_tensor(m: *mut pcv_model, name: *const c_char, out: *mut f32, cap: i64, out_n: *mut i64) -> c_int;
    pub fn pcv_model_encode_tokens(m: *mut pcv_model, ids: *const i64, mask: *const i64, B: c_int, L: c_int, out: *mut f32) -> c_int;
    pub fn pcv_model_encode_tokens_device(m: *mut pcv_model, ids: *const i64, mask: *const i64, B: c_int, L: c_int, d_out: *mut c_void, async_: c_int) -> c_int;
    pub fn pcv_model_debug_hidden(m: *mut pcv_model, layer: c_int, out: *mut f32, cap: i64) -> c_int;
    pub fn pcv_model_last_stats(m: *mut pcv_model, out: *mut pcv_encode_stats) -> c_int;
    pub fn pcv_model_type_dir_name(model_type: c_int) -> *const c_char;
    pub fn pcv_model_create_from_dir(ctx: *mut pcv_ctx, model_dir: *const c_char, compute: c_int, load_weights: c_int, out: *mut *mut pcv_model) -> c_int;
    pub fn pcv_model_dir_describe(model_dir: *const c_char, out_desc: *mut pcv_model_desc, out_arch: *mut c_int, out_lower_case: *mut c_int, out_strip_accents: *mut c_int) -> c_int;
    pub fn pcv_checkpoint_visit(path: *const c_char, visit: pcv_tensor_visitor, user: *mut c_void) -> c_int;
    pub fn pcv_model_load_hf_tensor(m: *mut pcv_model, hf_name: *const c_char, data: *const f32, numel: i64) -> c_int;
    pub fn pcv_model_check_loaded(m: *mut pcv_model) -> c_int;
    pub fn pcv_model_set_tokenizer(m: *mut pcv_model, t: *mut pcv_tokenizer, take_ownership: c_int) -> c_int;
    pub fn pcv_model_tokenizer(m: *mut pcv_model, out_tok: *mut *mut pcv_tokenizer) -> c_int;
    pub fn pcv_model_get_desc(m: *mut pcv_model, out_desc: *mut pcv_model_desc, out_pad_id: *mut i64) -> c_int;
    pub fn pcv_model_encode_text(m: *mut pcv_model, texts: *const *const c_char, n_bytes: *const usize, n_texts: c_int, out: *mut f32) -> c_int;
    pub fn pcv_model_highlight(m: *mut pcv_model, query: *const c_char, query_bytes: usize, docs: *const *const c_char, doc_bytes: *const usize, n_docs: c_int, chunk_size: c_int, chunk_overlap: c_int, out_begin: *mut i64, out_end: *mut i64) -> c_int;
    pub fn pcv_tokenizer_create(vocab_path: *const c_char, lower_case: c_int, strip_accents: c_int, out: *mut *mut pcv_tokenizer) -> c_int;
    pub fn pcv_tokenizer_create_bpe(vocab_json_path: *const c_char, merges_path: *const c_char, add_prefix_space: c_int, out: *mut *mut pcv_tokenizer) -> c_int;
    pub fn pcv_tokenizer_create_sentencepiece(model_path: *const c_char, lower_case: c_int, strip_accents: c_int, out: *mut *mut pcv_tokenizer) -> c_int;
    pub fn pcv_unicode_nfkc(text: *const c_char, n_bytes: usize, out: *mut c_char, cap: usize, out_n: *mut usize) -> c_int;
    pub fn pcv_tokenizer_destroy(t: *mut pcv_tokenizer) -> c_int;
    pub fn pcv_tokenizer_vocab_size(t: *mut pcv_tokenizer, out_n: *mut c_int) -> c_int;
    pub fn pcv_tokenizer_special_ids(t: *mut pcv_tokenizer, pad: *mut i64, unk: *mut i64, cls: *mut i64, sep: *mut i64) -> c_int;
    pub fn pcv_tokenizer_encode(t: *mut pcv_tokenizer, text: *const c_char, n_bytes: usize, max_len: c_int, out_ids: *mut i64, out_begin: *mut i32, out_end: *mut i32, out_special: *mut u8, cap: c_int, out_len: *mut c_int) -> c_int;
    pub fn pcv_tokenizer_encode_batch(t: *mut pcv_tokenizer, texts: *const *const c_char, n_bytes: *const usize, n_texts: c_int, max_len: c_int, pad_id: i64, out_ids: *mut i64, out_lens: *mut i32, n_threads: c_int) -> c_int;
}

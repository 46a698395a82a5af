"""ctypes binding of libperceive_hip.so (the C ABI declared in include/perceive_hip.h).

The library is the product: there is no Python/CPU fallback.  Loading fails loudly when the
in-tree build is missing (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libperceive_hip.so")

PCV_OK = 0
METRIC_COSINE, METRIC_DOT = 0, 1
KERNEL_AUTO, KERNEL_WAVE, KERNEL_MFMA = 0, 1, 2
POOL_MEAN, POOL_CLS, POOL_MAX, POOL_MEAN_SQRT_LEN = 0, 1, 2, 3
ACT_IDENTITY, ACT_TANH = 0, 1
COMPUTE_F32, COMPUTE_BF16X3, COMPUTE_F16X2 = 0, 1, 2


class PcvError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"perceive_hip status {status}: {message}")
        self.status = status


class Hit(C.Structure):
    _fields_ = [("score", C.c_double), ("pos", C.c_int64), ("id", C.c_int64)]


class ScanStats(C.Structure):
    _fields_ = [
        ("rows_scanned", C.c_int64),
        ("bytes_algorithmic", C.c_int64),
        ("scan_ms", C.c_float),
        ("total_ms", C.c_float),
        ("candidates", C.c_int64),
        ("scan_launches", C.c_int32),
        ("overflow_reruns", C.c_int32),
        ("kernel_used", C.c_int32),
        ("screening_copy", C.c_int32),
        ("host_enqueue_ms", C.c_float),
        ("host_wait_ms", C.c_float),
        ("bytes_streamed", C.c_int64),
        ("speculation_reruns", C.c_int32),
        ("mid_copy", C.c_int32),
        ("coarse_survivors", C.c_int64),
        ("mid_survivors", C.c_int64),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden", C.c_int32),
        ("layers", C.c_int32),
        ("heads", C.c_int32),
        ("intermediate", C.c_int32),
        ("max_positions", C.c_int32),
        ("type_vocab", C.c_int32),
        ("layer_norm_eps", C.c_float),
        ("pooling", C.c_int32),
        ("normalize", C.c_int32),
        ("dense_out", C.c_int32),
        ("dense_activation", C.c_int32),
        ("max_seq_length", C.c_int32),
        ("compute", C.c_int32),
        ("embedding_size", C.c_int32),
        ("shared_layers", C.c_int32),
        ("hidden_act", C.c_int32),
    ]


class EncodeStats(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("flops", C.c_double), ("batch", C.c_int32), ("seq_len", C.c_int32)]


# name -> (restype, argtypes); every symbol include/perceive_hip.h declares
# Data pointers are declared void* and passed as plain addresses (`ndarray.ctypes.data`): building typed
# ctypes pointers with `ndarray.ctypes.data_as` costs 2 us each and ~20-40 us each once torch has been
# imported into the process, which was 150 us per search call.  The names keep the C types readable.
_P = C.c_void_p
_I64P = C.c_void_p  # int64_t*
_F32P = C.c_void_p  # float*
_U8P = C.c_void_p  # uint8_t*
_INTP = C.c_void_p  # int* / int32_t*
# int visit(void* user, const char* name, const int64_t* shape, int rank, int dtype, const float* values, int64_t numel)
TENSOR_VISITOR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64)
SYMBOLS = {
    "pcv_last_error": (C.c_char_p, []),
    "pcv_version": (C.c_char_p, []),
    "pcv_device_count": (C.c_int, []),
    "pcv_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pcv_shutdown": (C.c_int, [_P]),
    "pcv_synchronize": (C.c_int, [_P]),
    "pcv_stream": (_P, [_P]),
    "pcv_set_stream": (C.c_int, [_P, _P, C.c_int]),
    "pcv_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "pcv_device_free": (C.c_int, [_P, _P]),
    "pcv_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pcv_copy_to_device": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pcv_deserialize_embedding": (C.c_int, [_U8P, C.c_size_t, _F32P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pcv_serialize_embedding": (C.c_int, [_F32P, C.c_size_t, _U8P, C.c_size_t]),
    "pcv_searcher_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pcv_searcher_destroy": (C.c_int, [_P]),
    "pcv_searcher_reserve": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "pcv_searcher_num_segments": (C.c_int, [_P, _INTP]),
    "pcv_searcher_add_rows": (C.c_int, [_P, C.c_int64, _I64P, _F32P, C.c_int64]),
    "pcv_searcher_add_blobs": (C.c_int, [_P, C.c_int64, _I64P, _U8P, C.c_int64]),
    "pcv_searcher_add_synthetic": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int]),
    "pcv_searcher_add_synthetic_clustered": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_float]),
    "pcv_searcher_add_synthetic_scaled": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_float, C.c_float]),
    "pcv_searcher_clear_source": (C.c_int, [_P, C.c_int64]),
    "pcv_searcher_replace_source": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "pcv_searcher_finalize": (C.c_int, [_P]),
    "pcv_searcher_load_sqlite": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint32, _I64P, _I64P]),
    "pcv_searcher_dim": (C.c_int, [_P, _INTP]),
    "pcv_searcher_num_rows": (C.c_int, [_P, _I64P]),
    "pcv_searcher_num_sources": (C.c_int, [_P, _INTP]),
    "pcv_searcher_source_ids": (C.c_int, [_P, _I64P, C.c_int]),
    "pcv_searcher_source_num_rows": (C.c_int, [_P, C.c_int64, _I64P]),
    "pcv_searcher_get_rows": (C.c_int, [_P, _I64P, C.c_int64, _F32P, _I64P]),
    "pcv_searcher_set_kernel": (C.c_int, [_P, C.c_int]),
    "pcv_searcher_set_screening_copy": (C.c_int, [_P, C.c_int]),
    "pcv_searcher_set_mid_copy": (C.c_int, [_P, C.c_int]),
    "pcv_searcher_wait_background": (C.c_int, [_P]),
    "pcv_searcher_allow_wide_sharded_pass": (C.c_int, [_P, C.c_int]),
    "pcv_searcher_set_candidate_capacity": (C.c_int, [_P, C.c_uint32]),
    "pcv_searcher_set_tuning": (C.c_int, [_P, C.c_uint32]),
    "pcv_searcher_search": (C.c_int, [_P, _F32P, C.c_int, _I64P, C.c_int, C.c_int, _I64P, _F32P, _INTP]),
    "pcv_searcher_set_shard_offset": (C.c_int, [_P, C.c_int64]),
    "pcv_searcher_search_device": (C.c_int, [_P, _F32P, C.c_int, _I64P, C.c_int, C.c_int, _P, C.c_int]),
    "pcv_searcher_search_device_begin": (C.c_int, [_P, _F32P, C.c_int, _I64P, C.c_int, C.c_int, _P]),
    "pcv_searcher_search_device_begin_dq": (C.c_int, [_P, _P, C.c_int, _I64P, C.c_int, C.c_int, _P]),
    "pcv_searcher_search_device_end": (C.c_int, [_P, _INTP]),
    "pcv_searcher_repeat_without_guess": (C.c_int, [_P]),
    "pcv_merge_topk_flagged": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _I64P, _F32P, _INTP, _INTP]),
    "pcv_merge_topk": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _I64P, _F32P, _INTP]),
    "pcv_merge_topk_host": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _I64P, _F32P, _INTP]),
    "pcv_comm_unique_id": (C.c_int, [_U8P]),
    "pcv_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _U8P, C.POINTER(_P)]),
    "pcv_comm_destroy": (C.c_int, [_P]),
    "pcv_searcher_search_sharded": (C.c_int, [_P, _P, _F32P, C.c_int, _I64P, C.c_int, C.c_int, _I64P, _F32P, _INTP]),
    "pcv_searcher_search_sharded_dq": (C.c_int, [_P, _P, _P, C.c_int, _I64P, C.c_int, C.c_int, _I64P, _F32P, _INTP]),
    "pcv_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pcv_dot_product": (C.c_int, [_P, _F32P, C.c_int, _F32P, C.c_int64, C.c_int, _F32P]),
    "pcv_cosine_similarity": (C.c_int, [_P, _F32P, C.c_int, _F32P, C.c_int64, C.c_int, _F32P]),
    "pcv_searcher_last_stats": (C.c_int, [_P, C.POINTER(ScanStats)]),
    "pcv_model_desc_minilm_l6": (None, [C.POINTER(ModelDesc)]),
    "pcv_model_create": (C.c_int, [_P, C.POINTER(ModelDesc), C.c_char_p, C.c_uint64, C.POINTER(_P)]),
    "pcv_model_destroy": (C.c_int, [_P]),
    "pcv_model_output_dim": (C.c_int, [_P, _INTP]),
    "pcv_model_set_tensor": (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    "pcv_model_get_tensor": (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64, _I64P]),
    "pcv_model_encode_tokens": (C.c_int, [_P, _I64P, _I64P, C.c_int, C.c_int, _F32P]),
    "pcv_model_encode_tokens_device": (C.c_int, [_P, _I64P, _I64P, C.c_int, C.c_int, _P, C.c_int]),
    "pcv_model_debug_hidden": (C.c_int, [_P, C.c_int, _F32P, C.c_int64]),
    "pcv_model_last_stats": (C.c_int, [_P, C.POINTER(EncodeStats)]),
    "pcv_model_type_dir_name": (C.c_char_p, [C.c_int]),
    "pcv_model_create_from_dir": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "pcv_model_dir_describe": (C.c_int, [C.c_char_p, C.POINTER(ModelDesc), _INTP, _INTP, _INTP]),
    "pcv_checkpoint_visit": (C.c_int, [C.c_char_p, TENSOR_VISITOR, _P]),
    "pcv_model_load_hf_tensor": (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    "pcv_model_check_loaded": (C.c_int, [_P]),
    "pcv_model_set_tokenizer": (C.c_int, [_P, _P, C.c_int]),
    "pcv_model_tokenizer": (C.c_int, [_P, C.POINTER(_P)]),
    "pcv_model_get_desc": (C.c_int, [_P, C.POINTER(ModelDesc), _I64P]),
    "pcv_model_encode_text": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, _F32P]),
    "pcv_model_highlight": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                      C.c_int, _I64P, _I64P]),
    "pcv_tokenizer_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "pcv_tokenizer_create_bpe": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_P)]),
    "pcv_tokenizer_create_sentencepiece": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "pcv_unicode_nfkc": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pcv_tokenizer_destroy": (C.c_int, [_P]),
    "pcv_tokenizer_vocab_size": (C.c_int, [_P, _INTP]),
    "pcv_tokenizer_special_ids": (C.c_int, [_P, _I64P, _I64P, _I64P, _I64P]),
    "pcv_tokenizer_encode_batch": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int64,
                                             _I64P, _INTP, C.c_int]),
    "pcv_tokenizer_encode": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_int, _I64P, _INTP,
                                       _INTP, _U8P, C.c_int, _INTP]),
}

_lib = None


def lib():
    """The loaded library; raises if the in-tree build is missing (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C perceive_amd/csrc` "
                "(or __graft_entry__.build()); perceive_amd has no CPU fallback"
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the build is stale
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != PCV_OK:
        raise PcvError(status, lib().pcv_last_error().decode("utf-8", "replace"))


def f32p(a):
    return a.ctypes.data


def i64p(a):
    return a.ctypes.data


def u8p(a):
    return a.ctypes.data


def i32p(a):
    return a.ctypes.data

"""Model::new_pretrained for a local model directory (crates/perceive-core/model.rs:68-174,
model/configs.rs:97-119).  The reference reads, per model, the sentence-transformers layout

    model_data/<model>/modules.json, config.json, sentence_bert_config.json, tokenizer_config.json,
                       vocab.txt, 1_Pooling/config.json, [2_Dense/config.json], weights

with weights converted to rust-bert's `rust_model.ot`.  The directory is read by the library itself
(pcv_model_create_from_dir: JSON configs, tokenizer files, and the weights as `rust_model.ot`,
`model.safetensors` or `pytorch_model.bin`, all in C++); this file converts arguments.
"""
import ctypes as C
import os

from . import _ffi
from .model import Model, ModelError, SentenceEmbeddingsModelType

_POOL = {_ffi.POOL_MEAN: "mean", _ffi.POOL_CLS: "cls", _ffi.POOL_MAX: "max", _ffi.POOL_MEAN_SQRT_LEN: "mean_sqrt_len"}
_ACT = {_ffi.ACT_IDENTITY: "identity", _ffi.ACT_TANH: "tanh"}
_ARCH = {0: "bert", 1: "distilbert", 2: "roberta", 3: "albert"}


def model_dir_name(model_type):
    """configs.rs:42-69,121-141: the directory of an enum variant under model_data/ (pcv_model_type_dir_name)."""
    name = _ffi.lib().pcv_model_type_dir_name(int(model_type.value))
    return name.decode() if name else None


MODEL_DIRS = {mt: model_dir_name(mt) for mt in SentenceEmbeddingsModelType}


def parse_model_dir(directory):
    """Everything model.rs:84-151 reads from the directory's JSON files (pcv_model_dir_describe; no GPU needed):
    (description dict, tokenizer options dict, dense dict or None)."""
    d = _ffi.ModelDesc()
    arch, lower, strip = C.c_int(), C.c_int(), C.c_int()
    try:
        _ffi.check(_ffi.lib().pcv_model_dir_describe(str(directory).encode(), C.byref(d), C.byref(arch), C.byref(lower), C.byref(strip)))
    except _ffi.PcvError as e:
        raise ModelError(str(e)) from e
    desc = dict(vocab_size=d.vocab_size, hidden=d.hidden, layers=d.layers, heads=d.heads, intermediate=d.intermediate,
                max_positions=d.max_positions, type_vocab=d.type_vocab, layer_norm_eps=d.layer_norm_eps, pooling=_POOL[d.pooling],
                normalize=bool(d.normalize), dense_out=d.dense_out, dense_activation=_ACT[d.dense_activation],
                max_seq_length=d.max_seq_length, arch=_ARCH[arch.value], embedding_size=d.embedding_size,
                shared_layers=bool(d.shared_layers), hidden_act="gelu_new" if d.hidden_act else "gelu")
    tok = dict(lower_case=bool(lower.value), strip_accents=None if strip.value < 0 else bool(strip.value))
    dense = dict(out=d.dense_out, activation=_ACT[d.dense_activation]) if d.dense_out else None
    return desc, tok, dense


def checkpoint_tensors(path):
    """{name: (f32 array, or None for an integer tensor)} of a checkpoint file — model.safetensors, rust_model.ot or
    pytorch_model.bin — as the library's own readers see it (pcv_checkpoint_visit; nothing in the file is executed)."""
    import numpy as np

    out = {}

    @_ffi.TENSOR_VISITOR
    def visit(_user, name, shape, rank, dtype, values, numel):
        dims = tuple(shape[i] for i in range(rank))
        out[name.decode()] = None if not values else np.ctypeslib.as_array(values, shape=(numel,)).reshape(dims).copy()
        return 0

    try:
        _ffi.check(_ffi.lib().pcv_checkpoint_visit(str(path).encode(), visit, None))
    except _ffi.PcvError as e:
        raise ModelError(str(e)) from e
    return out


def new_pretrained(ctx, model, model_data_dir=None, compute="f32"):
    """Model::new_pretrained(model_type).  `model` is a SentenceEmbeddingsModelType (resolved under
    `model_data_dir`, the reference's `model_data/`, configs.rs:87-95) or a path to a model directory."""
    if isinstance(model, SentenceEmbeddingsModelType):
        name = model_dir_name(model)
        directory = os.path.join(model_data_dir or os.environ.get("PERCEIVE_MODEL_DATA", "model_data"), name or "?")
        model_type = model
    else:
        directory, model_type = str(model), SentenceEmbeddingsModelType.AllMiniLmL6V2
    return Model.from_dir(ctx, directory, compute=compute, model_type=model_type)

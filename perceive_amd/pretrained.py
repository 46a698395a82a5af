"""Model::new_pretrained for a local model directory (crates/perceive-core/model.rs:68-174,
model/configs.rs:97-119).  The reference reads, per model, the sentence-transformers layout

    model_data/<model>/modules.json, config.json, sentence_bert_config.json, tokenizer_config.json,
                       vocab.txt, 1_Pooling/config.json, [2_Dense/config.json], weights

with weights converted to rust-bert's `rust_model.ot`.  Here the weights are taken from the files the
HF checkpoint itself ships (`model.safetensors`, or `pytorch_model.bin` loaded with
`torch.load(weights_only=True)`), since `.ot` is a libtorch pickle.  Built: `ModelType::Bert`
(all-MiniLM-*, msmarco-bert-base-dot-v5) and `ModelType::DistilBert` (msmarco-distilbert-*,
distiluse-base-multilingual-cased) — DistilBERT is the same post-LayerNorm encoder without token-type
embeddings, so its tensors are renamed onto the BERT graph with a zero token-type row — and `ModelType::Roberta`
(all-distilroberta-v1): the BERT graph with a byte-level BPE tokenizer and the position table read from row
padding_idx + 1.  The ALBERT variant raises ModelError (SentencePiece tokenizer, factorised embeddings).
"""
import json
import os

import numpy as np

from . import _ffi
from .model import Model, ModelError, SentenceEmbeddingsModelType, make_desc
from .tokenizer import BertTokenizer, RobertaTokenizer

# configs.rs:42-69,121-141: directory names of the enum variants (sentence-transformers repo names)
MODEL_DIRS = {
    SentenceEmbeddingsModelType.AllMiniLmL6V2: "all-MiniLM-L6-v2",
    SentenceEmbeddingsModelType.AllMiniLmL12V2: "all-MiniLM-L12-v2",
    SentenceEmbeddingsModelType.MsMarcoBertBaseDotV5: "msmarco-bert-base-dot-v5",
    SentenceEmbeddingsModelType.MsMarcoDistilbertDotV5: "msmarco-distilbert-dot-v5",        # configs.rs:124-131
    SentenceEmbeddingsModelType.MsMarcoDistilbertBaseTasB: "msmarco-distilbert-base-tas-b",  # configs.rs:133-140
    SentenceEmbeddingsModelType.DistiluseBaseMultilingualCased: "distiluse-base-multilingual-cased",
    SentenceEmbeddingsModelType.AllDistilrobertaV1: "all-distilroberta-v1",
}

# HF DistilBertModel tensor names -> the BERT names the encoder graph uses
_DISTIL_RENAMES = (
    ("transformer.layer.", "encoder.layer."),
    (".attention.q_lin.", ".attention.self.query."),
    (".attention.k_lin.", ".attention.self.key."),
    (".attention.v_lin.", ".attention.self.value."),
    (".attention.out_lin.", ".attention.output.dense."),
    (".sa_layer_norm.", ".attention.output.LayerNorm."),
    (".ffn.lin1.", ".intermediate.dense."),
    (".ffn.lin2.", ".output.dense."),
    (".output_layer_norm.", ".output.LayerNorm."),
)


def _distilbert_to_bert(tensors, hidden):
    out = {}
    for k, v in tensors.items():
        k = k[11:] if k.startswith("distilbert.") else k
        for a, b in _DISTIL_RENAMES:
            k = k.replace(a, b)
        out[k] = v
    out["embeddings.token_type_embeddings.weight"] = np.zeros((1, hidden), np.float32)  # DistilBERT has none
    return out


def _read_json(path, default=None):
    if not os.path.exists(path):
        if default is not None:
            return default
        raise ModelError(f"missing model file {path}")
    with open(path, encoding="utf-8") as f:
        return json.load(f)


def _load_tensors(directory):
    st = os.path.join(directory, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file

        return load_file(st)
    pt = os.path.join(directory, "pytorch_model.bin")
    if os.path.exists(pt):
        import torch

        sd = torch.load(pt, map_location="cpu", weights_only=True)  # never unpickle arbitrary objects
        return {k: v.float().numpy() for k, v in sd.items()}
    raise ModelError(f"no model.safetensors / pytorch_model.bin under {directory}")


def parse_model_dir(directory):
    """Everything model.rs:84-151 reads, as plain dicts: (desc kwargs, tokenizer kwargs, module list)."""
    modules = _read_json(os.path.join(directory, "modules.json"))                       # model.rs:84-86
    kinds = [m["type"].split(".")[-1] for m in modules]
    if not kinds or kinds[0] != "Transformer":
        raise ModelError(f"{directory}: first module must be a Transformer, got {kinds}")
    cfg = _read_json(os.path.join(directory, "config.json"))                            # model.rs:118-121
    arch = cfg.get("model_type", "bert")
    if arch not in ("bert", "distilbert", "roberta"):
        raise ModelError(f"transformer type '{arch}' is not supported (BERT, DistilBERT and RoBERTa only)")
    if cfg.get("hidden_act", cfg.get("activation", "gelu")) != "gelu":
        raise ModelError(f"activation '{cfg.get('hidden_act', cfg.get('activation'))}' is not supported (erf GELU only)")
    if arch == "distilbert":  # same quantities under DistilBertConfig's names
        if cfg.get("sinusoidal_pos_embds"):
            raise ModelError("sinusoidal position embeddings are not supported")
        cfg = dict(cfg, hidden_size=cfg["dim"], num_hidden_layers=cfg["n_layers"], num_attention_heads=cfg["n_heads"],
                   intermediate_size=cfg["hidden_dim"], type_vocab_size=1, layer_norm_eps=1e-12)
    sbert = _read_json(os.path.join(directory, "sentence_bert_config.json"), {})        # model.rs:93-95
    tok_cfg = _read_json(os.path.join(directory, "tokenizer_config.json"), {})          # model.rs:90-92
    pooling_dir = next((m["path"] for m in modules if m["type"].endswith("Pooling")), "1_Pooling")
    pool = _read_json(os.path.join(directory, pooling_dir, "config.json"))              # model.rs:134-135
    if pool.get("pooling_mode_cls_token"):
        pooling = "cls"
    elif pool.get("pooling_mode_max_tokens"):
        pooling = "max"
    elif pool.get("pooling_mode_mean_sqrt_len_tokens"):
        pooling = "mean_sqrt_len"
    else:
        pooling = "mean"
    dense = None
    for m in modules:                                                                    # model.rs:139-149
        if m["type"].endswith("Dense"):
            dc = _read_json(os.path.join(directory, m["path"], "config.json"))
            act = dc.get("activation_function", "torch.nn.modules.linear.Identity").split(".")[-1].lower()
            if act not in ("tanh", "identity"):
                raise ModelError(f"Dense activation '{act}' is not supported")
            dense = dict(path=m["path"], out=dc["out_features"], activation=act, bias=dc.get("bias", True))
    desc = dict(
        vocab_size=cfg["vocab_size"], hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"],
        heads=cfg["num_attention_heads"], intermediate=cfg["intermediate_size"],
        max_positions=cfg["max_position_embeddings"], type_vocab=cfg.get("type_vocab_size", 2),
        layer_norm_eps=cfg.get("layer_norm_eps", 1e-12), pooling=pooling,
        normalize=any(k == "Normalize" for k in kinds),                                  # has_normalization(), model.rs:151
        dense_out=dense["out"] if dense else 0, dense_activation=dense["activation"] if dense else "identity",
        max_seq_length=sbert.get("max_seq_length", 128),
    )
    lower = tok_cfg.get("do_lower_case", sbert.get("do_lower_case", True))               # model.rs:108-110
    tok = dict(lower_case=bool(lower), strip_accents=tok_cfg.get("strip_accents"))
    if arch == "roberta":
        # RoBERTa numbers positions from padding_idx + 1 (pad tokens sit at padding_idx): for right-padded batches
        # token l has position l + pad + 1, so the table is used from that row on (see new_pretrained)
        desc["_pos_shift"] = int(cfg.get("pad_token_id", 1)) + 1
        desc["max_positions"] = cfg["max_position_embeddings"] - desc["_pos_shift"]
        tok = dict(add_prefix_space=bool(tok_cfg.get("add_prefix_space", False)))
    desc["_arch"] = arch
    return desc, tok, dense


def new_pretrained(ctx, model, model_data_dir=None, compute="f32"):
    """Model::new_pretrained(model_type).  `model` is a SentenceEmbeddingsModelType (resolved under
    `model_data_dir`, the reference's `model_data/`, configs.rs:87-95) or a path to a model directory."""
    if isinstance(model, SentenceEmbeddingsModelType):
        if model not in MODEL_DIRS:
            raise ModelError(f"{model.name} is not a BERT / DistilBERT / RoBERTa model; only {[m.name for m in MODEL_DIRS]} are built")
        directory = os.path.join(model_data_dir or os.environ.get("PERCEIVE_MODEL_DATA", "model_data"), MODEL_DIRS[model])
        model_type = model
    else:
        directory, model_type = str(model), SentenceEmbeddingsModelType.AllMiniLmL6V2
    desc_kw, tok_kw, dense = parse_model_dir(directory)
    arch = desc_kw.pop("_arch")
    pos_shift = desc_kw.pop("_pos_shift", 0)
    if arch == "roberta":
        tokenizer = RobertaTokenizer(os.path.join(directory, "vocab.json"), os.path.join(directory, "merges.txt"), **tok_kw)
    else:
        tokenizer = BertTokenizer(os.path.join(directory, "vocab.txt"), **tok_kw)        # model.rs:96-113
    d = make_desc(compute=compute, **desc_kw)
    m = Model(ctx, d, synthetic_seed=0, model_type=model_type, tokenizer=tokenizer)
    tensors = _load_tensors(directory)                                                   # var_store.load, model.rs:124
    tensors = {(k[5:] if k.startswith("bert.") else k[8:] if k.startswith("roberta.") else k): v for k, v in tensors.items()}
    if pos_shift:
        tensors["embeddings.position_embeddings.weight"] = np.ascontiguousarray(
            tensors["embeddings.position_embeddings.weight"][pos_shift:])
    if arch == "distilbert":
        tensors = _distilbert_to_bert(tensors, desc_kw["hidden"])
    if dense:
        dt = _load_tensors(os.path.join(directory, dense["path"]))
        tensors["dense.linear.weight"] = dt["linear.weight"]
        tensors["dense.linear.bias"] = dt.get("linear.bias", np.zeros(dense["out"], np.float32))
    missing = [n for n in m.tensor_names() if n not in tensors]
    if missing:
        m.close()
        raise ModelError(f"{directory}: checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    try:
        m.load_state_dict({n: np.asarray(tensors[n], dtype=np.float32) for n in m.tensor_names()})
    except _ffi.PcvError as e:
        m.close()
        raise ModelError(str(e)) from e
    return m

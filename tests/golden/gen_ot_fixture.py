"""Generates tests/golden/ot/: a tiny sentence-transformers model directory whose weights are `rust_model.ot`
files written by libtorch's own OutputArchive (ot_writer.cpp = what tch's Tensor::save_multi calls), plus the
answers the tests check:

  ot/rust_model.ot, ot/2_Dense/rust_model.ot   the archives (BERT names with the "bert." prefix rust-bert's VarStore uses)
  ot/*.json, ot/1_Pooling, ot/2_Dense/config.json, ot/vocab.txt
  ot/expected.json       per tensor: shape, f64 sum, first and last values (CPU reader test);
                         texts + the embeddings Hugging Face BertModel + mean pooling + Dense(tanh) + L2 gives (GPU test)

No `.ot` file from the reference's pipeline exists offline (scripts/install_models.sh needs the network), so the
format is pinned by the writer the pipeline uses, from the libtorch of the installed PyTorch wheel (the reference
pins tch 0.10.1 = libtorch 1.13; the zip + data.pkl layout has been the same since 1.6).
Run once:   python tests/golden/gen_ot_fixture.py
"""
import json
import os
import shutil
import struct
import subprocess
import tempfile

os.environ["HF_HUB_OFFLINE"] = "1"
import numpy as np
import torch
from transformers import BertConfig, BertModel, BertTokenizerFast

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ot")


def write_raw(path, tensors):
    with open(path, "wb") as f:
        for name, arr in tensors.items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<I", arr.ndim) + struct.pack(f"<{arr.ndim}q", *arr.shape))
            f.write(arr.tobytes())


def build_writer(tmp):
    tdir = os.path.dirname(torch.__file__)
    exe = os.path.join(tmp, "ot_writer")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(HERE, "ot_writer.cpp"), f"-I{tdir}/include",
                           f"-I{tdir}/include/torch/csrc/api/include", f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10",
                           f"-Wl,-rpath,{tdir}/lib", "-o", exe])
    return exe


def summary(arr):
    flat = np.asarray(arr, dtype=np.float32).ravel()
    return {"shape": list(arr.shape), "sum": float(flat.astype(np.float64).sum()), "head": [float(x) for x in flat[:3]],
            "tail": [float(x) for x in flat[-3:]]}


def main():
    torch.manual_seed(11)
    torch.set_num_threads(1)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, "1_Pooling"))
    os.makedirs(os.path.join(OUT, "2_Dense"))
    vocab = os.path.join(HERE, "tokenizer_vocab.txt")
    nvocab = sum(1 for _ in open(vocab, encoding="utf-8"))
    cfg = BertConfig(vocab_size=nvocab, hidden_size=128, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=32, layer_norm_eps=1e-12, hidden_act="gelu")
    model = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if "LayerNorm.weight" in k:
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn_like(v))
            elif v.dim() == 2:
                v.mul_(4.0)
    rng = np.random.default_rng(5)
    dense_w = (rng.standard_normal((64, 128)) * 0.2).astype(np.float32)
    dense_b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    # rust-bert's VarStore paths: BertModel lives under "bert" (convert_model.py keeps the checkpoint's names)
    main_tensors = {"bert." + k: v.numpy() for k, v in model.state_dict().items() if v.dtype == torch.float32}
    dense_tensors = {"linear.weight": dense_w, "linear.bias": dense_b}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_writer(tmp)
        for tensors, dst in ((main_tensors, os.path.join(OUT, "rust_model.ot")), (dense_tensors, os.path.join(OUT, "2_Dense", "rust_model.ot"))):
            raw = os.path.join(tmp, "t.raw")
            write_raw(raw, tensors)
            subprocess.check_call([exe, raw, dst])

    shutil.copy(vocab, os.path.join(OUT, "vocab.txt"))
    cfg_json = json.loads(cfg.to_json_string())
    json.dump({k: cfg_json[k] for k in ("model_type", "vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                        "intermediate_size", "max_position_embeddings", "layer_norm_eps", "hidden_act", "type_vocab_size")},
              open(os.path.join(OUT, "config.json"), "w"), indent=1)
    json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
               {"idx": 2, "name": "2", "path": "2_Dense", "type": "sentence_transformers.models.Dense"},
               {"idx": 3, "name": "3", "path": "3_Normalize", "type": "sentence_transformers.models.Normalize"}],
              open(os.path.join(OUT, "modules.json"), "w"), indent=1)
    json.dump({"word_embedding_dimension": 128, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
               "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False},
              open(os.path.join(OUT, "1_Pooling", "config.json"), "w"), indent=1)
    json.dump({"in_features": 128, "out_features": 64, "bias": True, "activation_function": "torch.nn.modules.activation.Tanh"},
              open(os.path.join(OUT, "2_Dense", "config.json"), "w"), indent=1)
    json.dump({"max_seq_length": 24, "do_lower_case": False}, open(os.path.join(OUT, "sentence_bert_config.json"), "w"))
    json.dump({"do_lower_case": True}, open(os.path.join(OUT, "tokenizer_config.json"), "w"))

    texts = ["Hello world", "The search of embeddings, really?", "document " * 30, "Café naïve"]
    tok = BertTokenizerFast(os.path.join(OUT, "vocab.txt"), do_lower_case=True)
    enc = tok(texts, padding=True, truncation=True, max_length=24, return_tensors="pt")
    with torch.no_grad():
        h = model(**enc).last_hidden_state
        msk = enc["attention_mask"].unsqueeze(-1).float()
        pooled = (h * msk).sum(1) / msk.sum(1).clamp_min(1e-9)
        pooled = torch.tanh(pooled @ torch.from_numpy(dense_w).T + torch.from_numpy(dense_b))
        emb = (pooled / pooled.norm(dim=1, keepdim=True).clamp_min(1e-12)).numpy()
    expected = {"tensors": {k: summary(v) for k, v in main_tensors.items()},
                "dense": {k: summary(v) for k, v in dense_tensors.items()},
                "texts": texts, "embeddings": [[float(x) for x in row] for row in emb],
                "libtorch": torch.__version__}
    json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"))
    print("wrote", OUT, {f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if os.path.isfile(os.path.join(OUT, f))})


if __name__ == "__main__":
    main()

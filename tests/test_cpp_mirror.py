"""The C++ host mirror (include/perceive.hpp): compiles against the C ABI with plain g++ (CPU), and
its end-to-end program passes on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.bin")


def compile_mirror():
    subprocess.run(
        ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), SRC, "-o", BIN,
         "-L" + os.path.join(ROOT, "perceive_amd"), "-lperceive_hip", "-Wl,-rpath," + os.path.join(ROOT, "perceive_amd")],
        check=True, capture_output=True, text=True)


def test_cpp_mirror_compiles_and_links():
    compile_mirror()
    assert os.path.exists(BIN)


def write_model_dir(d, golden_dir):
    """A sentence-transformers directory without weights (JSON configs + vocab.txt)."""
    import json
    import shutil

    os.makedirs(os.path.join(d, "1_Pooling"), exist_ok=True)
    vocab = os.path.join(golden_dir, "tokenizer_vocab.txt")
    shutil.copy(vocab, os.path.join(d, "vocab.txt"))
    nvocab = sum(1 for _ in open(vocab, encoding="utf-8"))
    json.dump({"model_type": "bert", "vocab_size": nvocab, "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 4,
               "intermediate_size": 256, "max_position_embeddings": 128, "layer_norm_eps": 1e-12, "hidden_act": "gelu"},
              open(os.path.join(d, "config.json"), "w"))
    json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
               {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}],
              open(os.path.join(d, "modules.json"), "w"))
    json.dump({"pooling_mode_mean_tokens": True}, open(os.path.join(d, "1_Pooling", "config.json"), "w"))
    json.dump({"max_seq_length": 64, "do_lower_case": True}, open(os.path.join(d, "sentence_bert_config.json"), "w"))


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu(tmp_path, golden_dir):
    if not os.path.exists(BIN):
        compile_mirror()
    write_model_dir(str(tmp_path / "model"), golden_dir)
    ot = os.path.join(golden_dir, "ot", "2_Dense", "rust_model.ot")
    r = subprocess.run([BIN, str(tmp_path / "model"), ot], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_mirror_test: ok" in r.stdout

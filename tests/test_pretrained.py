"""Model::new_pretrained over a local sentence-transformers model directory (model.rs:68-174): config
parsing on the CPU, and — on the GPU — text in / embedding out against Hugging Face running the same
checkpoint (BertModel + mean pooling + normalisation) on the CPU.  The directory is synthetic (random
weights written by transformers' save_pretrained): no real checkpoint exists offline."""
import json
import os
import shutil

import numpy as np
import pytest

import perceive_amd as pa

os.environ.setdefault("HF_HUB_OFFLINE", "1")


def make_model_dir(tmp_path, golden_dir, with_dense=False, fmt="safetensors"):
    import torch
    from transformers import BertConfig, BertModel

    torch.manual_seed(3)
    d = tmp_path / "all-MiniLM-L6-v2"
    d.mkdir(parents=True)
    vocab = os.path.join(golden_dir, "tokenizer_vocab.txt")
    nvocab = sum(1 for _ in open(vocab, encoding="utf-8"))
    cfg = BertConfig(vocab_size=nvocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=256, max_position_embeddings=64, layer_norm_eps=1e-12, hidden_act="gelu")
    model = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dim() == 2:
                v.mul_(4.0)
    model.save_pretrained(d, safe_serialization=(fmt == "safetensors"))
    shutil.copy(vocab, d / "vocab.txt")
    mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
            {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": 128, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
        "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}))
    dense_w = None
    if with_dense:
        from safetensors.numpy import save_file

        (d / "2_Dense").mkdir()
        rng = np.random.default_rng(0)
        dense_w = ((rng.standard_normal((64, 128)) * 0.2).astype(np.float32), (rng.standard_normal(64) * 0.1).astype(np.float32))
        save_file({"linear.weight": dense_w[0], "linear.bias": dense_w[1]}, str(d / "2_Dense" / "model.safetensors"))
        (d / "2_Dense" / "config.json").write_text(json.dumps({
            "in_features": 128, "out_features": 64, "bias": True, "activation_function": "torch.nn.modules.activation.Tanh"}))
        mods.append({"idx": 2, "name": "2", "path": "2_Dense", "type": "sentence_transformers.models.Dense"})
    mods.append({"idx": len(mods), "name": str(len(mods)), "path": f"{len(mods)}_Normalize", "type": "sentence_transformers.models.Normalize"})
    (d / "modules.json").write_text(json.dumps(mods))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 32, "do_lower_case": False}))
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True}))
    return d, model, dense_w


def test_parse_model_dir(tmp_path, golden_dir):
    d, _, _ = make_model_dir(tmp_path, golden_dir, with_dense=True)
    desc, tok, dense = pa.parse_model_dir(str(d))
    assert desc["hidden"] == 128 and desc["layers"] == 2 and desc["heads"] == 4 and desc["intermediate"] == 256
    assert desc["pooling"] == "mean" and desc["normalize"] is True and desc["max_seq_length"] == 32
    assert desc["dense_out"] == 64 and desc["dense_activation"] == "tanh"
    assert tok == {"lower_case": True, "strip_accents": None}  # tokenizer_config wins over sentence_bert_config
    assert desc["arch"] == "bert" and desc["embedding_size"] == 0 and not desc["shared_layers"] and desc["hidden_act"] == "gelu"
    # an ALBERT config: factorised embeddings, shared layer weights, gelu_new
    cfg = json.loads((d / "config.json").read_text())
    cfg.update(model_type="albert", embedding_size=128, num_hidden_groups=1, inner_group_num=1, hidden_act="gelu_new")
    (d / "config.json").write_text(json.dumps(cfg))
    desc, _, _ = pa.parse_model_dir(str(d))
    assert desc["arch"] == "albert" and desc["embedding_size"] == 128 and desc["shared_layers"] and desc["hidden_act"] == "gelu_new"
    # unsupported architectures / shapes fail loudly
    for bad in (dict(model_type="t5"), dict(num_hidden_groups=2), dict(hidden_act="relu")):
        (d / "config.json").write_text(json.dumps({**cfg, **bad}))
        with pytest.raises(pa.ModelError):
            pa.parse_model_dir(str(d))
    with pytest.raises(pa.ModelError):
        pa.parse_model_dir(str(tmp_path / "nope"))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,with_dense", [("safetensors", False), ("bin", False), ("safetensors", True)])
def test_new_pretrained_matches_hf(ctx, tmp_path, golden_dir, fmt, with_dense):
    import torch
    from transformers import BertTokenizerFast

    d, hf, dense_w = make_model_dir(tmp_path, golden_dir, with_dense=with_dense, fmt=fmt)
    m = pa.new_pretrained(ctx, pa.SentenceEmbeddingsModelType.AllMiniLmL6V2, model_data_dir=str(tmp_path))
    assert m.model_type.model_id == 0 and m.desc.max_seq_length == 32
    texts = ["Hello world", "The search of embeddings, really?", "document " * 40, "Café naïve"]
    out = m.encode(texts)
    tok = BertTokenizerFast(str(d / "vocab.txt"), do_lower_case=True)
    enc = tok(texts, padding=True, truncation=True, max_length=32, return_tensors="pt")
    with torch.no_grad():
        h = hf(**enc).last_hidden_state
        msk = enc["attention_mask"].unsqueeze(-1).float()
        pooled = (h * msk).sum(1) / msk.sum(1).clamp_min(1e-9)
        if with_dense:
            pooled = torch.tanh(pooled @ torch.from_numpy(dense_w[0]).T + torch.from_numpy(dense_w[1]))
        ref = (pooled / pooled.norm(dim=1, keepdim=True).clamp_min(1e-12)).numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-4
    q = pa.encode_query(m, "hello world")
    np.testing.assert_allclose(q, out[0], atol=1e-6)
    m.close()
    with pytest.raises(pa.ModelError):  # no such directory
        pa.new_pretrained(ctx, pa.SentenceEmbeddingsModelType.MsMarcoDistilbertDotV5, model_data_dir=str(tmp_path))


def make_distilbert_dir(tmp_path, golden_dir, name, pooling_cls, normalize):
    import torch
    from transformers import DistilBertConfig, DistilBertModel

    torch.manual_seed(5)
    d = tmp_path / name
    d.mkdir(parents=True)
    vocab = os.path.join(golden_dir, "tokenizer_vocab.txt")
    nvocab = sum(1 for _ in open(vocab, encoding="utf-8"))
    cfg = DistilBertConfig(vocab_size=nvocab, dim=128, n_layers=2, n_heads=4, hidden_dim=256, max_position_embeddings=64,
                           activation="gelu", sinusoidal_pos_embds=False, dropout=0.0, attention_dropout=0.0)
    model = DistilBertModel(cfg).eval()
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dim() == 2:
                v.mul_(4.0)
    model.save_pretrained(d, safe_serialization=True)
    shutil.copy(vocab, d / "vocab.txt")
    mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
            {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    if normalize:
        mods.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": 128, "pooling_mode_cls_token": pooling_cls, "pooling_mode_mean_tokens": not pooling_cls,
        "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}))
    (d / "modules.json").write_text(json.dumps(mods))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 48, "do_lower_case": False}))
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True}))
    return d, model


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["MsMarcoDistilbertBaseTasB", "MsMarcoDistilbertDotV5"])
def test_distilbert_checkpoints_match_hf(ctx, tmp_path, golden_dir, variant):
    # ModelType::DistilBert (configs.rs:124-140): the DistilBERT tensors renamed onto the BERT graph with a zero
    # token-type row; tas-b pools the CLS token, dot-v5 the mean; neither normalises (dot-product models)
    import torch
    from transformers import BertTokenizerFast

    mt = getattr(pa.SentenceEmbeddingsModelType, variant)
    cls_pool = variant == "MsMarcoDistilbertBaseTasB"
    d, hf = make_distilbert_dir(tmp_path, golden_dir, pa.pretrained.MODEL_DIRS[mt], cls_pool, normalize=False)
    m = pa.new_pretrained(ctx, mt, model_data_dir=str(tmp_path))
    assert m.model_type is mt and m.desc.normalize == 0 and m.desc.type_vocab == 1
    texts = ["Hello world", "how do transformers without token types work?", "document " * 60, "x"]
    out = m.encode(texts)
    tok = BertTokenizerFast(str(d / "vocab.txt"), do_lower_case=True)
    enc = tok(texts, padding=True, truncation=True, max_length=48, return_tensors="pt")
    with torch.no_grad():
        h = hf(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]).last_hidden_state
        if cls_pool:
            ref = h[:, 0].numpy()
        else:
            msk = enc["attention_mask"].unsqueeze(-1).float()
            ref = ((h * msk).sum(1) / msk.sum(1).clamp_min(1e-9)).numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    m.close()


@pytest.mark.gpu
def test_roberta_checkpoint_matches_hf(ctx, tmp_path, golden_dir):
    # ModelType::Roberta (all-distilroberta-v1): byte-level BPE tokenizer + position ids from padding_idx + 1.
    # Expected values: HF RobertaModel on ids made by the `tokenizers` library from the same vocab / merges.
    import torch
    from tokenizers import ByteLevelBPETokenizer
    from transformers import RobertaConfig, RobertaModel

    torch.manual_seed(9)
    mt = pa.SentenceEmbeddingsModelType.AllDistilrobertaV1
    d = tmp_path / pa.pretrained.MODEL_DIRS[mt]
    d.mkdir(parents=True)
    shutil.copy(os.path.join(golden_dir, "bpe_vocab.json"), d / "vocab.json")
    shutil.copy(os.path.join(golden_dir, "bpe_merges.txt"), d / "merges.txt")
    nvocab = len(json.load(open(d / "vocab.json", encoding="utf-8")))
    cfg = RobertaConfig(vocab_size=nvocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                        max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0,
                        eos_token_id=2, hidden_act="gelu")
    hf = RobertaModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for k, v in hf.state_dict().items():
            if v.dim() == 2:
                v.mul_(4.0)
    hf.save_pretrained(d, safe_serialization=True)
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": 128, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
        "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}))
    (d / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
        {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}]))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 40, "do_lower_case": False}))
    (d / "tokenizer_config.json").write_text(json.dumps({"add_prefix_space": False}))
    m = pa.new_pretrained(ctx, mt, model_data_dir=str(tmp_path))
    assert m.pad_token_id == 1 and m.desc.max_positions == 64 and m.desc.type_vocab == 1
    texts = ["Hello world", "don't stop searching, it's 1234 times faster!", "word " * 60, "Ünïcödé 中文 🙂"]
    out = m.encode(texts)
    tok = ByteLevelBPETokenizer(str(d / "vocab.json"), str(d / "merges.txt"))
    rows = [[0] + tok.encode(t).ids[:38] + [2] for t in texts]
    L = max(len(r) for r in rows)
    ids = torch.tensor([r + [1] * (L - len(r)) for r in rows])
    am = (ids != 1).long()
    with torch.no_grad():
        h = hf(input_ids=ids, attention_mask=am).last_hidden_state
        msk = am.unsqueeze(-1).float()
        pooled = (h * msk).sum(1) / msk.sum(1).clamp_min(1e-9)
        ref = (pooled / pooled.norm(dim=1, keepdim=True).clamp_min(1e-12)).numpy()
    assert np.abs(out - ref).max() < 1e-4
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["f32", "bf16x3", "f16x2"])
def test_albert_checkpoint_matches_hf(ctx, tmp_path, golden_dir, compute):
    # ModelType::Albert (paraphrase-albert-small-v2, configs.rs:35): SentencePiece tokenizer, embeddings at width 128
    # mapped up to hidden, ONE set of layer weights run num_hidden_layers times, gelu_new.  Expected values: HF
    # AlbertModel on ids the sentencepiece library makes from the prepared text (tests/golden/gen_spm_golden.py).
    import sys

    import sentencepiece as spm
    import torch
    from transformers import AlbertConfig, AlbertModel

    sys.path.insert(0, golden_dir)
    try:
        import gen_spm_golden as gen
    finally:
        sys.path.remove(golden_dir)
    torch.manual_seed(13)
    mt = pa.SentenceEmbeddingsModelType.ParaphraseAlbertSmallV2
    d = tmp_path / pa.pretrained.MODEL_DIRS[mt]
    assert d.name == "paraphrase-albert-small-v2"
    d.mkdir(parents=True)
    shutil.copy(os.path.join(golden_dir, "spiece.model"), d / "spiece.model")
    sp = spm.SentencePieceProcessor(model_file=str(d / "spiece.model"))
    cfg = AlbertConfig(vocab_size=sp.get_piece_size(), embedding_size=128, hidden_size=256, num_hidden_layers=3, num_hidden_groups=1,
                       inner_group_num=1, num_attention_heads=4, intermediate_size=512, max_position_embeddings=64, type_vocab_size=2,
                       layer_norm_eps=1e-12, hidden_act="gelu_new", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       classifier_dropout_prob=0.0, pad_token_id=0, bos_token_id=2, eos_token_id=3)
    hf = AlbertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for k, v in hf.state_dict().items():
            if "LayerNorm.weight" in k or "layer_norm.weight" in k:
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn_like(v))
            elif v.dim() == 2:
                v.mul_(3.0)
    hf.save_pretrained(d, safe_serialization=True)
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": 256, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
        "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}))
    (d / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 48, "do_lower_case": False}))
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True, "keep_accents": False, "remove_space": True}))
    m = pa.new_pretrained(ctx, mt, model_data_dir=str(tmp_path), compute=compute)
    assert m.model_type.model_id == 4 and m.pad_token_id == 0
    assert m.desc.embedding_size == 128 and m.desc.shared_layers == 1 and m.desc.hidden_act == 1 and m.desc.layers == 3
    texts = ["Hello world", "The search of embeddings, really? In 1999, about 1,000 items", "returns a new string " * 20, "Café naïve"]
    out = m.encode(texts)
    rows = []
    for t in texts:
        pieces = gen.albert_pieces(sp, gen.prepare(t, True, True))
        rows.append([2] + [sp.piece_to_id(p) for p in pieces][:46] + [3])
    L = max(len(r) for r in rows)
    ids = torch.tensor([r + [0] * (L - len(r)) for r in rows])
    am = (ids != 0).long()
    with torch.no_grad():
        h = hf(input_ids=ids, attention_mask=am).last_hidden_state
        msk = am.unsqueeze(-1).float()
        ref = ((h * msk).sum(1) / msk.sum(1).clamp_min(1e-9)).numpy()
    assert out.shape == ref.shape == (4, 256)
    assert np.abs(out - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # highlight runs on the SentencePiece offsets as on any other tokenizer
    hl = m.highlight("string", ["nothing here", "it returns a new string object with the characters reversed and joined " * 3])
    assert len(hl) == 2
    m.close()

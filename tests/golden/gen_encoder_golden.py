"""Generates tests/golden/encoder_tiny.npz with Hugging Face `BertModel` on CPU (offline, built from a
local BertConfig with seeded random weights — no checkpoint exists in this environment).

rust-bert's BertModel (what `SentenceEmbeddingsOption::forward` runs at worker.rs:85-86) and HF's are
the same architecture: embeddings + LayerNorm, per layer self-attention / add&norm / erf-GELU FFN /
add&norm.  rust-bert adds (1-mask)*-10000 to the scores where HF adds dtype-min; both underflow to an
exact 0 after the f32 softmax for any realistic logits.  The pooling / normalisation tail restates
worker.rs:88-103.  Run once, offline:   python tests/golden/gen_encoder_golden.py
"""
import os

os.environ["HF_HUB_OFFLINE"] = "1"
import numpy as np
import torch
from transformers import BertConfig, BertModel

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(1234)
    torch.set_num_threads(1)
    desc = dict(vocab=300, hidden=128, layers=2, heads=4, inter=256, max_pos=64, eps=1e-12)
    cfg = BertConfig(vocab_size=desc["vocab"], hidden_size=desc["hidden"], num_hidden_layers=desc["layers"],
                     num_attention_heads=desc["heads"], intermediate_size=desc["inter"],
                     max_position_embeddings=desc["max_pos"], layer_norm_eps=desc["eps"], hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    sd = model.state_dict()
    with torch.no_grad():  # make weights less trivial than the 0.02-std init / unit LayerNorm
        for k, v in sd.items():
            if "LayerNorm.weight" in k:
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn_like(v))
            elif v.dim() == 2:
                v.copy_(v * 3.0)
    B, L = 5, 24
    lens = [24, 17, 9, 1, 24]
    ids = torch.randint(1, desc["vocab"], (B, L))
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    ids = ids * mask  # pad id 0, mask = id != pad (tokenize.rs:36-46)
    with torch.no_grad():
        o = model(input_ids=ids, attention_mask=mask, output_hidden_states=True)
        hs = torch.stack(o.hidden_states)  # [layers+1, B, L, H]
        tok = o.last_hidden_state
        m = mask.unsqueeze(-1).float()
        mean = (tok * m).sum(1) / m.sum(1).clamp_min(1e-9)                      # worker.rs:88-89
        normed = mean / mean.norm(2, dim=1, keepdim=True).clamp_min(1e-12)      # worker.rs:95-103
    out = {f"w.{k}": v.numpy() for k, v in sd.items() if "position_ids" not in k}
    out.update(ids=ids.numpy(), mask=mask.numpy(), hidden=hs.numpy(), mean=mean.numpy(), normed=normed.numpy(),
               desc=np.array([desc["vocab"], desc["hidden"], desc["layers"], desc["heads"], desc["inter"], desc["max_pos"]]))
    np.savez_compressed(os.path.join(HERE, "encoder_tiny.npz"), **out)
    print("wrote encoder_tiny.npz", os.path.getsize(os.path.join(HERE, "encoder_tiny.npz")))


if __name__ == "__main__":
    main()

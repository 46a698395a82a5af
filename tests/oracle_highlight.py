"""Independent CPU restatement of Model::highlight (crates/perceive-core/model/highlight.rs:23-165).
TEST INFRASTRUCTURE ONLY.  Tokenization by Hugging Face BertTokenizerFast, embeddings by the C
oracle encoder, scores by numpy — nothing from perceive_amd."""
import os

import numpy as np

os.environ.setdefault("HF_HUB_OFFLINE", "1")


def highlight(orc, desc, weights, vocab_path, max_seq_length, query, documents, chunk_size=20, chunk_overlap=4,
              pad_id=0):
    from transformers import BertTokenizerFast

    tok = BertTokenizerFast(vocab_path, do_lower_case=True)

    def enc_tokens(token_lists):
        L = max(len(t) for t in token_lists)
        ids = np.full((len(token_lists), L), pad_id, np.int64)
        for i, t in enumerate(token_lists):
            ids[i, : len(t)] = t
        mask = (ids != pad_id).astype(np.int64)  # tokenize.rs:36-46
        out, _ = orc.encode_tokens(desc, weights, ids, mask)
        return out

    q_ids = tok(query, add_special_tokens=True, max_length=max_seq_length, truncation=True)["input_ids"]
    query_encoding = enc_tokens([q_ids])  # highlight.rs:29
    docs = [tok(d, add_special_tokens=True, return_offsets_mapping=True, return_special_tokens_mask=True) for d in documents]
    inc = chunk_size - chunk_overlap
    chunks, bounds, doc_bounds = [], [], []
    for d in docs:  # highlight.rs:53-100
        ids, special = d["input_ids"], d["special_tokens_mask"]
        i = 0
        while i + chunk_overlap < len(ids):
            s, e = i, min(i + chunk_size, len(ids))
            ls = ll = cs = cl = 0
            for index, sp in enumerate(special[s:e]):
                if sp == 0:
                    cl += 1
                else:
                    if cl > ll:
                        ls, ll = cs, cl
                    cs, cl = index, 0
            if cl > ll:
                ls, ll = cs, cl
            s = s + ls
            e = min(s + ll, e)
            if e - s >= chunk_size // 2:
                bounds.append((s, e))
                chunks.append(ids[s:e])
            i += inc
        doc_bounds.append(len(chunks))
    scores = (query_encoding @ enc_tokens(chunks).T)[0] if chunks else np.zeros(0, np.float32)  # lib.rs:63-65
    out = []
    for index, end in enumerate(doc_bounds):  # highlight.rs:114-162
        start = 0 if index == 0 else doc_bounds[index - 1]
        sc = scores[start:end]
        if sc.size == 0:
            out.append(None)
            continue
        best = max(range(sc.size), key=lambda j: (sc[j], j))  # last of equal maxima
        lo, hi = bounds[start + best]
        ts = te = 0
        for o, sp in list(zip(docs[index]["offset_mapping"], docs[index]["special_tokens_mask"]))[lo:hi]:
            if sp:
                continue
            if ts == 0 and te == 0:
                ts, te = o
            else:
                ts, te = min(ts, o[0]), max(te, o[1])
        doc = documents[index]
        out.append(doc[ts:te + 1] if (ts < len(doc) and te + 1 < len(doc)) else "")
    return out, scores

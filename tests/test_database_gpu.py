"""GPU: Searcher::build / rebuild_source / search_vector_and_retrieve over a SQLite file shaped like
the one the reference's pipeline writes (tables and columns named as the queries at
search.rs:87-93,209-211 expect; the schema below is a minimal restatement for the test)."""
import os
import sqlite3

import numpy as np
import pytest

import perceive_amd as pa

pytestmark = pytest.mark.gpu

SCHEMA = """
CREATE TABLE sources (id INTEGER PRIMARY KEY, name TEXT NOT NULL);
CREATE TABLE items (id INTEGER PRIMARY KEY, source_id INTEGER NOT NULL, external_id TEXT NOT NULL,
  content TEXT NOT NULL, name TEXT, author TEXT, description TEXT, modified BIGINT, last_accessed BIGINT,
  skipped TEXT, hidden_at BIGINT);
CREATE TABLE item_embeddings (model_id INT NOT NULL, model_version INT NOT NULL, item_id BIGINT NOT NULL,
  item_index_version BIGINT NOT NULL, embedding BLOB NOT NULL, PRIMARY KEY(model_id, model_version, item_id));
"""


def test_build_search_retrieve_rebuild(ctx, oracle, tmp_path):
    rng = np.random.default_rng(8)
    D, N = 384, 400
    emb = rng.standard_normal((N, D)).astype(np.float32)
    path = str(tmp_path / "perceive.sqlite3")
    conn = sqlite3.connect(path)
    conn.executescript(SCHEMA)
    conn.executemany("INSERT INTO sources (id, name) VALUES (?, ?)", [(1, "notes"), (2, "history"), (3, "empty")])
    src = np.where(np.arange(N) % 4 == 0, 2, 1)
    for i in range(N):
        iid = 10 + i
        conn.execute("INSERT INTO items (id, source_id, external_id, content, name, skipped, hidden_at) VALUES (?,?,?,?,?,?,?)",
                     (iid, int(src[i]), f"doc{i}.md", f"content {i}", f"name {i}",
                      "not_found" if i == 5 else None, 123 if i == 6 else None))
        conn.execute("INSERT INTO item_embeddings VALUES (0, 0, ?, 1, ?)", (iid, pa.serialize_embedding(emb[i])))
        conn.execute("INSERT INTO item_embeddings VALUES (7, 0, ?, 1, ?)", (iid, pa.serialize_embedding(-emb[i])))  # other model
    conn.commit()
    conn.close()

    db = pa.Database(path)
    s = pa.build_searcher(ctx, db, 0, 0, metric="dot")
    assert s.num_rows == N - 2 and sorted(s.source_ids) == [1, 2]  # skipped / hidden rows never enter the index
    q = rng.standard_normal(D).astype(np.float32)
    keep = np.array([i for i in range(N) if i not in (5, 6)])
    order = np.concatenate([keep[src[keep] == 1], keep[src[keep] == 2]])
    for sources in ([1], [2], [1, 2]):
        got = pa.search_vector_and_retrieve(s, db, sources, 10, q)
        oi, od = oracle.search_vector(q, emb[order], 10 + order, src[order], sources, 10)
        assert [it.id for it, _ in got] == list(oi)
        np.testing.assert_allclose([si.score for _, si in got], od, atol=1e-6)
        assert all(item.external_id == f"doc{item.id - 10}.md" and item.metadata.name == f"name {item.id - 10}"
                   for item, _ in got)
    # an item hidden after the index was built is dropped at retrieval (search.rs:209-211)
    rw = sqlite3.connect(path)
    top = pa.search_vector_and_retrieve(s, db, [1, 2], 3, q)[0][0].id
    rw.execute("UPDATE items SET hidden_at = 1 WHERE id = ?", (top,))
    rw.commit()
    assert top not in [it.id for it, _ in pa.search_vector_and_retrieve(s, db, [1, 2], 3, q)]
    # rebuild_source picks the change up (search.rs:58-79)
    pa.rebuild_source(s, db, int(src[top - 10]), 0, 0)
    assert s.num_rows == N - 3
    assert top not in [it.id for it in s.search_vector([1, 2], 50, q)]
    rw.close()
    s.close()


def test_packed_corpus_cache_round_trip(ctx, oracle, tmp_path):
    # a searcher written to the flat cache file and streamed back gives the same rows, ids,
    # sources and search results (three sources, explicit ids, dot metric = the reference Searcher's)
    rng = np.random.default_rng(17)
    D = 96
    s = pa.Searcher(ctx, D, "dot")
    parts = {}
    for sid, n in ((7, 300), (2, 1), (40, 1500)):
        rows = rng.standard_normal((n, D)).astype(np.float32)
        ids = rng.permutation(10_000)[:n].astype(np.int64) + 100_000 * sid
        s.add_rows(sid, rows, ids)
        parts[sid] = (rows, ids)
    s.finalize()
    path = tmp_path / "corpus.pcvs"
    pa.save_searcher_cache(s, path, model_id=3, model_version=1)
    assert path.stat().st_size == 28 + sum(16 + n * 8 + n * D * 4 for n in (300, 1, 1500))
    with pytest.raises(ValueError):
        pa.load_searcher_cache(ctx, path, model_id=4, model_version=1)
    t = pa.load_searcher_cache(ctx, path, model_id=3, model_version=1)
    assert t.metric == "dot" and t.dim == D and t.source_ids == s.source_ids and t.num_rows == s.num_rows
    for sid, (rows, ids) in parts.items():
        assert t.source_num_rows(sid) == rows.shape[0]
    pos = np.arange(s.num_rows)
    r0, i0 = s.get_rows(pos)
    r1, i1 = t.get_rows(pos)
    np.testing.assert_array_equal(r0, r1)
    np.testing.assert_array_equal(i0, i1)
    q = rng.standard_normal((4, D)).astype(np.float32)
    for sources in (None, [40], [2, 7]):
        a, b = s.search_vectors(sources, 8, q), t.search_vectors(sources, 8, q)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    s.close()
    t.close()


def test_calculate_embeddings_pipeline_hook(ctx, golden_dir):
    # calculate_embeddings.rs:9-36: documents -> model.encode in batches -> blobs in input order; the blobs
    # decode to exactly what a single encode of the same documents returns (padding to the batch's longest
    # document does not change a row: masked positions carry no weight)
    tok = pa.BertTokenizer(os.path.join(golden_dir, "tokenizer_vocab.txt"))
    m = pa.Model(ctx, pa.make_desc(tok.vocab_size, 128, 2, 4, 256, 64), synthetic_seed=2, tokenizer=tok)
    docs = ["hello world", "the quick brown fox jumps over the lazy dog", "a", "search the index again and again " * 3,
            "unaffable", "x y z"] * 3
    blobs = pa.calculate_embeddings(m, docs, batch_size=4)
    assert len(blobs) == len(docs) and all(len(b) == 128 * 4 for b in blobs)
    whole = m.encode(docs)
    got = np.stack([pa.deserialize_embedding(b) for b in blobs])
    assert np.abs(got - whole).max() < 1e-6
    assert np.abs(got[:6] - got[6:12]).max() < 1e-6  # same document in another batch (other padding length)
    assert pa.EMBEDDING_BATCH_SIZE == 256
    m.close()

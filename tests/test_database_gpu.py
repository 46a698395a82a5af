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


def test_rebuild_of_one_source_is_atomic(ctx, oracle, tmp_path):
    """search.rs:57-79 builds the new SourceSearch first and swaps it in only on success.  A replacement with one blob of
    the wrong size (or any SQLite error on the way) must leave the old rows of that source searchable, through the
    library's own loader (pcv_searcher_load_sqlite) and through the row-stream form (Searcher.rebuild_source)."""
    rng = np.random.default_rng(3)
    D, N = 64, 300
    emb = rng.standard_normal((N, D)).astype(np.float32)
    path = str(tmp_path / "p.sqlite3")
    conn = sqlite3.connect(path)
    conn.executescript(SCHEMA)
    conn.executemany("INSERT INTO sources (id, name) VALUES (?, ?)", [(1, "a"), (2, "b")])
    src = np.where(np.arange(N) % 3 == 0, 2, 1)
    for i in range(N):
        conn.execute("INSERT INTO items (id, source_id, external_id, content) VALUES (?,?,?,?)", (i, int(src[i]), f"d{i}", "c"))
        conn.execute("INSERT INTO item_embeddings VALUES (0, 0, ?, 1, ?)", (i, pa.serialize_embedding(emb[i])))
    conn.commit()
    db = pa.Database(path)
    s = pa.build_searcher(ctx, db, 0, 0, metric="dot")
    q = rng.standard_normal((5, D)).astype(np.float32)
    before = [s.search_vectors(f, 10, q) for f in (None, [1], [2])]
    # source 2's replacement: 40 good rows, then one blob of the wrong size
    conn.execute("DELETE FROM item_embeddings WHERE item_id IN (SELECT id FROM items WHERE source_id = 2 AND id > 150)")
    conn.execute("UPDATE item_embeddings SET embedding = ? WHERE item_id = 150", (b"\0" * (4 * D - 4),))
    conn.commit()
    with pytest.raises(pa.PcvError, match="bytes"):
        pa.rebuild_source(s, db, 2, 0, 0)
    assert s.num_rows == N and sorted(s.source_ids) == [1, 2]  # nothing of the half-built replacement is left
    for f, ref in zip((None, [1], [2]), before):
        got = s.search_vectors(f, 10, q)
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    # the row-stream form: a row of the wrong width in the middle of the stream
    rows = [(1000 + i, 2, emb[i]) for i in range(20)] + [(9999, 2, emb[0][:-1])] + [(2000 + i, 2, emb[i]) for i in range(20)]
    with pytest.raises(ValueError):
        s.rebuild_source(rows, 2)
    assert s.num_rows == N
    np.testing.assert_array_equal(s.search_vectors([2], 10, q)[0], before[2][0])
    # a good replacement goes in, the other source keeps its place and its hits
    conn.execute("UPDATE item_embeddings SET embedding = ? WHERE item_id = 150", (pa.serialize_embedding(emb[150]),))
    conn.commit()
    pa.rebuild_source(s, db, 2, 0, 0)
    keep2 = np.array([i for i in range(N) if src[i] == 2 and i <= 150])
    assert s.source_num_rows(2) == keep2.size and s.source_num_rows(1) == int((src == 1).sum())
    np.testing.assert_array_equal(s.search_vectors([1], 10, q)[0], before[1][0])
    oi, _ = oracle.search_vector(q[0], emb[keep2], keep2, np.full(keep2.size, 2), [2], 10)
    assert list(s.search_vectors([2], 10, q[:1])[0][0]) == list(oi)
    conn.close()
    s.close()


def test_staged_rows_are_nobodys_rows_until_the_swap(ctx, oracle):
    """Between the staging finalize and pcv_searcher_replace_source the new rows of a source are resident beside the old ones
    (PCV_STAGING_SOURCE).  They are not counted and a search of every source does not see them — the reference swaps a finished
    SourceSearch in one step (search.rs:57-79), so no caller ever sees both generations."""
    rng = np.random.default_rng(8)
    D = 32
    old = rng.standard_normal((200, D)).astype(np.float32)
    new = (old[:50] * 1.5).astype(np.float32)  # the replacement of source 2: scaled copies, they would tie or win under dot
    q = rng.standard_normal((4, D)).astype(np.float32)
    s = pa.Searcher(ctx, D, "dot")
    s.add_rows(1, old[:100], np.arange(100))
    s.add_rows(2, old[100:], np.arange(100, 200))
    s.finalize()
    before = s.search_vectors(None, 10, q)
    s.add_rows(pa.search.STAGING_SOURCE, new, 1000 + np.arange(50))
    s.finalize()  # staged and searchable by id, but nobody's rows yet
    assert s.num_rows == 200 and sorted(s.source_ids) == [1, 2]
    during = s.search_vectors(None, 10, q)
    np.testing.assert_array_equal(during[0], before[0])
    np.testing.assert_array_equal(during[1], before[1])
    staged = s.search_vectors([pa.search.STAGING_SOURCE], 5, q)  # named explicitly they can be searched (the loaders do not)
    assert (staged[0] >= 1000).all()
    from perceive_amd import _ffi
    _ffi.check(_ffi.lib().pcv_searcher_replace_source(s._handle, pa.search.STAGING_SOURCE, 2))
    s.finalize()
    assert s.num_rows == 150 and sorted(s.source_ids) == [1, 2]
    m = np.concatenate([old[:100], new])
    ids = np.concatenate([np.arange(100), 1000 + np.arange(50)])
    got = s.search_vectors(None, 10, q)
    for b in range(4):
        oi, _ = oracle.search_vector(q[b], m, ids, np.concatenate([np.full(100, 1), np.full(50, 2)]), [1, 2], 10)
        assert list(got[0][b]) == list(oi)
    s.close()


def test_replace_source_swaps_and_keeps_the_order_of_sources(ctx, oracle):
    """pcv_searcher_replace_source: `from` takes `to`'s place (global positions of the other sources do not move), an unknown
    or empty `from` leaves `to` absent (search.rs:67-69), a source cannot replace itself."""
    from perceive_amd import _ffi

    rng = np.random.default_rng(5)
    D = 32
    a, b, c, n = (rng.standard_normal((m, D)).astype(np.float32) for m in (50, 70, 30, 40))
    s = pa.Searcher(ctx, D, "cosine")
    s.add_rows(1, a, np.arange(50))
    s.add_rows(2, b, 100 + np.arange(70))
    s.add_rows(3, c, 200 + np.arange(30))
    s.add_rows(pa.search.STAGING_SOURCE, n, 300 + np.arange(40))
    s.finalize()
    _ffi.check(_ffi.lib().pcv_searcher_replace_source(s._handle, pa.search.STAGING_SOURCE, 2))
    with pytest.raises(pa.PcvError):  # dirty until finalize
        s.search_vectors(None, 5, a[:1])
    s.finalize()
    assert s.source_ids == [1, 2, 3] and s.num_rows == 120
    rows, ids = s.get_rows(np.arange(120))
    np.testing.assert_array_equal(rows, np.concatenate([a, n, c]))  # source 2's new rows sit where its old ones did
    np.testing.assert_array_equal(ids, np.concatenate([np.arange(50), 300 + np.arange(40), 200 + np.arange(30)]))
    q = rng.standard_normal((3, D)).astype(np.float32)
    np.testing.assert_array_equal(s.search_vectors(None, 7, q)[0], ids[oracle.topk(q, rows, 7)[0]])
    _ffi.check(_ffi.lib().pcv_searcher_replace_source(s._handle, 777, 3))  # unknown `from`: source 3 disappears
    s.finalize()
    assert s.source_ids == [1, 2] and s.num_rows == 90
    assert _ffi.lib().pcv_searcher_replace_source(s._handle, 1, 1) != 0
    s.close()

"""The slice of perceive-core's SQLite storage the hot path touches (crates/perceive-core/db.rs,
search.rs:38-113,195-259): reading `(items.id, source_id, embedding)` rows for a model version and
hydrating result items.  Storage itself is out of scope; this module only runs the
reference's read queries against a database the reference's pipeline wrote
(`item_embeddings.embedding` = little-endian f32 blob, update_db.rs:118-126).
"""
import sqlite3
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .search import SearchItem, Searcher

_CACHE_CHUNK = 1 << 18  # rows per read / upload step (ingestion and the packed corpus cache)


@dataclass
class ItemMetadata:  # lib.rs:14-21
    name: Optional[str] = None
    author: Optional[str] = None
    description: Optional[str] = None
    mtime: Optional[int] = None
    atime: Optional[int] = None


@dataclass
class Item:  # lib.rs:49-61
    id: int
    source_id: int
    external_id: str
    hash: Optional[str] = None
    content: Optional[str] = None
    raw_content: Optional[bytes] = None
    process_version: int = 0
    metadata: ItemMetadata = field(default_factory=ItemMetadata)
    skipped: Optional[str] = None


class Database:
    """db.rs:43-59 reduced to a read connection."""

    def __init__(self, path):
        self.path = None if path == ":memory:" else str(path)
        self.conn = sqlite3.connect(f"file:{path}?mode=ro", uri=True) if path != ":memory:" else sqlite3.connect(path)

    @classmethod
    def from_connection(cls, conn):
        d = cls.__new__(cls)
        d.conn = conn
        d.path = None  # no file the library could open itself: rows are streamed from this connection
        return d


_ROWS_SQL = """SELECT items.id, source_id, embedding
        FROM items
        JOIN item_embeddings ie ON model_id=? AND model_version=? AND ie.item_id=items.id
        WHERE skipped IS NULL AND hidden_at IS NULL"""  # search.rs:87-93


def _embedding_dim(conn, model_id, model_version):
    row = conn.execute("SELECT length(embedding) FROM item_embeddings WHERE model_id=? AND model_version=? LIMIT 1",
                       (model_id, model_version)).fetchone()
    return None if row is None else row[0] // 4


def _load(searcher, conn, model_id, model_version, sources):
    """build_sources (search.rs:81-155): stream the join, keep rows of `sources`, group by source."""
    src_set = set(int(s) for s in sources)
    by_source = {int(s): ([], bytearray()) for s in sources}

    def flush(source_id):
        ids, blobs = by_source[source_id]
        if ids:
            searcher.add_blobs(source_id, bytes(blobs), len(ids), np.asarray(ids, dtype=np.int64))
            by_source[source_id] = ([], bytearray())

    for item_id, source_id, blob in conn.execute(_ROWS_SQL, (model_id, model_version)):
        if source_id not in src_set:  # search.rs:106-109
            continue
        if len(blob) != searcher.dim * 4:
            raise ValueError(f"embedding of item {item_id} has {len(blob)} bytes, index is {searcher.dim}-d")
        ids, blobs = by_source[source_id]
        ids.append(item_id)
        blobs.extend(blob)
        if len(ids) >= _CACHE_CHUNK:  # hand full chunks over as they arrive: no second whole-corpus copy on the Python side
            flush(source_id)
    for source_id in list(by_source):
        flush(source_id)


def _load_sqlite(searcher, path, model_id, model_version, only_source):
    """build_sources (search.rs:81-155) inside the library: SQL, blob streaming, finalize."""
    import ctypes as C

    from . import _ffi

    src = None if only_source is None else np.array([only_source], dtype=np.int64)
    n = C.c_int64()
    _ffi.check(_ffi.lib().pcv_searcher_load_sqlite(searcher._handle, str(path).encode(), int(model_id), int(model_version),
                                                   None if src is None else _ffi.i64p(src), C.byref(n)))
    return n.value


def build_searcher(ctx, database, model_id, model_version, metric="dot", dim=None):
    """Searcher::build(database, model_id, model_version) — search.rs:38-56."""
    conn = database.conn
    sources = [r[0] for r in conn.execute("SELECT id FROM sources")]  # search.rs:45-48
    dim = dim or _embedding_dim(conn, model_id, model_version)
    if dim is None:
        raise ValueError("no embeddings stored for this model version; pass dim= to build an empty index")
    s = Searcher(ctx, dim, metric)
    if database.path is not None:  # the library reads the file itself (pcv_searcher_load_sqlite)
        _load_sqlite(s, database.path, model_id, model_version, None)
        return s
    _load(s, conn, model_id, model_version, sources)
    s.finalize()
    return s


def rebuild_source(searcher, database, source_id, model_id, model_version):
    """Searcher::rebuild_source — search.rs:58-79."""
    from . import _ffi

    if database.path is not None:
        _load_sqlite(searcher, database.path, model_id, model_version, int(source_id))
        return
    # in-memory database: the same staging as Searcher.rebuild_source (search.rs:57-79: build first, swap on success)
    searcher.rebuild_source(database.conn.execute(_ROWS_SQL, (model_id, model_version)), source_id)


def search_vector_and_retrieve(searcher, database, sources, num_results, vector):
    """search.rs:195-247: search, hydrate the items that are still visible, re-sort ascending."""
    items = searcher.search_vector(sources, num_results, vector)
    if not items:
        return []
    by_id = {it.id: it for it in items}
    marks = ",".join("?" * len(items))  # `id IN rarray(?)`
    rows = database.conn.execute(
        f"""SELECT id, source_id, external_id, content, name, author, description, modified, last_accessed
            FROM items WHERE skipped is NULL AND hidden_at IS NULL AND id IN ({marks})""", [it.id for it in items])
    out = []
    for r in rows:
        item = Item(id=r[0], source_id=r[1], external_id=r[2], content=r[3],
                    metadata=ItemMetadata(name=r[4], author=r[5], description=r[6], mtime=r[7], atime=r[8]))
        out.append((item, by_id[item.id]))
    out.sort(key=lambda p: p[1].score)  # search.rs:245 (cosine searchers: use reverse order at the call site)
    return out


def search_and_retrieve(searcher, database, model, sources, num_results, query):
    """search.rs:249-259"""
    from .search import encode_query

    return search_vector_and_retrieve(searcher, database, sources, num_results, encode_query(model, query))


# ---- packed corpus cache (start-up as a straight H2D stream instead of SQL + decode) ---------
_CACHE_MAGIC = b"PCVS0001"


def save_searcher_cache(searcher: Searcher, path, model_id=0, model_version=0):
    """Write the searcher's rows, per source, to a flat little-endian file:
    magic, dim i32, metric i32 (0 cosine / 1 dot), model_id u32, model_version u32, n_sources i32, then per
    source: source_id i64, n i64, ids i64[n], rows f32[n][dim] — the blob format of search.rs:288-294, row
    after row.  The key (model_id, model_version) is the one Searcher::build is called with."""
    metric = {"cosine": 0, "dot": 1}[searcher.metric]
    sources = searcher.source_ids
    with open(path, "wb") as f:
        f.write(_CACHE_MAGIC)
        f.write(np.array([searcher.dim, metric], dtype="<i4").tobytes())
        f.write(np.array([model_id, model_version], dtype="<u4").tobytes())
        f.write(np.array([len(sources)], dtype="<i4").tobytes())
        pos = 0
        for sid in sources:
            n = searcher.source_num_rows(sid)
            f.write(np.array([sid, n], dtype="<i8").tobytes())
            ids_at = f.tell()
            f.write(b"\0" * (8 * n))  # ids, filled in after the rows have been streamed
            all_ids = np.empty(n, dtype="<i8")
            for r0 in range(0, n, _CACHE_CHUNK):
                r1 = min(n, r0 + _CACHE_CHUNK)
                rows, ids = searcher.get_rows(np.arange(pos + r0, pos + r1, dtype=np.int64))
                all_ids[r0:r1] = ids
                f.write(np.ascontiguousarray(rows, dtype="<f4").tobytes())
            end = f.tell()
            f.seek(ids_at)
            f.write(all_ids.tobytes())
            f.seek(end)
            pos += n


def load_searcher_cache(ctx, path, model_id=None, model_version=None) -> Searcher:
    """Rebuild a Searcher from `save_searcher_cache` output: the file is memory-mapped and each source goes up
    in chunks through `add_rows` (pinned staging inside the library), no decode, no SQL.  If model_id /
    model_version are given they must match the key stored in the file."""
    with open(path, "rb") as f:
        if f.read(8) != _CACHE_MAGIC:
            raise ValueError(f"{path}: not a perceive-hip corpus cache")
        dim, metric = np.frombuffer(f.read(8), dtype="<i4")
        mid, mver = np.frombuffer(f.read(8), dtype="<u4")
        (nsrc,) = np.frombuffer(f.read(4), dtype="<i4")
    if model_id is not None and (int(mid), int(mver)) != (int(model_id), int(model_version or 0)):
        raise ValueError(f"{path}: cache holds model ({mid}, {mver}), wanted ({model_id}, {model_version})")
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    off = 28
    s = Searcher(ctx, int(dim), "dot" if metric == 1 else "cosine")
    for _ in range(int(nsrc)):
        sid, n = (int(x) for x in np.frombuffer(mm[off:off + 16], dtype="<i8"))
        off += 16
        ids = np.frombuffer(mm[off:off + 8 * n], dtype="<i8")
        off += 8 * n
        rows = np.frombuffer(mm[off:off + 4 * n * int(dim)], dtype="<f4").reshape(n, int(dim))
        off += 4 * n * int(dim)
        for r0 in range(0, n, _CACHE_CHUNK):
            s.add_rows(sid, rows[r0:r0 + _CACHE_CHUNK], ids[r0:r0 + _CACHE_CHUNK])
    s.finalize()
    return s


# ---- pipeline hook (sources/pipeline/calculate_embeddings.rs:9-36) --------------------------------------------
EMBEDDING_BATCH_SIZE = 256  # the reference batches 64 documents for its CPU model (pipeline.rs:76); a 256 x 256
#                             token batch is what keeps an MI355X's matrix cores busy


def calculate_embeddings(model, documents, batch_size=EMBEDDING_BATCH_SIZE):
    """`calculate_embeddings_batch` for a list of documents: `model.encode` in batches of `batch_size`, each
    embedding returned as the little-endian f32 blob the reference stores in `item_embeddings.embedding`
    (update_db.rs:118-126).  Order follows the input, like the reference's drain/zip."""
    from .search import serialize_embedding

    out = []
    for i in range(0, len(documents), batch_size):
        emb = model.encode(documents[i:i + batch_size])
        out.extend(serialize_embedding(row) for row in emb)
    return out

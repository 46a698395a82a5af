//! `Searcher` over libperceive_hip.so — replaces search.rs of the reference: the per-source HNSW graphs
//! become one exact, GPU-resident scan (recall 1.0), everything callers see stays:
//!   `SearchItem { id, score }`, `Searcher::{build, rebuild_source, search_vector, search,
//!   search_vector_and_retrieve, search_and_retrieve}`, pub field `hidden`, `encode_query`,
//!   `deserialize_embedding`, `serialize_embedding`                               (search.rs:18-294)
//! Scores keep the reference's convention: `max(0, 1 - dot/len)`, ascending (PCV_METRIC_DOT).
//! NOT COMPILED in this repository's build image (no Rust toolchain): see ../README.md.
use std::ptr;
use std::rc::Rc;

use ahash::HashSet;
use time::OffsetDateTime;

use crate::{
    db::{Database, DbError},
    ffi,
    hip::{self, HipError},
    model::Model,
    Item, ItemMetadata,
};

#[derive(Debug, Copy, Clone)]
pub struct SearchItem {
    pub id: i64,
    pub score: f32,
}

pub struct Searcher {
    handle: *mut ffi::pcv_searcher,
    /// ids hidden after the index was built (`perceive hide`, cmd/hide.rs:17).  Kept as the pub field it
    /// is in the reference (search.rs:31-34), whose `search_vector` does not consult it either.
    pub hidden: HashSet<i64>,
}

// `&self` searches run concurrently from a thread pool in the callers (app_state.rs:52-57); the library
// serialises them on the handle
unsafe impl Send for Searcher {}
unsafe impl Sync for Searcher {}

/// rows per `pcv_searcher_add_blobs` call while streaming the build query
const BUILD_CHUNK_ROWS: usize = 8192;

impl Searcher {
    pub fn build(database: &Database, model_id: u32, model_version: u32) -> Result<Searcher, eyre::Report> {
        let conn = database.read_pool.get()?;

        let mut sources_stmt = conn.prepare("SELECT id FROM sources")?;
        let sources = sources_stmt
            .query_map([], |row| row.get::<_, i64>(0))?
            .collect::<Result<Vec<_>, _>>()?;

        let mut searcher = Searcher { handle: ptr::null_mut(), hidden: HashSet::default() };
        searcher.load_sources(&conn, model_id, model_version, &sources, None)?;
        Ok(searcher)
    }

    pub fn rebuild_source(
        &mut self,
        database: &Database,
        source_id: i64,
        model_id: u32,
        model_version: u32,
    ) -> Result<(), eyre::Report> {
        let conn = database.read_pool.get()?;
        if self.handle.is_null() {
            return self.load_sources(&conn, model_id, model_version, &[source_id], None);
        }
        // search.rs:57-79 builds the new SourceSearch first and swaps it in only on success: the replacement is staged
        // under PCV_STAGING_SOURCE; after a failure (a blob of the wrong size, an SQLite error) the old rows of the
        // source are still searchable.
        hip::check(unsafe { ffi::pcv_searcher_clear_source(self.handle, ffi::PCV_STAGING_SOURCE) })?;
        if let Err(e) = self.load_sources(&conn, model_id, model_version, &[source_id], Some(ffi::PCV_STAGING_SOURCE)) {
            unsafe {
                ffi::pcv_searcher_clear_source(self.handle, ffi::PCV_STAGING_SOURCE);
                ffi::pcv_searcher_finalize(self.handle);
            }
            return Err(e);
        }
        hip::check(unsafe { ffi::pcv_searcher_replace_source(self.handle, ffi::PCV_STAGING_SOURCE, source_id) })?;
        hip::check(unsafe { ffi::pcv_searcher_finalize(self.handle) })?;
        Ok(())
    }

    /// build_sources (search.rs:81-155): the same query, its blobs streamed into the device segments of
    /// their sources instead of into HNSW inserts.  The index is created at the first row, when the
    /// embedding width is known.
    fn load_sources(
        &mut self,
        conn: &rusqlite::Connection,
        model_id: u32,
        model_version: u32,
        sources: &[i64],
        into: Option<i64>, // Some(id): every row goes to that (staging) source instead of its own
    ) -> Result<(), eyre::Report> {
        let mut stmt = conn.prepare(
            r##"SELECT items.id, source_id, embedding
        FROM items
        JOIN item_embeddings ie ON model_id=? AND model_version=? AND ie.item_id=items.id
        WHERE skipped IS NULL AND hidden_at IS NULL"##,
        )?;

        // (ids, blob bytes) waiting to go up, per source
        let mut pending: Vec<(Vec<i64>, Vec<u8>)> = sources.iter().map(|_| (Vec::new(), Vec::new())).collect();
        let mut rows = stmt.query([model_id, model_version])?;
        while let Some(row) = rows.next()? {
            let id: i64 = row.get(0)?;
            let source_id: i64 = row.get(1)?;
            let Some(source_idx) = sources.iter().position(|&s| s == source_id) else {
                continue;
            };
            let blob = row.get_ref(2)?.as_blob().map_err(DbError::query)?;
            if self.handle.is_null() {
                let ctx = hip::context()?;
                hip::check(unsafe {
                    ffi::pcv_searcher_create(ctx.0, (blob.len() / 4) as i32, ffi::PCV_METRIC_DOT, &mut self.handle)
                })?;
            }
            let (ids, bytes) = &mut pending[source_idx];
            ids.push(id);
            bytes.extend_from_slice(blob);
            if ids.len() >= BUILD_CHUNK_ROWS {
                self.flush(into.unwrap_or(source_id), ids, bytes)?;
            }
        }
        for (source_idx, (ids, bytes)) in pending.iter_mut().enumerate() {
            self.flush(into.unwrap_or(sources[source_idx]), ids, bytes)?;
        }
        if !self.handle.is_null() {
            hip::check(unsafe { ffi::pcv_searcher_finalize(self.handle) })?;
        }
        Ok(())
    }

    fn flush(&self, source_id: i64, ids: &mut Vec<i64>, bytes: &mut Vec<u8>) -> Result<(), HipError> {
        if ids.is_empty() {
            return Ok(());
        }
        hip::check(unsafe {
            ffi::pcv_searcher_add_blobs(self.handle, source_id, ids.as_ptr(), bytes.as_ptr(), ids.len() as i64)
        })?;
        ids.clear();
        bytes.clear();
        Ok(())
    }

    pub fn search_vector(&self, sources: &[i64], num_results: usize, vector: Vec<f32>) -> Vec<SearchItem> {
        if self.handle.is_null() || num_results == 0 {
            return Vec::new();
        }
        // the C ABI reads `dim` floats from the pointer: a vector of another width must not reach it (the reference
        // works on slices and panics inside hnsw_rs on a width mismatch; so does this)
        let mut dim: i32 = 0;
        hip::check(unsafe { ffi::pcv_searcher_dim(self.handle, &mut dim) }).expect("searcher_dim failed");
        assert_eq!(vector.len(), dim as usize, "search_vector: the query has {} values, the index is {}-d", vector.len(), dim);
        // num_results is not limited (search.rs:157-182; perceive-cli's --num-results is user input): beyond the
        // PCV_MAX_RESULTS (128) hits one pass over the rows ranks, the library goes over them again for the next 128.
        let k = num_results;
        let mut ids = vec![-1i64; k];
        let mut scores = vec![f32::NAN; k];
        let mut count: i32 = 0;
        // `sources.as_ptr()` of an empty slice is non-null and n_sources = 0 matches nothing, like
        // `sources.contains(..)` at search.rs:166
        hip::check(unsafe {
            ffi::pcv_searcher_search(
                self.handle,
                vector.as_ptr(),
                1,
                sources.as_ptr(),
                sources.len() as i32,
                k as i32,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut count,
            )
        })
        .expect("search failed"); // the reference unwraps here too (NaN scores panic at search.rs:179)
        (0..count as usize).map(|i| SearchItem { id: ids[i], score: scores[i] }).collect()
    }

    pub fn search(&self, model: &Model, sources: &[i64], num_results: usize, query: &str) -> Vec<SearchItem> {
        let term_embedding = encode_query(model, query);
        self.search_vector(sources, num_results, term_embedding)
    }

    /// search.rs:195-247: the hits of `search_vector`, each with its `Item` read back from the database.
    /// Items hidden or skipped since the index was built drop out here (the SQL filters them); the result
    /// keeps the order of the hits (ascending score).
    pub fn search_vector_and_retrieve(
        &self,
        database: &Database,
        sources: &[i64],
        num_results: usize,
        vector: Vec<f32>,
    ) -> Result<Vec<(Item, SearchItem)>, DbError> {
        let hits = self.search_vector(sources, num_results, vector);
        let wanted = Rc::new(hits.iter().map(|h| rusqlite::types::Value::from(h.id)).collect::<Vec<_>>());

        let conn = database.read_pool.get()?;
        let mut stmt = conn.prepare_cached(HYDRATE_SQL)?;
        let mut by_id: ahash::HashMap<i64, Item> = ahash::HashMap::default();
        for item in stmt.query_map([wanted], item_from_row)? {
            let item = item?;
            by_id.insert(item.id, item);
        }
        // `hits` is already sorted the way the caller wants it
        Ok(hits.into_iter().filter_map(|hit| by_id.remove(&hit.id).map(|item| (item, hit))).collect())
    }

    pub fn search_and_retrieve(
        &self,
        database: &Database,
        model: &Model,
        sources: &[i64],
        num_results: usize,
        query: &str,
    ) -> Result<Vec<(Item, SearchItem)>, DbError> {
        self.search_vector_and_retrieve(database, sources, num_results, encode_query(model, query))
    }
}

/// the read query of search.rs:209-211 (the `rarray` virtual table is registered on the read pool, db.rs:79-85)
const HYDRATE_SQL: &str = "SELECT id, source_id, external_id, content, name, author, description, modified, last_accessed \
     FROM items WHERE skipped is NULL AND hidden_at IS NULL AND id IN rarray(?)";

fn unix_time(seconds: Option<i64>) -> Option<OffsetDateTime> {
    seconds.map(|t| OffsetDateTime::from_unix_timestamp(t).unwrap())
}

fn item_from_row(row: &rusqlite::Row<'_>) -> rusqlite::Result<Item> {
    let metadata = ItemMetadata {
        name: row.get(4)?,
        author: row.get(5)?,
        description: row.get(6)?,
        mtime: unix_time(row.get(7)?),
        atime: unix_time(row.get(8)?),
    };
    Ok(Item {
        id: row.get(0)?,
        source_id: row.get(1)?,
        external_id: row.get(2)?,
        content: row.get(3)?,
        raw_content: None,
        hash: None,
        skipped: None,
        process_version: 0,
        metadata,
    })
}

impl Drop for Searcher {
    fn drop(&mut self) {
        if !self.handle.is_null() {
            unsafe { ffi::pcv_searcher_destroy(self.handle) };
        }
    }
}

pub fn encode_query(model: &Model, query: &str) -> Vec<f32> {
    Vec::<Vec<f32>>::from(model.encode(&[query]).unwrap()).pop().unwrap()
}

pub fn deserialize_embedding(value: &[u8]) -> Vec<f32> {
    let mut out = vec![0f32; value.len() / 4];
    let mut n: usize = 0;
    hip::check(unsafe { ffi::pcv_deserialize_embedding(value.as_ptr(), value.len(), out.as_mut_ptr(), out.len(), &mut n) })
        .expect("embedding blob is not a whole number of f32"); // the reference panics on a short chunk
    out.truncate(n);
    out
}

pub fn serialize_embedding(embedding: &[f32]) -> Vec<u8> {
    let mut bytes_vec = vec![0u8; embedding.len() * std::mem::size_of::<f32>()];
    hip::check(unsafe {
        ffi::pcv_serialize_embedding(embedding.as_ptr(), embedding.len(), bytes_vec.as_mut_ptr(), bytes_vec.len())
    })
    .expect("serialize_embedding");
    bytes_vec
}

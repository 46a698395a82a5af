//! The crate-level similarity functions of `crates/perceive-core/lib.rs:63-77`, over the C ABI.
//! Paste into `lib.rs` in place of the three `tch` functions (and drop `use tch::{Kind, Tensor};`), or keep
//! as a module and `pub use similarity::*;` there.  The reference took and returned `tch::Tensor`s; its only
//! in-tree caller is `model/highlight.rs:109`, which the shim's `Model::highlight` no longer needs (the chunk
//! scores are computed on the device by `pcv_model_highlight`).  `Embeddings` is the shim's `[n, dim]` f32
//! matrix (model.rs), what `Model::encode` returns.
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain); checked by tests/test_rust_shim.py.
use crate::ffi;
use crate::hip::{self, HipError};
use crate::model::Embeddings;

fn similarity(set1: &Embeddings, set2: &Embeddings, cosine: bool) -> Result<Embeddings, HipError> {
    assert_eq!(set1.dim, set2.dim, "similarity: {}-d against {}-d vectors", set1.dim, set2.dim);
    let (b, n) = (set1.len(), set2.len());
    let mut out = Embeddings { dim: n, data: vec![0f32; b * n] };
    if b == 0 || n == 0 {
        return Ok(out);
    }
    let ctx = hip::context()?;
    let (a, m, o) = (set1.data.as_ptr(), set2.data.as_ptr(), out.data.as_mut_ptr());
    hip::check(unsafe {
        if cosine {
            ffi::pcv_cosine_similarity(ctx.0, a, b as i32, m, n as i64, set1.dim as i32, o)
        } else {
            ffi::pcv_dot_product(ctx.0, a, b as i32, m, n as i64, set1.dim as i32, o)
        }
    })?;
    Ok(out)
}

/// lib.rs:63-65 — `set1.matmul(set2.T)`: `[len(set1), len(set2)]`, row i = dot products of set1[i] with every row of set2.
pub fn dot_product(set1: &Embeddings, set2: &Embeddings) -> Embeddings {
    similarity(set1, set2, false).expect("dot_product failed") // the tch functions panic on a device error too
}

/// lib.rs:67-71 — one query (`query.len() == 1`) against `matches`: `[1, len(matches)]` cosines.  No epsilon in the
/// divide, as in the reference: a zero row gives NaN.
pub fn cosine_similarity_single_query(query: &Embeddings, matches: &Embeddings) -> Embeddings {
    assert_eq!(query.len(), 1, "cosine_similarity_single_query: one query vector expected");
    similarity(query, matches, true).expect("cosine_similarity_single_query failed")
}

/// lib.rs:73-77 — `[len(set1), len(set2)]` cosines.
pub fn cosine_similarity_multi_query(set1: &Embeddings, set2: &Embeddings) -> Embeddings {
    similarity(set1, set2, true).expect("cosine_similarity_multi_query failed")
}

//! `Model` over libperceive_hip.so — replaces model.rs, model/worker.rs, model/tokenize.rs and
//! model/highlight.rs of the reference.  Public surface kept:
//!   `Model::new_pretrained(SentenceEmbeddingsModelType) -> Result<Model, ModelError>`   (model.rs:68)
//!   `Model::encode(&[S]) -> Result<Embeddings, ModelError>`, `Vec<Vec<f32>>: From<Embeddings>` (model.rs:176)
//!   `Model::highlight(&self, &str, &[S]) -> Result<Vec<Option<&str>>, ModelError>`      (highlight.rs:23)
//!   `Model: Send + Sync`, pub field `model_type`                                        (model.rs:56-65)
//! The worker thread and its channels are gone: the library serialises forwards on the handle.
//! NOT COMPILED in this repository's build image (no Rust toolchain): see ../README.md.
mod configs;

use std::ffi::CString;
use std::os::raw::c_char;
use std::ptr;

use thiserror::Error;

pub use configs::SentenceEmbeddingsModelType;

use crate::ffi;
use crate::hip::{self, HipError};

#[derive(Debug, Error)]
pub enum ModelError {
    /// kept for callers that match on it: the device forward failed (worker.rs:71-75 reported a panic here)
    #[error("Model error: {0}")]
    ModelPanic(eyre::Report),

    #[error(transparent)]
    Hip(#[from] HipError),
}

/// What `Model::encode` returns: `[n, dim]` f32, row-major.  The reference returned a `tch::Tensor`; its
/// callers only ever convert it (calculate_embeddings.rs:21, search.rs:263).
#[derive(Debug, Clone)]
pub struct Embeddings {
    pub dim: usize,
    pub data: Vec<f32>,
}

impl Embeddings {
    pub fn len(&self) -> usize {
        if self.dim == 0 { 0 } else { self.data.len() / self.dim }
    }
    pub fn is_empty(&self) -> bool {
        self.data.is_empty()
    }
}

impl From<Embeddings> for Vec<Vec<f32>> {
    fn from(e: Embeddings) -> Self {
        if e.dim == 0 {
            return Vec::new();
        }
        e.data.chunks(e.dim).map(|row| row.to_vec()).collect()
    }
}

pub struct Model {
    pub model_type: SentenceEmbeddingsModelType,
    handle: *mut ffi::pcv_model,
    output_dim: usize,
}

// calls on one handle serialise inside the library (model.rs:161,187 did it with a bounded channel)
unsafe impl Send for Model {}
unsafe impl Sync for Model {}

impl Model {
    pub fn new_pretrained(model_type: SentenceEmbeddingsModelType) -> Result<Model, ModelError> {
        let ctx = hip::context()?;
        let dir = model_type.directory();
        let c_dir = CString::new(dir.to_string_lossy().as_bytes()).map_err(|e| ModelError::ModelPanic(eyre::eyre!(e)))?;
        let mut handle: *mut ffi::pcv_model = ptr::null_mut();
        // weights: rust_model.ot (configs.rs:109), else model.safetensors / pytorch_model.bin, read by the library
        hip::check(unsafe { ffi::pcv_model_create_from_dir(ctx.0, c_dir.as_ptr(), ffi::PCV_COMPUTE_F32, 1, &mut handle) })?;
        let mut dim: i32 = 0;
        if let Err(e) = hip::check(unsafe { ffi::pcv_model_output_dim(handle, &mut dim) }) {
            unsafe { ffi::pcv_model_destroy(handle) };
            return Err(e.into());
        }
        Ok(Model { model_type, handle, output_dim: dim as usize })
    }

    /// `(pointers, lengths)` of a batch of strings for the C ABI (no copies: the library takes byte lengths)
    fn str_args<S: AsRef<str>>(inputs: &[S]) -> (Vec<*const c_char>, Vec<usize>) {
        let ptrs = inputs.iter().map(|s| s.as_ref().as_ptr() as *const c_char).collect();
        let lens = inputs.iter().map(|s| s.as_ref().len()).collect();
        (ptrs, lens)
    }

    pub fn encode<S: AsRef<str> + Sync>(&self, inputs: &[S]) -> Result<Embeddings, ModelError> {
        let (ptrs, lens) = Self::str_args(inputs);
        let mut data = vec![0f32; inputs.len() * self.output_dim];
        hip::check(unsafe {
            ffi::pcv_model_encode_text(self.handle, ptrs.as_ptr(), lens.as_ptr(), inputs.len() as i32, data.as_mut_ptr())
        })?;
        Ok(Embeddings { dim: self.output_dim, data })
    }

    /// Given a query and a set of matching documents, try to find the chunk of text from each document that
    /// best matches to the query (highlight.rs:23-165; computed by `pcv_model_highlight`).
    pub fn highlight<'s, 'doc, S: AsRef<str> + Sync>(
        &'s self,
        query: &'doc str,
        documents: &'doc [S],
    ) -> Result<Vec<Option<&'doc str>>, ModelError> {
        let (ptrs, lens) = Self::str_args(documents);
        let mut begin = vec![-1i64; documents.len()];
        let mut end = vec![-1i64; documents.len()];
        hip::check(unsafe {
            ffi::pcv_model_highlight(
                self.handle,
                query.as_ptr() as *const c_char,
                query.len(),
                ptrs.as_ptr(),
                lens.as_ptr(),
                documents.len() as i32,
                0,  // CHUNK_SIZE / CHUNK_OVERLAP from the environment, defaults 20 / 4 (highlight.rs:7-18)
                -1,
                begin.as_mut_ptr(),
                end.as_mut_ptr(),
            )
        })?;
        Ok(documents
            .iter()
            .zip(begin.iter().zip(end.iter()))
            .map(|(doc, (&b, &e))| if b < 0 { None } else { Some(&doc.as_ref()[b as usize..e as usize]) })
            .collect())
    }
}

impl Drop for Model {
    fn drop(&mut self) {
        unsafe { ffi::pcv_model_destroy(self.handle) };
    }
}

impl std::fmt::Debug for Model {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("Model").field("model_type", &self.model_type).finish()
    }
}

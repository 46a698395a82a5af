//! `SentenceEmbeddingsModelType` (reference: model/configs.rs:14-83) and where a variant's files live
//! (configs.rs:85-141).  The enum, its derives and `model_id()` are the crate's API and stay as they were;
//! the rust-bert `SentenceEmbeddingsConfig` plumbing is gone: the library reads the model directory itself
//! (`pcv_model_create_from_dir`).
//! NOT COMPILED in this repository's build image (no Rust toolchain): see ../README.md.
use std::ffi::CStr;
use std::path::PathBuf;

use once_cell::sync::Lazy;
use strum::{AsRefStr, Display, EnumIter, EnumString, EnumVariantNames};

use crate::ffi;

/// The supported model types.
#[derive(Debug, Clone, Copy, Hash, Eq, PartialEq, Display, EnumString, AsRefStr, EnumIter, EnumVariantNames)]
#[cfg_attr(feature = "cli", derive(clap::ValueEnum))]
pub enum SentenceEmbeddingsModelType {
    AllMiniLmL6V2,
    AllMiniLmL12V2,
    DistiluseBaseMultilingualCased,
    AllDistilrobertaV1,
    ParaphraseAlbertSmallV2,
    MsMarcoDistilbertDotV5,
    MsMarcoDistilbertBaseTasB,
    MsMarcoBertBaseDotV5,
}

impl SentenceEmbeddingsModelType {
    /// Map the model to the ID in the database.  The reference spells the mapping out as a match
    /// (configs.rs:72-83); its values are the variants' positions in the enum, which is also the index the
    /// library uses (`pcv_model_type_dir_name`).
    pub fn model_id(&self) -> u32 {
        *self as u32
    }

    /// `model_data/<name>` of this variant (configs.rs:42-69,121-141); the names come from the library
    /// (`pcv_model_type_dir_name`) so that both sides agree.
    pub(super) fn directory(&self) -> PathBuf {
        let name = unsafe {
            let p = ffi::pcv_model_type_dir_name(self.model_id() as i32);
            assert!(!p.is_null(), "unknown model type");
            CStr::from_ptr(p).to_string_lossy().into_owned()
        };
        PathBuf::from(Lazy::force(&MODEL_DATA_DIR)).join(name)
    }
}

// TODO (as in the reference) get this from a runtime value indicating where the app was installed
static MODEL_DATA_DIR: Lazy<String> = Lazy::new(|| {
    std::env::var("PERCEIVE_MODEL_DATA").unwrap_or_else(|_| format!("{}model_data", env!("CARGO_WORKSPACE_DIR")))
});

//! Process-wide device context, error plumbing and handle ownership over the C ABI.
//! NOT COMPILED in this repository's build image (no Rust toolchain): see ../README.md.
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

use once_cell::sync::OnceCell;
use thiserror::Error;

use crate::ffi;

/// A failed `pcv_*` call: its status and the library's thread-local message (`pcv_last_error`).
#[derive(Debug, Error, Clone)]
#[error("perceive-hip status {status}: {message}")]
pub struct HipError {
    pub status: i32,
    pub message: String,
}

/// Turn a `pcv_status` into a `Result`.
pub(crate) fn check(status: c_int) -> Result<(), HipError> {
    if status == ffi::PCV_OK {
        return Ok(());
    }
    let message = unsafe {
        let p = ffi::pcv_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    };
    Err(HipError { status, message })
}

/// One context per process: one process drives one GPU (`tch::Device::cuda_if_available()` in the
/// reference, model.rs:117).  `PERCEIVE_HIP_DEVICE` selects the device index (default 0).
pub(crate) struct Context(pub *mut ffi::pcv_ctx);

// The library serialises the calls on a handle itself (include/perceive_hip.h, conventions).
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

static CONTEXT: OnceCell<Context> = OnceCell::new();

pub(crate) fn context() -> Result<&'static Context, HipError> {
    CONTEXT.get_or_try_init(|| {
        let device = std::env::var("PERCEIVE_HIP_DEVICE")
            .ok()
            .and_then(|v| v.parse::<c_int>().ok())
            .unwrap_or(0);
        let mut ctx: *mut ffi::pcv_ctx = ptr::null_mut();
        check(unsafe { ffi::pcv_init(device, &mut ctx) })?;
        Ok(Context(ctx))
    })
}

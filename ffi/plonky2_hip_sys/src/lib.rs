//! FFI to `libplonky2_hip.so`, the MI355X-native prover hot path for plonky2 — the crate that takes the place of
//! the reference's `cuda/` (`plonky2_cuda`).
//!
//! * The seven `extern "C"` functions below are the reference's, name for name and argument for argument
//!   (`cuda/src/lib.rs:58-145`; their semantics, including the device-memory region contract of the three Merkle entry
//!   points, are documented in `include/plonky2_hip.h` section (A)). `Error` and `DataSlice` have the reference's layout
//!   (`cuda/src/lib.rs:20-56`): a by-value `{i32, char*}` whose message was `strdup`ed by the library.
//! * [`bindings`] is the library's own `gl_*` API — 64-bit sizes, strides, real error returns, contexts, the whole
//!   `prove()` — generated from the header by `tools/gen_ffi_rs.py`.
//! * [`HipInnerContext`] stands where `CudaInnerContext {stream, stream2}` stood (`plonky2/src/fri/oracle.rs:43-47`):
//!   `as_ptr()` is what the `ctx: *mut c_void` arguments take.
//!
//! This image has no Rust toolchain, so the crate has not been compiled here; `tests/test_ffi_crate.py` keeps it
//! honest mechanically (every symbol declared here is exported by the library; `bindings.rs` equals the generator's
//! output; argument counts of the seven reference functions equal the header's).
#![allow(clippy::missing_safety_doc, clippy::too_many_arguments)]

use std::ffi::c_void;

pub mod bindings;
pub use bindings::*;

/// `cuda::Error` of the reference (`cuda/src/lib.rs:20-49`): returned by value; `code == 0` is success; a non-null
/// message was allocated with `strdup` and is freed on drop.
#[repr(C)]
pub struct Error {
    pub code: i32,
    str: Option<core::ptr::NonNull<i8>>,
}

impl Error {
    pub fn is_ok(&self) -> bool {
        self.code == 0
    }
    /// `Ok(())` or the library's message (falling back to `cudaGetErrorString(code)`, which the library exports
    /// under that name because the reference's wrapper binds it).
    pub fn into_result(self) -> Result<(), String> {
        if self.code == 0 { Ok(()) } else { Err(String::from(self)) }
    }
}

impl Drop for Error {
    fn drop(&mut self) {
        extern "C" {
            fn free(p: Option<core::ptr::NonNull<i8>>);
        }
        unsafe { free(self.str.take()) };
    }
}

impl From<Error> for String {
    fn from(status: Error) -> Self {
        let c_str = match status.str {
            Some(ptr) => unsafe { std::ffi::CStr::from_ptr(ptr.as_ptr() as *const _) },
            None => unsafe { std::ffi::CStr::from_ptr(cudaGetErrorString(status.code) as *const _) },
        };
        String::from(c_str.to_str().unwrap_or("unintelligible"))
    }
}

/// `DataSlice` of the reference (`cuda/src/lib.rs:52-56`): a host struct holding a device pointer and an i32 length.
#[repr(C)]
pub struct DataSlice {
    pub ptr: *const c_void,
    pub len: i32,
}

extern "C" {
    /// `cuda/src/lib.rs:59`
    pub fn init();

    /// `cuda/src/lib.rs:61-69`
    pub fn ifft(
        values_flatten: *mut u64,
        poly_num: i32,
        values_num_per_poly: i32,
        log_len: i32,
        root_table: *const u64,
        n_inv: *const u64,
        ctx: *mut c_void,
    ) -> Error;

    /// `cuda/src/lib.rs:71-81`
    pub fn build_merkle_tree(
        ext_values_flatten: *mut u64,
        poly_num: i32,
        values_num_per_poly: i32,
        log_len: i32,
        rate_bits: i32,
        salt_size: i32,
        cap_height: i32,
        pad_extvalues_len: i32,
        ctx: *mut c_void,
    ) -> Error;

    /// `cuda/src/lib.rs:83-98`
    pub fn merkle_tree_from_values(
        values_flatten: *mut u64,
        ext_values_flatten: *mut u64,
        poly_num: i32,
        values_num_per_poly: i32,
        log_len: i32,
        root_table: *const u64,
        root_table2: *const u64,
        shift_powers: *const u64,
        n_inv: *const u64,
        rate_bits: i32,
        salt_size: i32,
        cap_height: i32,
        pad_extvalues_len: i32,
        ctx: *mut c_void,
    ) -> Error;

    /// `cuda/src/lib.rs:100-114`
    pub fn merkle_tree_from_coeffs(
        values_flatten: *mut u64,
        ext_values_flatten: *mut u64,
        poly_num: i32,
        values_num_per_poly: i32,
        log_len: i32,
        root_table: *const u64,
        root_table2: *const u64,
        shift_powers: *const u64,
        rate_bits: i32,
        salt_size: i32,
        cap_height: i32,
        pad_extvalues_len: i32,
        ctx: *mut c_void,
    ) -> Error;

    /// `cuda/src/lib.rs:117-143`
    pub fn compute_quotient_polys(
        ext_values_flatten: *const u64,
        poly_num: i32,
        values_num_per_poly: i32,
        log_len: i32,
        root_table2: *const u64,
        shift_inv_powers: *const u64,
        rate_bits: i32,
        salt_size: i32,
        zs_partial_products_commitment_leaves: *const DataSlice,
        constants_sigmas_commitment_leaves: *const DataSlice,
        d_outs: *mut c_void,
        d_quotient_polys: *mut c_void,
        points: *const DataSlice,
        z_h_on_coset_evals: *const DataSlice,
        z_h_on_coset_inverses: *const DataSlice,
        k_is: *const DataSlice,
        alphas: *const DataSlice,
        betas: *const DataSlice,
        gammas: *const DataSlice,
        ctx: *mut c_void,
    ) -> Error;

    /// bound by the reference's `From<Error> for String` (`cuda/src/lib.rs:42-45`)
    pub fn cudaGetErrorString(code: i32) -> *const i8;
}

/// Two HIP streams on one device: the twin of `CudaInnerContext` (`plonky2/src/fri/oracle.rs:43-47`). Every entry
/// point that takes a `ctx` runs on this context's device, whatever device the calling thread had current.
pub struct HipInnerContext(*mut c_void);

unsafe impl Send for HipInnerContext {}

impl HipInnerContext {
    /// One context per GPU; with one process per GPU pass `LOCAL_RANK`.
    pub fn new(device: i32) -> Result<Self, String> {
        let p = unsafe { gl_ctx_create(device) };
        if p.is_null() { Err(format!("gl_ctx_create({device}) failed: no such HIP device?")) } else { Ok(Self(p)) }
    }
    pub fn as_ptr(&self) -> *mut c_void {
        self.0
    }
    pub fn synchronize(&self) -> Result<(), String> {
        unsafe { gl_ctx_synchronize(self.0) }.into_result()
    }
}

impl Drop for HipInnerContext {
    fn drop(&mut self) {
        unsafe { gl_ctx_destroy(self.0) }
    }
}

/// `len` u64 field elements in HBM on the context's device (what `rustacuda::memory::DeviceBuffer<u64>` was on the
/// reference's side of the boundary).
pub struct DeviceBuffer {
    ptr: *mut u64,
    len: usize,
}

unsafe impl Send for DeviceBuffer {}

impl DeviceBuffer {
    pub fn new(ctx: &HipInnerContext, len: usize) -> Result<Self, String> {
        let mut p: *mut c_void = core::ptr::null_mut();
        unsafe { gl_ctx_malloc(&mut p, (len as u64) * 8, ctx.as_ptr()) }.into_result()?;
        Ok(Self { ptr: p as *mut u64, len })
    }
    pub fn from_slice(ctx: &HipInnerContext, host: &[u64]) -> Result<Self, String> {
        let b = Self::new(ctx, host.len())?;
        unsafe { gl_memcpy_h2d(b.ptr as *mut c_void, host.as_ptr() as *const c_void, (host.len() as u64) * 8, ctx.as_ptr()) }
            .into_result()?;
        Ok(b)
    }
    pub fn copy_to(&self, ctx: &HipInnerContext, host: &mut [u64]) -> Result<(), String> {
        assert!(host.len() <= self.len);
        unsafe { gl_memcpy_d2h(host.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, (host.len() as u64) * 8, ctx.as_ptr()) }
            .into_result()
    }
    pub fn as_ptr(&self) -> *const u64 {
        self.ptr
    }
    pub fn as_mut_ptr(&mut self) -> *mut u64 {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
}

impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        let _ = unsafe { gl_free(self.ptr as *mut c_void) };
    }
}

/// The whole of `prove()` (`plonky2/src/plonk/prover.rs:40-233`) on the device: `circuit` from
/// [`bindings::gl_circuit_create`], the full witness `[num_wires][2^degree_bits]` in `d_wires` (the layout of
/// `my_full_witness`, `iop/witness.rs:351-362`). Returns the proof in plonky2's wire format
/// (`ProofWithPublicInputs::to_bytes`, `util/serialization.rs`).
pub fn prove(circuit: *const c_void, d_wires: &DeviceBuffer, public_inputs: &[u64], ctx: &HipInnerContext) -> Result<Vec<u8>, String> {
    let mut proof: *mut u8 = core::ptr::null_mut();
    let mut len: u64 = 0;
    unsafe {
        gl_prove(circuit, d_wires.as_ptr(), public_inputs.as_ptr(), public_inputs.len() as u32, &mut proof, &mut len, core::ptr::null_mut(), ctx.as_ptr())
    }
    .into_result()?;
    let bytes = unsafe { std::slice::from_raw_parts(proof, len as usize) }.to_vec();
    unsafe { gl_bytes_free(proof) };
    Ok(bytes)
}

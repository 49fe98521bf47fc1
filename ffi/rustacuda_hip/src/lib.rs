//! `rustacuda` on the HIP runtime: exactly the subset of rustacuda 0.1.3 that sideprotocol/plonky2-gpu's Rust uses.
//!
//! The reference's prover is written against `rustacuda` (`plonky2/src/fri/oracle.rs:37-38, 43-109, 352-458`,
//! `plonk/prover.rs:37-39, 446-566`, `fri/prover.rs:6, 196-207`, `hash/merkle_tree.rs:9`, `field/src/goldilocks_field.rs:8`):
//! two streams in `CudaInnerContext`, one big `DeviceBuffer<F>` carved into `DeviceSlice`s, pinned host vectors, and
//! `async_copy_from` / `async_copy_to` between them. This crate provides those names with rustacuda's signatures on top of
//! `hipStream*`, `hipMalloc`, `hipHostMalloc` and `hipMemcpy[Async]`, so the reference's sources compile unchanged once
//! `plonky2/Cargo.toml` and `field/Cargo.toml` point the dependency NAMED `rustacuda` here (ffi/patches/plonky2-hip.patch).
//!
//! Layout contract with libplonky2_hip.so: `Stream` is `#[repr(transparent)]` over `hipStream_t`, so the reference's
//! `CudaInnerContext { stream, stream2 }` IS the `{hipStream_t stream, stream2}` that every `ctx: *mut c_void` argument
//! of the library points to (include/plonky2_hip.h, section (A)).
//!
//! Not compiled in this repository's image (no Rust toolchain); see ffi/plonky2_hip_sys for the library's own bindings.
#![allow(clippy::missing_safety_doc)]

#[macro_use]
extern crate bitflags;
extern crate rustacuda_core;
#[allow(unused_imports)]
#[macro_use]
extern crate rustacuda_derive;

use std::ffi::c_void;

pub use rustacuda_core::{DeviceCopy, DevicePointer};
#[doc(hidden)]
pub use rustacuda_derive::*;

// ---------------------------------------------------------------------------------------------- HIP runtime
#[allow(non_camel_case_types)]
type hipStream_t = *mut c_void;
const HIP_MEMCPY_H2D: i32 = 1;
const HIP_MEMCPY_D2H: i32 = 2;

extern "C" {
    fn hipInit(flags: u32) -> i32;
    fn hipGetDeviceCount(count: *mut i32) -> i32;
    fn hipSetDevice(device: i32) -> i32;
    fn hipDeviceSynchronize() -> i32;
    fn hipStreamCreateWithFlags(stream: *mut hipStream_t, flags: u32) -> i32;
    fn hipStreamCreateWithPriority(stream: *mut hipStream_t, flags: u32, priority: i32) -> i32;
    fn hipStreamSynchronize(stream: hipStream_t) -> i32;
    fn hipStreamDestroy(stream: hipStream_t) -> i32;
    fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> i32;
    fn hipFree(ptr: *mut c_void) -> i32;
    fn hipHostMalloc(ptr: *mut *mut c_void, bytes: usize, flags: u32) -> i32;
    fn hipHostFree(ptr: *mut c_void) -> i32;
    fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: i32) -> i32;
    fn hipMemcpyAsync(dst: *mut c_void, src: *const c_void, bytes: usize, kind: i32, stream: hipStream_t) -> i32;
    fn hipGetErrorString(code: i32) -> *const i8;
}

// ---------------------------------------------------------------------------------------------- errors, init
pub mod error {
    /// A HIP error code (rustacuda: an enum; the reference only `unwrap()`s it).
    #[derive(Clone, Copy, PartialEq, Eq)]
    pub struct CudaError(pub i32);
    pub type CudaResult<T> = Result<T, CudaError>;
    impl std::fmt::Debug for CudaError {
        fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
            let s = unsafe { std::ffi::CStr::from_ptr(super::hipGetErrorString(self.0)) };
            write!(f, "hip error {} ({})", self.0, s.to_string_lossy())
        }
    }
    impl std::fmt::Display for CudaError {
        fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
            std::fmt::Debug::fmt(self, f)
        }
    }
    impl std::error::Error for CudaError {}
    pub(crate) fn check(code: i32) -> CudaResult<()> {
        if code == 0 { Ok(()) } else { Err(CudaError(code)) }
    }
}
use error::{check, CudaResult};

bitflags! {
    /// `rustacuda::CudaFlags` (none are defined by CUDA either)
    pub struct CudaFlags: u32 { const _ZERO = 0; }
}

/// `rustacuda::init`
pub fn init(flags: CudaFlags) -> CudaResult<()> {
    check(unsafe { hipInit(flags.bits()) })
}

/// `rustacuda::quick_init`: initialise, take device 0, create a context
pub fn quick_init() -> CudaResult<context::Context> {
    init(CudaFlags::empty())?;
    let device = device::Device::get_device(0)?;
    context::Context::create_and_push(context::ContextFlags::MAP_HOST | context::ContextFlags::SCHED_AUTO, device)
}

// ---------------------------------------------------------------------------------------------- device, context
pub mod device {
    use super::*;
    /// `rustacuda::device::Device`: an ordinal
    #[derive(Clone, Copy, Debug)]
    pub struct Device(pub(crate) i32);
    impl Device {
        pub fn num_devices() -> CudaResult<u32> {
            let mut n = 0i32;
            check(unsafe { hipGetDeviceCount(&mut n) })?;
            Ok(n as u32)
        }
        pub fn get_device(ordinal: u32) -> CudaResult<Device> {
            if ordinal >= Self::num_devices()? { return Err(error::CudaError(101)); } // hipErrorInvalidDevice
            Ok(Device(ordinal as i32))
        }
        pub fn ordinal(&self) -> i32 { self.0 }
    }
}

pub mod context {
    use super::*;
    bitflags! {
        pub struct ContextFlags: u32 {
            const SCHED_AUTO = 0x00;
            const SCHED_SPIN = 0x01;
            const SCHED_YIELD = 0x02;
            const SCHED_BLOCKING_SYNC = 0x04;
            const MAP_HOST = 0x08;
            const LMEM_RESIZE_TO_MAX = 0x10;
        }
    }
    /// HIP has one primary context per device: "creating and pushing" a context selects the device for this thread.
    /// Threads that did not create the `Context` must call [`CurrentContext::set_current`] (or hipSetDevice) themselves,
    /// exactly as with CUDA's driver API — and libplonky2_hip.so makes the ctx's device current inside every entry point.
    #[derive(Debug)]
    pub struct Context { device: i32 }
    impl Context {
        pub fn create_and_push(_flags: ContextFlags, device: device::Device) -> CudaResult<Context> {
            check(unsafe { hipSetDevice(device.0) })?;
            Ok(Context { device: device.0 })
        }
        pub fn get_unowned(&self) -> UnownedContext { UnownedContext { device: self.device } }
    }
    #[derive(Debug, Clone)]
    pub struct UnownedContext { device: i32 }
    pub struct CurrentContext;
    impl CurrentContext {
        pub fn set_current(c: &Context) -> CudaResult<()> { check(unsafe { hipSetDevice(c.device) }) }
        pub fn set_current_unowned(c: &UnownedContext) -> CudaResult<()> { check(unsafe { hipSetDevice(c.device) }) }
        pub fn synchronize() -> CudaResult<()> { check(unsafe { hipDeviceSynchronize() }) }
    }
}

// ---------------------------------------------------------------------------------------------- streams
pub mod stream {
    use super::*;
    bitflags! {
        pub struct StreamFlags: u32 {
            const DEFAULT = 0x00;
            const NON_BLOCKING = 0x01;
        }
    }
    /// `rustacuda::stream::Stream`. `repr(transparent)` over the raw `hipStream_t`: see the crate documentation.
    #[repr(transparent)]
    #[derive(Debug)]
    pub struct Stream { inner: hipStream_t }
    unsafe impl Send for Stream {}
    unsafe impl Sync for Stream {}
    impl Stream {
        pub fn new(flags: StreamFlags, priority: Option<i32>) -> CudaResult<Stream> {
            let mut s: hipStream_t = std::ptr::null_mut();
            check(unsafe {
                match priority {
                    Some(p) => hipStreamCreateWithPriority(&mut s, flags.bits(), p),
                    None => hipStreamCreateWithFlags(&mut s, flags.bits()),
                }
            })?;
            Ok(Stream { inner: s })
        }
        pub fn synchronize(&self) -> CudaResult<()> { check(unsafe { hipStreamSynchronize(self.inner) }) }
        pub(crate) fn as_inner(&self) -> hipStream_t { self.inner }
    }
    impl Drop for Stream {
        fn drop(&mut self) {
            if !self.inner.is_null() { unsafe { hipStreamDestroy(self.inner) }; }
        }
    }
}

// ---------------------------------------------------------------------------------------------- memory
pub mod memory {
    use super::*;
    use std::mem::size_of;
    use std::ops::{Deref, DerefMut, Index, IndexMut, Range, RangeFrom, RangeFull, RangeInclusive, RangeTo, RangeToInclusive};
    pub use rustacuda_core::{DeviceCopy, DevicePointer};
    use stream::Stream;

    /// `cuda_malloc_locked`: page-locked host memory (the reference's `MyAllocator`, fri/oracle.rs:50-72)
    pub unsafe fn cuda_malloc_locked<T>(count: usize) -> CudaResult<*mut T> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(hipHostMalloc(&mut p, count.checked_mul(size_of::<T>()).ok_or(error::CudaError(1))?, 0))?;
        Ok(p as *mut T)
    }
    pub unsafe fn cuda_free_locked<T>(ptr: *mut T) -> CudaResult<()> { check(hipHostFree(ptr as *mut c_void)) }
    pub unsafe fn cuda_malloc<T>(count: usize) -> CudaResult<DevicePointer<T>> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(hipMalloc(&mut p, count.checked_mul(size_of::<T>()).ok_or(error::CudaError(1))?))?;
        Ok(DevicePointer::wrap(p as *mut T))
    }
    pub unsafe fn cuda_free<T>(mut ptr: DevicePointer<T>) -> CudaResult<()> { check(hipFree(ptr.as_raw_mut() as *mut c_void)) }

    /// A range of device memory seen as an unsized slice (rustacuda's trick: the fat pointer holds a DEVICE address and a
    /// length; it is never dereferenced on the host, only re-sliced and handed to copies and kernels).
    #[repr(C)]
    pub struct DeviceSlice<T>([T]);
    impl<T> DeviceSlice<T> {
        pub fn len(&self) -> usize { self.0.len() }
        pub fn is_empty(&self) -> bool { self.0.is_empty() }
        pub fn as_ptr(&self) -> *const T { self.0.as_ptr() }
        pub fn as_mut_ptr(&mut self) -> *mut T { self.0.as_mut_ptr() }
        pub fn as_device_ptr(&mut self) -> DevicePointer<T> { unsafe { DevicePointer::wrap(self.0.as_mut_ptr()) } }
        pub fn split_at(&self, mid: usize) -> (&DeviceSlice<T>, &DeviceSlice<T>) {
            let (l, r) = self.0.split_at(mid);
            unsafe { (DeviceSlice::from_slice(l), DeviceSlice::from_slice(r)) }
        }
        pub fn split_at_mut(&mut self, mid: usize) -> (&mut DeviceSlice<T>, &mut DeviceSlice<T>) {
            let (l, r) = self.0.split_at_mut(mid);
            unsafe { (DeviceSlice::from_slice_mut(l), DeviceSlice::from_slice_mut(r)) }
        }
        pub unsafe fn from_slice(slice: &[T]) -> &DeviceSlice<T> { &*(slice as *const [T] as *const DeviceSlice<T>) }
        pub unsafe fn from_slice_mut(slice: &mut [T]) -> &mut DeviceSlice<T> { &mut *(slice as *mut [T] as *mut DeviceSlice<T>) }
        pub unsafe fn from_raw_parts<'a>(data: DevicePointer<T>, len: usize) -> &'a DeviceSlice<T> {
            DeviceSlice::from_slice(std::slice::from_raw_parts(data.as_raw(), len))
        }
        pub unsafe fn from_raw_parts_mut<'a>(mut data: DevicePointer<T>, len: usize) -> &'a mut DeviceSlice<T> {
            DeviceSlice::from_slice_mut(std::slice::from_raw_parts_mut(data.as_raw_mut(), len))
        }
    }
    macro_rules! impl_index {
        ($($t:ty)*) => { $(
            impl<T> Index<$t> for DeviceSlice<T> {
                type Output = DeviceSlice<T>;
                fn index(&self, index: $t) -> &Self { unsafe { DeviceSlice::from_slice(self.0.index(index)) } }
            }
            impl<T> IndexMut<$t> for DeviceSlice<T> {
                fn index_mut(&mut self, index: $t) -> &mut Self { unsafe { DeviceSlice::from_slice_mut(self.0.index_mut(index)) } }
            }
        )* };
    }
    impl_index! { Range<usize> RangeFull RangeFrom<usize> RangeInclusive<usize> RangeTo<usize> RangeToInclusive<usize> }

    /// `copy_from` / `copy_to` (synchronous)
    pub trait CopyDestination<O: ?Sized>: crate::private::Sealed {
        fn copy_from(&mut self, source: &O) -> CudaResult<()>;
        fn copy_to(&self, dest: &mut O) -> CudaResult<()>;
    }
    /// `async_copy_from` / `async_copy_to`: the host side must stay alive (and, to be truly asynchronous, be page-locked)
    /// until the stream has been synchronised — rustacuda's contract, which is why both are `unsafe`.
    pub trait AsyncCopyDestination<O: ?Sized>: crate::private::Sealed {
        unsafe fn async_copy_from(&mut self, source: &O, stream: &Stream) -> CudaResult<()>;
        unsafe fn async_copy_to(&self, dest: &mut O, stream: &Stream) -> CudaResult<()>;
    }
    impl<T> crate::private::Sealed for DeviceSlice<T> {}
    impl<T: DeviceCopy, I: AsRef<[T]> + AsMut<[T]> + ?Sized> CopyDestination<I> for DeviceSlice<T> {
        fn copy_from(&mut self, source: &I) -> CudaResult<()> {
            let s = source.as_ref();
            assert!(self.len() == s.len(), "destination and source slices have different lengths");
            if s.is_empty() { return Ok(()); }
            check(unsafe { hipMemcpy(self.0.as_mut_ptr() as *mut c_void, s.as_ptr() as *const c_void, size_of::<T>() * s.len(), HIP_MEMCPY_H2D) })
        }
        fn copy_to(&self, dest: &mut I) -> CudaResult<()> {
            let d = dest.as_mut();
            assert!(self.len() == d.len(), "destination and source slices have different lengths");
            if d.is_empty() { return Ok(()); }
            check(unsafe { hipMemcpy(d.as_mut_ptr() as *mut c_void, self.0.as_ptr() as *const c_void, size_of::<T>() * d.len(), HIP_MEMCPY_D2H) })
        }
    }
    impl<T: DeviceCopy, I: AsRef<[T]> + AsMut<[T]> + ?Sized> AsyncCopyDestination<I> for DeviceSlice<T> {
        unsafe fn async_copy_from(&mut self, source: &I, stream: &Stream) -> CudaResult<()> {
            let s = source.as_ref();
            assert!(self.len() == s.len(), "destination and source slices have different lengths");
            if s.is_empty() { return Ok(()); }
            check(hipMemcpyAsync(self.0.as_mut_ptr() as *mut c_void, s.as_ptr() as *const c_void, size_of::<T>() * s.len(), HIP_MEMCPY_H2D, stream.as_inner()))
        }
        unsafe fn async_copy_to(&self, dest: &mut I, stream: &Stream) -> CudaResult<()> {
            let d = dest.as_mut();
            assert!(self.len() == d.len(), "destination and source slices have different lengths");
            if d.is_empty() { return Ok(()); }
            check(hipMemcpyAsync(d.as_mut_ptr() as *mut c_void, self.0.as_ptr() as *const c_void, size_of::<T>() * d.len(), HIP_MEMCPY_D2H, stream.as_inner()))
        }
    }

    /// `rustacuda::memory::DeviceBuffer`: an owned allocation that derefs to a [`DeviceSlice`]
    pub struct DeviceBuffer<T> { buf: DevicePointer<T>, capacity: usize }
    unsafe impl<T: Send> Send for DeviceBuffer<T> {}
    unsafe impl<T: Sync> Sync for DeviceBuffer<T> {}
    impl<T> DeviceBuffer<T> {
        pub unsafe fn uninitialized(size: usize) -> CudaResult<Self> {
            let ptr = if size > 0 && size_of::<T>() > 0 { cuda_malloc(size)? } else { DevicePointer::wrap(std::ptr::NonNull::dangling().as_ptr()) };
            Ok(DeviceBuffer { buf: ptr, capacity: size })
        }
        pub fn as_slice(&self) -> &DeviceSlice<T> { self }
        pub fn as_mut_slice(&mut self) -> &mut DeviceSlice<T> { self }
        pub fn drop(mut dev_buf: DeviceBuffer<T>) -> Result<(), (error::CudaError, DeviceBuffer<T>)> {
            if dev_buf.capacity > 0 && size_of::<T>() > 0 {
                let ptr = std::mem::replace(&mut dev_buf.buf, DevicePointer::null());
                let cap = std::mem::replace(&mut dev_buf.capacity, 0);
                match unsafe { cuda_free(ptr) } {
                    Ok(()) => { std::mem::forget(dev_buf); Ok(()) }
                    Err(e) => Err((e, DeviceBuffer { buf: ptr, capacity: cap })),
                }
            } else { Ok(()) }
        }
    }
    impl<T: DeviceCopy> DeviceBuffer<T> {
        pub fn from_slice(slice: &[T]) -> CudaResult<Self> {
            unsafe {
                let mut b = DeviceBuffer::uninitialized(slice.len())?;
                b.copy_from(slice)?;
                Ok(b)
            }
        }
        pub unsafe fn from_slice_async(slice: &[T], stream: &Stream) -> CudaResult<Self> {
            let mut b = DeviceBuffer::uninitialized(slice.len())?;
            b.async_copy_from(slice, stream)?;
            Ok(b)
        }
    }
    impl<T> Deref for DeviceBuffer<T> {
        type Target = DeviceSlice<T>;
        fn deref(&self) -> &DeviceSlice<T> { unsafe { DeviceSlice::from_slice(std::slice::from_raw_parts(self.buf.as_raw(), self.capacity)) } }
    }
    impl<T> DerefMut for DeviceBuffer<T> {
        fn deref_mut(&mut self) -> &mut DeviceSlice<T> { unsafe { DeviceSlice::from_slice_mut(std::slice::from_raw_parts_mut(self.buf.as_raw_mut(), self.capacity)) } }
    }
    impl<T> Drop for DeviceBuffer<T> {
        fn drop(&mut self) {
            if self.capacity > 0 && size_of::<T>() > 0 && !self.buf.is_null() {
                let ptr = std::mem::replace(&mut self.buf, DevicePointer::null());
                unsafe { let _ = cuda_free(ptr); }
            }
            self.capacity = 0;
        }
    }
}

mod private {
    pub trait Sealed {}
}

/// `rustacuda::prelude`
pub mod prelude {
    pub use crate::context::{Context, ContextFlags};
    pub use crate::device::Device;
    pub use crate::memory::{CopyDestination, DeviceBuffer};
    pub use crate::stream::{Stream, StreamFlags};
    pub use crate::CudaFlags;
}

//! `decaf377::gpu` — batch entry points backed by libdecaf377_amd.so (MI355X).
//!
//! UNBUILT SOURCE (no Rust toolchain in the build image). Mirrors include/decaf377_amd.h.
//! Per-element semantics are those of the existing methods:
//!   Encoding::vartime_decompress   src/ark_curve/encoding.rs:32-83
//!   Element::vartime_compress      src/ark_curve/encoding.rs:116-128
//!   Element::encode_to_curve       src/ark_curve/elligator.rs:74-76
//!   Element::hash_to_curve         src/ark_curve/elligator.rs:67-71
//!   Element * Fr                   src/ark_curve/ops/projective.rs:106-131
//!   Element::vartime_multiscalar_mul  src/ark_curve/element/projective.rs:99-117
//!   Fq::sqrt_ratio_zeta            src/ark_curve/invsqrt.rs:75-166
#![cfg(feature = "gpu")]

use core::ffi::{c_char, c_int};
use std::ffi::CStr;

use crate::{Element, Encoding, EncodingError, Fq, Fr};

#[repr(C)]
pub struct D377Ctx {
    _private: [u8; 0],
}

#[allow(non_snake_case)]
extern "C" {
    fn d377_ctx_create(device_ids: *const c_int, n_dev: c_int, out: *mut *mut D377Ctx) -> c_int;
    fn d377_ctx_destroy(ctx: *mut D377Ctx);
    fn d377_last_error() -> *const c_char;
    fn d377_batch_sqrt_ratio_zeta(ctx: *mut D377Ctx, num32: *const u8, den32: *const u8, n: usize,
                                  root32: *mut u8, was_square: *mut u8) -> c_int;
    fn d377_batch_decompress(ctx: *mut D377Ctx, enc32: *const u8, n: usize, xyzt: *mut u64, status: *mut u8) -> c_int;
    fn d377_batch_compress(ctx: *mut D377Ctx, xyzt: *const u64, n: usize, enc32: *mut u8) -> c_int;
    fn d377_batch_scalar_mul_base(ctx: *mut D377Ctx, scalar32: *const u8, n: usize, out32: *mut u8) -> c_int;
    fn d377_batch_scalar_mul_var(ctx: *mut D377Ctx, enc32: *const u8, scalar32: *const u8, n: usize,
                                 out32: *mut u8, status: *mut u8) -> c_int;
    fn d377_batch_encode_to_curve(ctx: *mut D377Ctx, fq32: *const u8, n: usize, out32: *mut u8) -> c_int;
    fn d377_batch_hash_to_curve(ctx: *mut D377Ctx, r1: *const u8, r2: *const u8, n: usize, out32: *mut u8) -> c_int;
    fn d377_msm(ctx: *mut D377Ctx, xyzt: *const u64, scalar32: *const u8, n: usize, enc32_out: *mut u8,
                xyzt_out: *mut u64) -> c_int;
}

/// Owns one `d377_ctx` (device tables + scratch). `Send`, not `Sync`: one call in flight.
pub struct GpuContext(*mut D377Ctx);
unsafe impl Send for GpuContext {}

#[derive(Debug)]
pub struct GpuError(pub i32, pub String);

fn check(rc: c_int) -> Result<(), GpuError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(d377_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError(rc, msg))
}

impl GpuContext {
    pub fn new(device_ids: &[i32]) -> Result<Self, GpuError> {
        let mut p = core::ptr::null_mut();
        check(unsafe { d377_ctx_create(device_ids.as_ptr(), device_ids.len() as c_int, &mut p) })?;
        Ok(Self(p))
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { d377_ctx_destroy(self.0) }
    }
}

// ---- record conversions ------------------------------------------------------------------------
// Encoding(pub [u8; 32]) is repr-transparent over the bytes, so &[Encoding] is already packed.
fn enc_ptr(e: &[Encoding]) -> *const u8 {
    e.as_ptr() as *const u8
}
fn pack32<T>(xs: &[T], f: impl Fn(&T) -> [u8; 32]) -> Vec<u8> {
    let mut v = Vec::with_capacity(32 * xs.len());
    for x in xs {
        v.extend_from_slice(&f(x));
    }
    v
}
/// X, Y, Z, T Montgomery limbs (the C ABI order); ark's Projective stores (x, y, t, z).
fn element_to_xyzt(e: &Element) -> [u64; 16] {
    let p = &e.inner;
    let mut o = [0u64; 16];
    o[0..4].copy_from_slice(&p.x.0 .0 .0);
    o[4..8].copy_from_slice(&p.y.0 .0 .0);
    o[8..12].copy_from_slice(&p.z.0 .0 .0);
    o[12..16].copy_from_slice(&p.t.0 .0 .0);
    o
}
fn element_from_xyzt(o: &[u64]) -> Element {
    let f = |l: &[u64]| Fq::from_montgomery_limbs([l[0], l[1], l[2], l[3]]);
    Element {
        inner: crate::ark_curve::edwards::EdwardsProjective::new_unchecked(
            f(&o[0..4]).0, f(&o[4..8]).0, f(&o[12..16]).0, f(&o[8..12]).0, // (x, y, t, z)
        ),
    }
}

// ---- batch API ---------------------------------------------------------------------------------
impl Encoding {
    /// One `Result` per input, same order as `encs`.
    pub fn vartime_decompress_batch(ctx: &mut GpuContext, encs: &[Encoding])
        -> Result<Vec<Result<Element, EncodingError>>, GpuError> {
        let n = encs.len();
        let mut xyzt = vec![0u64; 16 * n];
        let mut st = vec![0u8; n];
        check(unsafe { d377_batch_decompress(ctx.0, enc_ptr(encs), n, xyzt.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok((0..n)
            .map(|i| if st[i] == 0 { Ok(element_from_xyzt(&xyzt[16 * i..16 * i + 16])) } else { Err(EncodingError::InvalidEncoding) })
            .collect())
    }
}

impl Element {
    pub fn vartime_compress_batch(ctx: &mut GpuContext, els: &[Element]) -> Result<Vec<Encoding>, GpuError> {
        let n = els.len();
        let xyzt: Vec<u64> = els.iter().flat_map(|e| element_to_xyzt(e)).collect();
        let mut out = vec![Encoding([0u8; 32]); n];
        check(unsafe { d377_batch_compress(ctx.0, xyzt.as_ptr(), n, out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }
    pub fn encode_to_curve_batch(ctx: &mut GpuContext, rs: &[Fq]) -> Result<Vec<Encoding>, GpuError> {
        let bytes = pack32(rs, |r| r.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); rs.len()];
        check(unsafe { d377_batch_encode_to_curve(ctx.0, bytes.as_ptr(), rs.len(), out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }
    pub fn hash_to_curve_batch(ctx: &mut GpuContext, r1: &[Fq], r2: &[Fq]) -> Result<Vec<Encoding>, GpuError> {
        assert_eq!(r1.len(), r2.len());
        let (a, b) = (pack32(r1, |r| r.to_bytes()), pack32(r2, |r| r.to_bytes()));
        let mut out = vec![Encoding([0u8; 32]); r1.len()];
        check(unsafe { d377_batch_hash_to_curve(ctx.0, a.as_ptr(), b.as_ptr(), r1.len(), out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }
    /// `Element::GENERATOR * k` for every k.
    pub fn mul_generator_batch(ctx: &mut GpuContext, ks: &[Fr]) -> Result<Vec<Encoding>, GpuError> {
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); ks.len()];
        check(unsafe { d377_batch_scalar_mul_base(ctx.0, bytes.as_ptr(), ks.len(), out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }
    /// `decompress(P_i)? * k_i`, compressed.
    pub fn scalar_mul_batch(ctx: &mut GpuContext, ps: &[Encoding], ks: &[Fr])
        -> Result<Vec<Result<Encoding, EncodingError>>, GpuError> {
        assert_eq!(ps.len(), ks.len());
        let n = ps.len();
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); n];
        let mut st = vec![0u8; n];
        check(unsafe { d377_batch_scalar_mul_var(ctx.0, enc_ptr(ps), bytes.as_ptr(), n, out.as_mut_ptr() as *mut u8, st.as_mut_ptr()) })?;
        Ok((0..n).map(|i| if st[i] == 0 { Ok(out[i]) } else { Err(EncodingError::InvalidEncoding) }).collect())
    }
    /// GPU form of `vartime_multiscalar_mul` (Pippenger).
    pub fn vartime_multiscalar_mul_gpu(ctx: &mut GpuContext, scalars: &[Fr], points: &[Element]) -> Result<Element, GpuError> {
        assert_eq!(scalars.len(), points.len());
        let xyzt: Vec<u64> = points.iter().flat_map(|e| element_to_xyzt(e)).collect();
        let bytes = pack32(scalars, |k| k.to_bytes());
        let mut enc = [0u8; 32];
        let mut out = [0u64; 16];
        check(unsafe { d377_msm(ctx.0, xyzt.as_ptr(), bytes.as_ptr(), points.len(), enc.as_mut_ptr(), out.as_mut_ptr()) })?;
        Ok(element_from_xyzt(&out))
    }
}

impl Fq {
    pub fn sqrt_ratio_zeta_batch(ctx: &mut GpuContext, num: &[Fq], den: &[Fq]) -> Result<Vec<(bool, Fq)>, GpuError> {
        assert_eq!(num.len(), den.len());
        let n = num.len();
        let (a, b) = (pack32(num, |x| x.to_bytes()), pack32(den, |x| x.to_bytes()));
        let mut root = vec![0u8; 32 * n];
        let mut ws = vec![0u8; n];
        check(unsafe { d377_batch_sqrt_ratio_zeta(ctx.0, a.as_ptr(), b.as_ptr(), n, root.as_mut_ptr(), ws.as_mut_ptr()) })?;
        Ok((0..n)
            .map(|i| {
                let mut r = [0u8; 32];
                r.copy_from_slice(&root[32 * i..32 * i + 32]);
                (ws[i] != 0, Fq::from_bytes_checked(&r).expect("library returns canonical bytes"))
            })
            .collect())
    }
}

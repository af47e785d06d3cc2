//! `decaf377::gpu` — batch entry points backed by libdecaf377_amd.so (MI355X).
//!
//! UNBUILT SOURCE (no Rust toolchain in the build image).  Intended placement: `src/gpu.rs` inside the
//! crate behind a `gpu` feature, with `src/gpu/ffi.rs` = rust/src/ffi.rs (generated from
//! include/decaf377_amd.h).  tests/test_rust_shim.py checks every `ffi::d377_*` call below against
//! the header (name and argument count) and the extern block against the header (every type).
//!
//! Per-element semantics are those of the existing methods:
//!   Encoding::vartime_decompress      src/ark_curve/encoding.rs:32-83
//!   Element::vartime_compress         src/ark_curve/encoding.rs:116-128
//!   Element::encode_to_curve          src/ark_curve/elligator.rs:74-76
//!   Element::hash_to_curve            src/ark_curve/elligator.rs:67-71
//!   Element + / double / neg / ==     src/ark_curve/ops/projective.rs:5-104, element/projective.rs:65-86
//!   Element * Fr                      src/ark_curve/ops/projective.rs:106-131
//!   Element::vartime_multiscalar_mul  src/ark_curve/element/projective.rs:99-117
//!   CurveGroup::normalize_batch       src/ark_curve/element.rs:74-81
//!   Fq::sqrt_ratio_zeta               src/ark_curve/invsqrt.rs:75-166 (min_curve: src/min_curve/invsqrt.rs:73-95)
//!   Fq add/sub/mul/square/neg/inverse src/fields/fq/u64/wrapper.rs:99-132
//!   Fq::from_le_bytes_mod_order       src/fields/fq.rs:90-102
//!
//! One three-line addition to the crate is needed: `Fq` keeps its arkworks value private to
//! `src/fields/fq/u64/wrapper.rs`, and the C ABI exchanges Elements as Montgomery limbs, so that file gets
//!
//!     pub(crate) fn to_montgomery_limbs(&self) -> [u64; N] { self.0 .0 .0 }
//!
//! next to the existing `pub const fn from_montgomery_limbs` (wrapper.rs:82-85).
#![cfg(feature = "gpu")]

use core::ffi::c_void;
use std::ffi::CStr;

use crate::ark_curve::EdwardsProjective;
use crate::{Element, Encoding, EncodingError, Fq, Fr};

#[path = "gpu/ffi.rs"]
pub mod ffi;

/// Which backend's root `sqrt_ratio_zeta_batch` returns (the flags are the same).
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub enum SqrtRoot {
    /// `Fq::sqrt_ratio_zeta` of the `arkworks` backend (Sarkar tables).
    Ark = ffi::D377_SQRT_ROOT_ARK as isize,
    /// `Fq::non_arkworks_sqrt_ratio_zeta` of the `min_curve` backend (Tonelli-Shanks, seed 11^m).
    MinCurve = ffi::D377_SQRT_ROOT_MIN_CURVE as isize,
}

/// Owns one `d377_ctx` (device tables, streams, scratch).  Calls on one context are serialised inside the
/// library, so `&GpuContext` may be shared between threads.
pub struct GpuContext(*mut ffi::D377Ctx);
unsafe impl Send for GpuContext {}
unsafe impl Sync for GpuContext {}

#[derive(Debug)]
pub struct GpuError(pub i32, pub String);
impl GpuError {
    /// `D377_ERR_STARVED`: workgroups of the call found no free lane set for 10 s and wrote no output -- the call
    /// produced nothing usable (`GpuContext::health`, `GpuContext::reset_scratch`, then call again).
    pub fn is_starved(&self) -> bool {
        self.0 == ffi::D377_ERR_STARVED
    }
}

fn check(rc: i32) -> Result<(), GpuError> {
    if rc == ffi::D377_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::d377_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError(rc, msg))
}

impl GpuContext {
    /// `device_ids` empty = device 0.  Several ids: host batches are split into contiguous slices over them.
    pub fn new(device_ids: &[i32]) -> Result<Self, GpuError> {
        let mut p = core::ptr::null_mut();
        check(unsafe { ffi::d377_ctx_create(device_ids.as_ptr(), device_ids.len() as i32, &mut p) })?;
        Ok(Self(p))
    }
    /// The same with the fixed-base comb's options (d377_ctx_create_ex): `comb_bits` 0 (the library's default, 23), 18, 21
    /// or 23 -- 0.24 / 1.6 / 5.9 GB per device; `comb_lazy`: build the table on the first `GENERATOR * Fr` batch instead of
    /// now, so that a context that never multiplies by the generator does not pay for it (in the crate
    /// `Element::GENERATOR` is a constant, src/min_curve/element.rs:61-81).
    pub fn with_comb(device_ids: &[i32], comb_bits: i32, comb_lazy: bool) -> Result<Self, GpuError> {
        let mut p = core::ptr::null_mut();
        let opts = ffi::D377CtxOpts { size: core::mem::size_of::<ffi::D377CtxOpts>(), comb_bits, comb_lazy: comb_lazy as i32 };
        check(unsafe { ffi::d377_ctx_create_ex(device_ids.as_ptr(), device_ids.len() as i32, &opts, &mut p) })?;
        Ok(Self(p))
    }
    /// (comb width in bits, built yet?, table bytes) of device `dev`.
    pub fn comb_info(&self, dev: i32) -> Result<(i32, bool, u64), GpuError> {
        let (mut bits, mut built, mut bytes) = (0i32, 0i32, 0u64);
        check(unsafe { ffi::d377_ctx_comb_info(self.0, dev, &mut bits, &mut built, &mut bytes) })?;
        Ok((bits, built != 0, bytes))
    }
    pub fn num_devices(&self) -> usize {
        unsafe { ffi::d377_ctx_num_devices(self.0) as usize }
    }
    /// (lane sets claimed now, workgroups that waited > 0.25 s for one, workgroups that gave up after 10 s) of device `dev`:
    /// (0, 0, 0) on a healthy idle context.  Does not wait for running kernels.
    pub fn health(&self, dev: i32) -> Result<(i32, u64, u64), GpuError> {
        let (mut claimed, mut waited, mut gave_up) = (0i32, 0u64, 0u64);
        check(unsafe { ffi::d377_ctx_health(self.0, dev, &mut claimed, &mut waited, &mut gave_up) })?;
        Ok((claimed, waited, gave_up))
    }
    /// Frees lane sets leaked by a launch that died; returns how many it freed (0 on a healthy context).
    pub fn reset_scratch(&self, dev: i32) -> Result<i32, GpuError> {
        let mut freed = 0i32;
        check(unsafe { ffi::d377_ctx_reset_scratch(self.0, dev, &mut freed) })?;
        Ok(freed)
    }
    /// Device address of the 32-bit gave-up counter of `dev` (for callers of the `_dev` entry points: copy it on the
    /// call's stream before and after, compare once the stream is synchronised).
    pub fn starved_counter_dev(&self, dev: i32) -> Result<*const u32, GpuError> {
        let mut p: *const u32 = core::ptr::null();
        check(unsafe { ffi::d377_ctx_starved_counter_dev(self.0, dev, &mut p) })?;
        Ok(p)
    }
    /// Developer interface: override one launch rule (`ffi::D377_TUNE_*`); `None` restores the built-in rule.
    pub fn set_tuning(&self, key: i32, value: Option<i64>) -> Result<(), GpuError> {
        check(unsafe { ffi::d377_ctx_set_tuning(self.0, key, value.unwrap_or(ffi::D377_TUNE_DEFAULT as i64)) })
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { ffi::d377_ctx_destroy(self.0) }
    }
}

// ---- record conversions ------------------------------------------------------------------------
// Encoding(pub [u8; 32]) is a newtype over the bytes, so &[Encoding] is already a packed [n][32] array.
fn enc_ptr(e: &[Encoding]) -> *const u8 {
    e.as_ptr() as *const u8
}
fn enc_mut_ptr(e: &mut [Encoding]) -> *mut u8 {
    e.as_mut_ptr() as *mut u8
}
fn pack32<T>(xs: &[T], f: impl Fn(&T) -> [u8; 32]) -> Vec<u8> {
    let mut v = Vec::with_capacity(32 * xs.len());
    for x in xs {
        v.extend_from_slice(&f(x));
    }
    v
}
fn fq_from_canonical(bytes: &[u8]) -> Fq {
    let mut r = [0u8; 32];
    r.copy_from_slice(bytes);
    Fq::from_bytes_checked(&r).expect("the library returns canonical bytes")
}
fn results<T>(status: &[u8], mut ok: impl FnMut(usize) -> T) -> Vec<Result<T, EncodingError>> {
    (0..status.len())
        .map(|i| if status[i] == 0 { Ok(ok(i)) } else { Err(EncodingError::InvalidEncoding) })
        .collect()
}
/// X, Y, Z, T Montgomery limbs (the C ABI order); ark's Projective stores (x, y, t, z).
fn element_to_xyzt(e: &Element, o: &mut [u64]) {
    let p = &e.inner;
    o[0..4].copy_from_slice(&p.x.to_montgomery_limbs());
    o[4..8].copy_from_slice(&p.y.to_montgomery_limbs());
    o[8..12].copy_from_slice(&p.z.to_montgomery_limbs());
    o[12..16].copy_from_slice(&p.t.to_montgomery_limbs());
}
fn elements_to_xyzt(es: &[Element]) -> Vec<u64> {
    let mut v = vec![0u64; 16 * es.len()];
    for (e, o) in es.iter().zip(v.chunks_exact_mut(16)) {
        element_to_xyzt(e, o);
    }
    v
}
fn fq_limbs(l: &[u64]) -> Fq {
    Fq::from_montgomery_limbs([l[0], l[1], l[2], l[3]])
}
fn element_from_xyzt(o: &[u64]) -> Element {
    // `Decaf377EdwardsConfig::BaseField` is the crate's own `Fq` (src/ark_curve/edwards.rs:21), and
    // `Projective::new_unchecked` takes (x, y, t, z)
    Element {
        inner: EdwardsProjective::new_unchecked(fq_limbs(&o[0..4]), fq_limbs(&o[4..8]), fq_limbs(&o[12..16]), fq_limbs(&o[8..12])),
    }
}
fn elements_from_xyzt(v: &[u64]) -> Vec<Element> {
    v.chunks_exact(16).map(element_from_xyzt).collect()
}
fn fqs_to_limbs(xs: &[Fq]) -> Vec<u64> {
    xs.iter().flat_map(|x| x.to_montgomery_limbs()).collect()
}

// ---- Encoding ----------------------------------------------------------------------------------
impl Encoding {
    /// Batch form of `vartime_decompress`: one `Result` per input, same order.
    pub fn vartime_decompress_batch(ctx: &GpuContext, encs: &[Encoding]) -> Result<Vec<Result<Element, EncodingError>>, GpuError> {
        let n = encs.len();
        let mut xyzt = vec![0u64; 16 * n];
        let mut st = vec![0u8; n];
        check(unsafe { ffi::d377_batch_decompress(ctx.0, enc_ptr(encs), n, xyzt.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok(results(&st, |i| element_from_xyzt(&xyzt[16 * i..16 * i + 16])))
    }
    /// decompress then compress (the round trip of tests/encoding.rs:97-107): `Ok(e)` has `e == input`.
    pub fn roundtrip_batch(ctx: &GpuContext, encs: &[Encoding]) -> Result<Vec<Result<Encoding, EncodingError>>, GpuError> {
        let n = encs.len();
        let mut out = vec![Encoding([0u8; 32]); n];
        let mut st = vec![0u8; n];
        check(unsafe { ffi::d377_batch_roundtrip(ctx.0, enc_ptr(encs), n, enc_mut_ptr(&mut out), st.as_mut_ptr()) })?;
        Ok(results(&st, |i| out[i]))
    }
}

// ---- Element -----------------------------------------------------------------------------------
impl Element {
    pub fn vartime_compress_batch(ctx: &GpuContext, els: &[Element]) -> Result<Vec<Encoding>, GpuError> {
        let xyzt = elements_to_xyzt(els);
        let mut out = vec![Encoding([0u8; 32]); els.len()];
        check(unsafe { ffi::d377_batch_compress(ctx.0, xyzt.as_ptr(), els.len(), enc_mut_ptr(&mut out)) })?;
        Ok(out)
    }
    pub fn encode_to_curve_batch(ctx: &GpuContext, rs: &[Fq]) -> Result<Vec<Encoding>, GpuError> {
        let bytes = pack32(rs, |r| r.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); rs.len()];
        check(unsafe { ffi::d377_batch_encode_to_curve(ctx.0, bytes.as_ptr(), rs.len(), enc_mut_ptr(&mut out)) })?;
        Ok(out)
    }
    /// `encode_to_curve(Fq::from_le_bytes_mod_order(h))` for 48- or 64-byte hash outputs, fused on the device.
    pub fn encode_to_curve_wide_batch(ctx: &GpuContext, hashes: &[u8], len: usize) -> Result<Vec<Encoding>, GpuError> {
        assert!(len == 48 || len == 64);
        assert_eq!(hashes.len() % len, 0);
        let n = hashes.len() / len;
        let mut out = vec![Encoding([0u8; 32]); n];
        check(unsafe { ffi::d377_batch_encode_to_curve_wide(ctx.0, hashes.as_ptr(), len, n, enc_mut_ptr(&mut out)) })?;
        Ok(out)
    }
    pub fn hash_to_curve_batch(ctx: &GpuContext, r1: &[Fq], r2: &[Fq]) -> Result<Vec<Encoding>, GpuError> {
        assert_eq!(r1.len(), r2.len());
        let (a, b) = (pack32(r1, |r| r.to_bytes()), pack32(r2, |r| r.to_bytes()));
        let mut out = vec![Encoding([0u8; 32]); r1.len()];
        check(unsafe { ffi::d377_batch_hash_to_curve(ctx.0, a.as_ptr(), b.as_ptr(), r1.len(), enc_mut_ptr(&mut out)) })?;
        Ok(out)
    }
    /// `Element::GENERATOR * k` for every k.
    pub fn mul_generator_batch(ctx: &GpuContext, ks: &[Fr]) -> Result<Vec<Encoding>, GpuError> {
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); ks.len()];
        check(unsafe { ffi::d377_batch_scalar_mul_base(ctx.0, bytes.as_ptr(), ks.len(), enc_mut_ptr(&mut out)) })?;
        Ok(out)
    }
    /// `decompress(P_i)? * k_i`, compressed.
    pub fn scalar_mul_batch(ctx: &GpuContext, ps: &[Encoding], ks: &[Fr]) -> Result<Vec<Result<Encoding, EncodingError>>, GpuError> {
        assert_eq!(ps.len(), ks.len());
        let n = ps.len();
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![Encoding([0u8; 32]); n];
        let mut st = vec![0u8; n];
        check(unsafe {
            ffi::d377_batch_scalar_mul_var(ctx.0, enc_ptr(ps), bytes.as_ptr(), n, enc_mut_ptr(&mut out), st.as_mut_ptr())
        })?;
        Ok(results(&st, |i| out[i]))
    }
    /// `Element * Fr` element-wise, Elements in and out (`impl Mul<Fr> for Element`, src/min_curve/ops.rs:89-95).
    /// The result is the reference's group element; its projective coordinates are the library's own.
    pub fn mul_batch(ctx: &GpuContext, ps: &[Element], ks: &[Fr]) -> Result<Vec<Element>, GpuError> {
        assert_eq!(ps.len(), ks.len());
        let n = ps.len();
        let xyzt = elements_to_xyzt(ps);
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![0u64; 16 * n];
        check(unsafe { ffi::d377_batch_scalar_mul_var_element(ctx.0, xyzt.as_ptr(), bytes.as_ptr(), n, out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    /// `Element::GENERATOR * k` as Elements.
    pub fn mul_generator_element_batch(ctx: &GpuContext, ks: &[Fr]) -> Result<Vec<Element>, GpuError> {
        let bytes = pack32(ks, |k| k.to_bytes());
        let mut out = vec![0u64; 16 * ks.len()];
        check(unsafe { ffi::d377_batch_scalar_mul_base_element(ctx.0, bytes.as_ptr(), ks.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    /// `Element::vartime_compress_to_field` (src/min_curve/element.rs:163-181).
    pub fn vartime_compress_to_field_batch(ctx: &GpuContext, els: &[Element]) -> Result<Vec<Fq>, GpuError> {
        let xyzt = elements_to_xyzt(els);
        let mut out = vec![0u64; 4 * els.len()];
        check(unsafe { ffi::d377_batch_compress_to_field(ctx.0, xyzt.as_ptr(), els.len(), out.as_mut_ptr()) })?;
        Ok(out.chunks_exact(4).map(fq_limbs).collect())
    }
    /// `Element::encode_to_curve` as Elements (the coordinates the reference's formulas give).
    pub fn encode_to_curve_element_batch(ctx: &GpuContext, rs: &[Fq]) -> Result<Vec<Element>, GpuError> {
        let bytes = pack32(rs, |r| r.to_bytes());
        let mut out = vec![0u64; 16 * rs.len()];
        check(unsafe { ffi::d377_batch_encode_to_curve_element(ctx.0, bytes.as_ptr(), rs.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    /// `Element::hash_to_curve` as Elements.
    pub fn hash_to_curve_element_batch(ctx: &GpuContext, r1: &[Fq], r2: &[Fq]) -> Result<Vec<Element>, GpuError> {
        assert_eq!(r1.len(), r2.len());
        let (a, b) = (pack32(r1, |r| r.to_bytes()), pack32(r2, |r| r.to_bytes()));
        let mut out = vec![0u64; 16 * r1.len()];
        check(unsafe { ffi::d377_batch_hash_to_curve_element(ctx.0, a.as_ptr(), b.as_ptr(), r1.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    pub fn add_batch(ctx: &GpuContext, ps: &[Element], qs: &[Element]) -> Result<Vec<Element>, GpuError> {
        assert_eq!(ps.len(), qs.len());
        let (a, b) = (elements_to_xyzt(ps), elements_to_xyzt(qs));
        let mut out = vec![0u64; 16 * ps.len()];
        check(unsafe { ffi::d377_batch_add(ctx.0, a.as_ptr(), b.as_ptr(), ps.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    /// `Element - Element` (src/min_curve/ops.rs:43-87)
    pub fn sub_batch(ctx: &GpuContext, ps: &[Element], qs: &[Element]) -> Result<Vec<Element>, GpuError> {
        assert_eq!(ps.len(), qs.len());
        let (a, b) = (elements_to_xyzt(ps), elements_to_xyzt(qs));
        let mut out = vec![0u64; 16 * ps.len()];
        check(unsafe { ffi::d377_batch_sub(ctx.0, a.as_ptr(), b.as_ptr(), ps.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    pub fn double_batch(ctx: &GpuContext, ps: &[Element]) -> Result<Vec<Element>, GpuError> {
        let a = elements_to_xyzt(ps);
        let mut out = vec![0u64; 16 * ps.len()];
        check(unsafe { ffi::d377_batch_double(ctx.0, a.as_ptr(), ps.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    pub fn neg_batch(ctx: &GpuContext, ps: &[Element]) -> Result<Vec<Element>, GpuError> {
        let a = elements_to_xyzt(ps);
        let mut out = vec![0u64; 16 * ps.len()];
        check(unsafe { ffi::d377_batch_neg(ctx.0, a.as_ptr(), ps.len(), out.as_mut_ptr()) })?;
        Ok(elements_from_xyzt(&out))
    }
    /// `PartialEq` for every pair (x1*y2 == x2*y1).
    pub fn eq_batch(ctx: &GpuContext, ps: &[Element], qs: &[Element]) -> Result<Vec<bool>, GpuError> {
        assert_eq!(ps.len(), qs.len());
        let (a, b) = (elements_to_xyzt(ps), elements_to_xyzt(qs));
        let mut eq = vec![0u8; ps.len()];
        check(unsafe { ffi::d377_batch_eq(ctx.0, a.as_ptr(), b.as_ptr(), ps.len(), eq.as_mut_ptr()) })?;
        Ok(eq.into_iter().map(|v| v != 0).collect())
    }
    pub fn is_identity_batch(ctx: &GpuContext, ps: &[Element]) -> Result<Vec<bool>, GpuError> {
        let a = elements_to_xyzt(ps);
        let mut id = vec![0u8; ps.len()];
        check(unsafe { ffi::d377_batch_is_identity(ctx.0, a.as_ptr(), ps.len(), id.as_mut_ptr()) })?;
        Ok(id.into_iter().map(|v| v != 0).collect())
    }
    /// `CurveGroup::normalize_batch`: affine (x, y) of every element (one batched inversion per device lane).
    pub fn normalize_batch_gpu(ctx: &GpuContext, ps: &[Element]) -> Result<Vec<(Fq, Fq)>, GpuError> {
        let a = elements_to_xyzt(ps);
        let mut xy = vec![0u64; 8 * ps.len()];
        check(unsafe { ffi::d377_batch_to_affine(ctx.0, a.as_ptr(), ps.len(), xy.as_mut_ptr()) })?;
        Ok(xy.chunks_exact(8).map(|c| (fq_limbs(&c[0..4]), fq_limbs(&c[4..8]))).collect())
    }
    /// GPU form of `vartime_multiscalar_mul` (Pippenger bucket method).
    pub fn vartime_multiscalar_mul_gpu(ctx: &GpuContext, scalars: &[Fr], points: &[Element]) -> Result<Element, GpuError> {
        assert_eq!(scalars.len(), points.len());
        let xyzt = elements_to_xyzt(points);
        let bytes = pack32(scalars, |k| k.to_bytes());
        let mut enc = [0u8; 32];
        let mut out = [0u64; 16];
        check(unsafe { ffi::d377_msm(ctx.0, xyzt.as_ptr(), bytes.as_ptr(), points.len(), enc.as_mut_ptr(), out.as_mut_ptr()) })?;
        Ok(element_from_xyzt(&out))
    }
    /// The same over Encodings: invalid ones are reported and left out of the sum.
    pub fn vartime_multiscalar_mul_encoded_gpu(ctx: &GpuContext, scalars: &[Fr], points: &[Encoding])
        -> Result<(Element, Vec<Result<(), EncodingError>>), GpuError> {
        assert_eq!(scalars.len(), points.len());
        let bytes = pack32(scalars, |k| k.to_bytes());
        let mut enc = [0u8; 32];
        let mut out = [0u64; 16];
        let mut st = vec![0u8; points.len()];
        check(unsafe {
            ffi::d377_msm_encoded(ctx.0, enc_ptr(points), bytes.as_ptr(), points.len(), enc.as_mut_ptr(), out.as_mut_ptr(), st.as_mut_ptr())
        })?;
        Ok((element_from_xyzt(&out), results(&st, |_| ())))
    }
    /// MANY small sums at once: sum i = scalars[i m ..][..m] . points[i m ..][..m], `m` terms each (1..=8) -- the shape of the
    /// crate's own multiscalar test, a 3-term sum per case (tests/operations.rs:44-60).  One Straus chain per sum on the GPU
    /// (the m points share the doublings).  Returns the sums as Elements, like `vartime_multiscalar_mul`, and their
    /// Encodings, which the same pass computes.
    pub fn vartime_multiscalar_mul_batch_gpu(ctx: &GpuContext, m: usize, scalars: &[Fr], points: &[Element])
        -> Result<(Vec<Element>, Vec<Encoding>), GpuError> {
        assert_eq!(scalars.len(), points.len());
        assert!(m >= 1 && points.len() % m == 0);
        let n = points.len() / m;
        let xyzt = elements_to_xyzt(points);
        let bytes = pack32(scalars, |k| k.to_bytes());
        let mut enc = vec![Encoding([0u8; 32]); n];
        let mut out = vec![0u64; 16 * n];
        check(unsafe {
            ffi::d377_batch_msm_small(ctx.0, xyzt.as_ptr(), bytes.as_ptr(), m, n, enc.as_mut_ptr() as *mut u8, out.as_mut_ptr())
        })?;
        Ok((out.chunks_exact(16).map(element_from_xyzt).collect(), enc))
    }
    /// The same over Encodings: an invalid one is reported (per term) and left out of its sum.
    pub fn vartime_multiscalar_mul_batch_encoded_gpu(ctx: &GpuContext, m: usize, scalars: &[Fr], points: &[Encoding])
        -> Result<(Vec<Element>, Vec<Encoding>, Vec<Result<(), EncodingError>>), GpuError> {
        assert_eq!(scalars.len(), points.len());
        assert!(m >= 1 && points.len() % m == 0);
        let n = points.len() / m;
        let bytes = pack32(scalars, |k| k.to_bytes());
        let mut enc = vec![Encoding([0u8; 32]); n];
        let mut out = vec![0u64; 16 * n];
        let mut st = vec![0u8; points.len()];
        check(unsafe {
            ffi::d377_batch_msm_small_encoded(ctx.0, enc_ptr(points), bytes.as_ptr(), m, n, enc.as_mut_ptr() as *mut u8, out.as_mut_ptr(),
                                              st.as_mut_ptr())
        })?;
        Ok((out.chunks_exact(16).map(element_from_xyzt).collect(), enc, results(&st, |_| ())))
    }
}

// ---- Fq / Fr -----------------------------------------------------------------------------------
/// Which `Fq` operator `fq_op_batch` applies (D377_FQ_*).
#[derive(Copy, Clone, Debug)]
pub enum FqOp {
    Add = ffi::D377_FQ_ADD as isize,
    Sub = ffi::D377_FQ_SUB as isize,
    Mul = ffi::D377_FQ_MUL as isize,
    Square = ffi::D377_FQ_SQUARE as isize,
    Neg = ffi::D377_FQ_NEG as isize,
    Inverse = ffi::D377_FQ_INVERSE as isize,
}

impl Fq {
    pub fn sqrt_ratio_zeta_batch(ctx: &GpuContext, root: SqrtRoot, num: &[Fq], den: &[Fq]) -> Result<Vec<(bool, Fq)>, GpuError> {
        assert_eq!(num.len(), den.len());
        let n = num.len();
        let (a, b) = (pack32(num, |x| x.to_bytes()), pack32(den, |x| x.to_bytes()));
        let mut out = vec![0u8; 32 * n];
        let mut ws = vec![0u8; n];
        check(unsafe {
            ffi::d377_batch_sqrt_ratio_zeta_ex(ctx.0, root as i32, a.as_ptr(), b.as_ptr(), n, out.as_mut_ptr(), ws.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| (ws[i] != 0, fq_from_canonical(&out[32 * i..32 * i + 32]))).collect())
    }
    /// Element-wise operator; `Inverse` yields `None` for zero inputs like `Fq::inverse`.
    pub fn op_batch(ctx: &GpuContext, op: FqOp, a: &[Fq], b: Option<&[Fq]>) -> Result<Vec<Option<Fq>>, GpuError> {
        let n = a.len();
        let la = fqs_to_limbs(a);
        let lb = b.map(|b| {
            assert_eq!(b.len(), n);
            fqs_to_limbs(b)
        });
        let mut out = vec![0u64; 4 * n];
        let mut st = vec![0u8; n];
        check(unsafe {
            ffi::d377_batch_fq_op(ctx.0, op as i32, la.as_ptr(), lb.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()), n,
                                  out.as_mut_ptr(), st.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| if st[i] == 0 { Some(fq_limbs(&out[4 * i..4 * i + 4])) } else { None }).collect())
    }
    /// `Fq::from_le_bytes_mod_order` on 48- or 64-byte strings.
    pub fn from_wide_bytes_batch(ctx: &GpuContext, bytes: &[u8], len: usize) -> Result<Vec<Fq>, GpuError> {
        assert!(len == 48 || len == 64);
        assert_eq!(bytes.len() % len, 0);
        let n = bytes.len() / len;
        let mut out = vec![0u8; 32 * n];
        check(unsafe { ffi::d377_batch_fq_from_wide_bytes(ctx.0, bytes.as_ptr(), len, n, out.as_mut_ptr()) })?;
        Ok(out.chunks_exact(32).map(fq_from_canonical).collect())
    }
    /// `Fq::from_bytes_checked` for every 32-byte string.
    pub fn from_bytes_checked_batch(ctx: &GpuContext, bytes: &[[u8; 32]]) -> Result<Vec<Result<Fq, EncodingError>>, GpuError> {
        let n = bytes.len();
        let mut out = vec![0u64; 4 * n];
        let mut st = vec![0u8; n];
        check(unsafe { ffi::d377_batch_fq_from_bytes_checked(ctx.0, bytes.as_ptr() as *const u8, n, out.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok(results(&st, |i| fq_limbs(&out[4 * i..4 * i + 4])))
    }
    pub fn to_bytes_batch(ctx: &GpuContext, xs: &[Fq]) -> Result<Vec<[u8; 32]>, GpuError> {
        let l = fqs_to_limbs(xs);
        let mut out = vec![[0u8; 32]; xs.len()];
        check(unsafe { ffi::d377_batch_fq_to_bytes(ctx.0, l.as_ptr(), xs.len(), out.as_mut_ptr() as *mut u8) })?;
        Ok(out)
    }
}

impl Fr {
    /// `Fr::from_le_bytes_mod_order` for every 32-byte string -> canonical bytes.
    pub fn from_le_bytes_mod_order_batch(ctx: &GpuContext, bytes: &[[u8; 32]]) -> Result<Vec<[u8; 32]>, GpuError> {
        let mut out = vec![[0u8; 32]; bytes.len()];
        check(unsafe {
            ffi::d377_batch_fr_from_le_bytes_mod_order(ctx.0, bytes.as_ptr() as *const u8, bytes.len(), out.as_mut_ptr() as *mut u8)
        })?;
        Ok(out)
    }
    /// `Fr::from_bytes_checked`: `Err` for strings >= r.
    pub fn from_bytes_checked_batch(ctx: &GpuContext, bytes: &[[u8; 32]]) -> Result<Vec<Result<[u8; 32], EncodingError>>, GpuError> {
        let mut out = vec![[0u8; 32]; bytes.len()];
        let mut st = vec![0u8; bytes.len()];
        check(unsafe {
            ffi::d377_batch_fr_from_bytes_checked(ctx.0, bytes.as_ptr() as *const u8, bytes.len(), out.as_mut_ptr() as *mut u8, st.as_mut_ptr())
        })?;
        Ok(results(&st, |i| out[i]))
    }
}

impl Fr {
    /// Element-wise `Fr` operator (src/fields/fr/u64/wrapper.rs:76-108); `Inverse` yields `None` for zero like `Fr::inverse`.
    pub fn op_batch(ctx: &GpuContext, op: FqOp, a: &[Fr], b: Option<&[Fr]>) -> Result<Vec<Option<Fr>>, GpuError> {
        let n = a.len();
        let la = pack32(a, |x| x.to_bytes());
        let lb = b.map(|b| {
            assert_eq!(b.len(), n);
            pack32(b, |x| x.to_bytes())
        });
        let mut out = vec![0u8; 32 * n];
        let mut st = vec![0u8; n];
        check(unsafe {
            ffi::d377_batch_fr_op(ctx.0, op as i32, la.as_ptr(), lb.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()), n,
                                  out.as_mut_ptr(), st.as_mut_ptr())
        })?;
        Ok((0..n).map(|i| if st[i] == 0 { Some(Fr::from_le_bytes_mod_order(&out[32 * i..32 * i + 32])) } else { None }).collect())
    }
    /// `Fr::from_le_bytes_mod_order` on 48- or 64-byte strings (hash outputs; `Fr::rand` draws 48 bytes).
    pub fn from_wide_bytes_batch(ctx: &GpuContext, bytes: &[u8], len: usize) -> Result<Vec<Fr>, GpuError> {
        assert!(len == 48 || len == 64);
        assert_eq!(bytes.len() % len, 0);
        let n = bytes.len() / len;
        let mut out = vec![0u8; 32 * n];
        check(unsafe { ffi::d377_batch_fr_from_wide_bytes(ctx.0, bytes.as_ptr(), len, n, out.as_mut_ptr()) })?;
        Ok(out.chunks_exact(32).map(Fr::from_le_bytes_mod_order).collect())
    }
}

// ---- device-pointer forms -----------------------------------------------------------------------
/// Batches that already live in HBM (hipMalloc'ed by the caller, 16-byte aligned): no copies, enqueued on
/// `stream` (a `hipStream_t`), no host synchronisation.  Thin `unsafe` pass-throughs: the caller owns the
/// device memory and its lifetime.
pub mod dev {
    use super::*;

    /// `Encoding` records in, `Encoding` records + status bytes out.
    pub unsafe fn scalar_mul_var(ctx: &GpuContext, dev: i32, stream: *mut c_void, enc32: *const u8, scalar32: *const u8, n: usize,
                                 enc32_out: *mut u8, status: *mut u8) -> Result<(), GpuError> {
        check(ffi::d377_batch_scalar_mul_var_dev(ctx.0, dev, stream, enc32, scalar32, n, enc32_out, status))
    }
    /// `Element` records (16 x u64) in and out: `Element * Fr` with no encoding step.
    pub unsafe fn scalar_mul_var_element(ctx: &GpuContext, dev: i32, stream: *mut c_void, xyzt: *const u64, scalar32: *const u8, n: usize,
                                         xyzt_out: *mut u64) -> Result<(), GpuError> {
        check(ffi::d377_batch_scalar_mul_var_element_dev(ctx.0, dev, stream, xyzt, scalar32, n, xyzt_out))
    }
    pub unsafe fn fr_op(ctx: &GpuContext, dev: i32, stream: *mut c_void, op: FqOp, a32: *const u8, b32: *const u8, n: usize, out32: *mut u8,
                        status: *mut u8) -> Result<(), GpuError> {
        check(ffi::d377_batch_fr_op_dev(ctx.0, dev, stream, op as i32, a32, b32, n, out32, status))
    }
    pub unsafe fn scalar_mul_base(ctx: &GpuContext, dev: i32, stream: *mut c_void, scalar32: *const u8, n: usize, enc32_out: *mut u8)
        -> Result<(), GpuError> {
        check(ffi::d377_batch_scalar_mul_base_dev(ctx.0, dev, stream, scalar32, n, enc32_out))
    }
    pub unsafe fn decompress(ctx: &GpuContext, dev: i32, stream: *mut c_void, enc32: *const u8, n: usize, xyzt: *mut u64, status: *mut u8)
        -> Result<(), GpuError> {
        check(ffi::d377_batch_decompress_dev(ctx.0, dev, stream, enc32, n, xyzt, status))
    }
    pub unsafe fn compress(ctx: &GpuContext, dev: i32, stream: *mut c_void, xyzt: *const u64, n: usize, enc32: *mut u8) -> Result<(), GpuError> {
        check(ffi::d377_batch_compress_dev(ctx.0, dev, stream, xyzt, n, enc32))
    }
    pub unsafe fn roundtrip(ctx: &GpuContext, dev: i32, stream: *mut c_void, enc32: *const u8, n: usize, enc32_out: *mut u8, status: *mut u8)
        -> Result<(), GpuError> {
        check(ffi::d377_batch_roundtrip_dev(ctx.0, dev, stream, enc32, n, enc32_out, status))
    }
    pub unsafe fn encode_to_curve(ctx: &GpuContext, dev: i32, stream: *mut c_void, fq32: *const u8, n: usize, enc32_out: *mut u8)
        -> Result<(), GpuError> {
        check(ffi::d377_batch_encode_to_curve_dev(ctx.0, dev, stream, fq32, n, enc32_out))
    }
    pub unsafe fn msm(ctx: &GpuContext, dev: i32, stream: *mut c_void, xyzt: *const u64, scalar32: *const u8, n: usize, enc32_out: *mut u8,
                      xyzt_out: *mut u64) -> Result<(), GpuError> {
        check(ffi::d377_msm_dev(ctx.0, dev, stream, xyzt, scalar32, n, enc32_out, xyzt_out))
    }
    /// An HBM-resident batch on device `root_dev`, split over all the context's GPUs by peer copies (xGMI).
    /// `op` is one of `ffi::D377_OP_*`; buffers are those of the matching `_dev` entry point, unused ones null.
    pub unsafe fn sharded(ctx: &GpuContext, root_dev: i32, stream: *mut c_void, op: i32, in0: *const c_void, in1: *const c_void, n: usize,
                          out0: *mut c_void, out1: *mut c_void) -> Result<(), GpuError> {
        check(ffi::d377_batch_sharded_dev(ctx.0, root_dev, stream, op, in0, in1, n, out0, out1))
    }
}

// decaf377_amd.hpp -- C++ host-side mirror of the reference crate's hot-path API over the C ABI.
//
// Same names and error behaviour as penumbra-zone/decaf377 (v0.10.1), batch-shaped:
//   decaf377::Encoding            Encoding(pub [u8; 32])             src/ark_curve/encoding.rs:14-15
//   decaf377::Element             opaque X, Y, Z, T Montgomery limbs  src/min_curve/element.rs:31-38
//   decaf377::Fq / Fr             32-byte little-endian field elements src/fields/fq.rs, fr.rs
//   decaf377::EncodingError       InvalidEncoding / InvalidSliceLength src/error.rs:1-5
// Header-only; link with -ldecaf377_amd.  All computation happens on the GPU.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "decaf377_amd.h"

namespace decaf377 {

enum class EncodingError { InvalidEncoding, InvalidSliceLength };

struct DeviceError : std::runtime_error {
  explicit DeviceError(int code) : std::runtime_error(std::string("decaf377_amd: ") + d377_last_error()), code(code) {}
  int code;
  // D377_ERR_STARVED: workgroups found no free lane set for 10 s; the call wrote nothing usable (Context health / reset_scratch)
  bool starved() const { return code == D377_ERR_STARVED; }
};

// Result<T, EncodingError>
template <class T>
struct Result {
  bool ok;
  T value;
  EncodingError err;
  const T& unwrap() const {
    if (!ok) throw std::runtime_error("called unwrap() on Err(InvalidEncoding)");
    return value;
  }
  bool is_err() const { return !ok; }
};

struct Bytes32 {
  std::array<uint8_t, 32> b{};
  bool operator==(const Bytes32& o) const { return b == o.b; }
  bool operator!=(const Bytes32& o) const { return !(*this == o); }
};

namespace detail {
// little-endian comparison of a 32-byte string against a modulus given as 4 u64 limbs
inline bool lt_modulus(const std::array<uint8_t, 32>& v, const uint64_t (&m)[4]) {
  for (int i = 3; i >= 0; --i) {
    uint64_t w = 0;
    for (int j = 7; j >= 0; --j) w = (w << 8) | v[8 * i + j];
    if (w != m[i]) return w < m[i];
  }
  return false;
}
}  // namespace detail

struct Fq : Bytes32 {
  // src/fields/fq.rs:29-34
  static constexpr uint64_t MODULUS_LIMBS[4] = {725501752471715841ULL, 6461107452199829505ULL,
                                                6968279316240510977ULL, 1345280370688173398ULL};
  /// Fq::from_le_bytes_mod_order: the bytes are carried as-is; the engine reduces them mod q.
  static Fq from_le_bytes_mod_order(const uint8_t* bytes32) { Fq f; std::memcpy(f.b.data(), bytes32, 32); return f; }
  /// Fq::from_bytes_checked (src/fields/fq.rs:108-115)
  static Result<Fq> from_bytes_checked(const std::array<uint8_t, 32>& bytes) {
    Fq f; f.b = bytes;
    return {detail::lt_modulus(bytes, MODULUS_LIMBS), f, EncodingError::InvalidEncoding};
  }
  static Fq from_u64(uint64_t x) { Fq f; for (int i = 0; i < 8; ++i) f.b[i] = (uint8_t)(x >> (8 * i)); return f; }
  std::array<uint8_t, 32> to_bytes() const { return b; }
};

struct Fr : Bytes32 {
  // src/fields/fr.rs:29-34
  static constexpr uint64_t MODULUS_LIMBS[4] = {13356249993388743167ULL, 5950279507993463550ULL,
                                                10965441865914903552ULL, 336320092672043349ULL};
  static Fr from_le_bytes_mod_order(const uint8_t* bytes32) { Fr f; std::memcpy(f.b.data(), bytes32, 32); return f; }
  static Result<Fr> from_bytes_checked(const std::array<uint8_t, 32>& bytes) {
    Fr f; f.b = bytes;
    return {detail::lt_modulus(bytes, MODULUS_LIMBS), f, EncodingError::InvalidEncoding};
  }
  static Fr from_u64(uint64_t x) { Fr f; for (int i = 0; i < 8; ++i) f.b[i] = (uint8_t)(x >> (8 * i)); return f; }
};

struct Encoding : Bytes32 {
  Encoding() = default;
  explicit Encoding(const std::array<uint8_t, 32>& a) { b = a; }
  /// TryFrom<&[u8]> (src/ark_curve/encoding.rs:131-143): wrong length -> InvalidSliceLength
  static Result<Encoding> try_from(const uint8_t* data, size_t len) {
    Encoding e;
    if (len != 32) return {false, e, EncodingError::InvalidSliceLength};
    std::memcpy(e.b.data(), data, 32);
    return {true, e, EncodingError::InvalidEncoding};
  }
};

struct Element {
  std::array<uint64_t, 16> xyzt{};   // X, Y, Z, T; 4 Montgomery limbs each (R = 2^256)
};

static_assert(sizeof(Encoding) == 32 && sizeof(Fq) == 32 && sizeof(Fr) == 32 && sizeof(Element) == 128,
              "records must be packed");

/// Which backend's root sqrt_ratio_zeta returns: the arkworks one (Sarkar tables, src/ark_curve/invsqrt.rs:75-166)
/// or the min_curve one (Tonelli-Shanks seeded with 11^m, src/min_curve/invsqrt.rs:11-95).  Same flags.
enum class SqrtRoot : int { Ark = D377_SQRT_ROOT_ARK, MinCurve = D377_SQRT_ROOT_MIN_CURVE };

/// Owns one d377_ctx (device tables, streams, scratch).  Calls on one Engine are serialised inside the library.
class Engine {
 public:
  explicit Engine(const std::vector<int>& device_ids = {0}) {
    int rc = d377_ctx_create(device_ids.data(), (int)device_ids.size(), &ctx_);
    if (rc != D377_OK) throw DeviceError(rc);
  }
  /// With the fixed-base comb's options (d377_ctx_create_ex): comb_bits 0 (default: 23), 18, 21 or 23; comb_lazy: the table
  /// is built by the first GENERATOR * Fr batch -- Element::GENERATOR is a constant in the crate (src/min_curve/element.rs:61-81).
  Engine(const std::vector<int>& device_ids, int comb_bits, bool comb_lazy) {
    d377_ctx_opts o{sizeof(d377_ctx_opts), comb_bits, comb_lazy ? 1 : 0};
    int rc = d377_ctx_create_ex(device_ids.data(), (int)device_ids.size(), &o, &ctx_);
    if (rc != D377_OK) throw DeviceError(rc);
  }
  ~Engine() { d377_ctx_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  /// Encoding::vartime_decompress, one Result per input (src/ark_curve/encoding.rs:32-83)
  std::vector<Result<Element>> vartime_decompress(const std::vector<Encoding>& encs) {
    std::vector<Element> out(encs.size());
    std::vector<uint8_t> st(encs.size());
    check(d377_batch_decompress(ctx_, u8(encs), encs.size(), reinterpret_cast<uint64_t*>(out.data()), st.data()));
    return results(out, st);
  }
  /// Element::vartime_compress (src/ark_curve/encoding.rs:116-128)
  std::vector<Encoding> vartime_compress(const std::vector<Element>& els) {
    std::vector<Encoding> out(els.size());
    check(d377_batch_compress(ctx_, reinterpret_cast<const uint64_t*>(els.data()), els.size(), u8m(out)));
    return out;
  }
  /// Element::encode_to_curve, compressed (src/ark_curve/elligator.rs:74-76)
  std::vector<Encoding> encode_to_curve(const std::vector<Fq>& rs) {
    std::vector<Encoding> out(rs.size());
    check(d377_batch_encode_to_curve(ctx_, u8(rs), rs.size(), u8m(out)));
    return out;
  }
  /// Element::hash_to_curve, compressed (src/ark_curve/elligator.rs:67-71)
  std::vector<Encoding> hash_to_curve(const std::vector<Fq>& r1, const std::vector<Fq>& r2) {
    if (r1.size() != r2.size()) throw std::invalid_argument("length mismatch");
    std::vector<Encoding> out(r1.size());
    check(d377_batch_hash_to_curve(ctx_, u8(r1), u8(r2), r1.size(), u8m(out)));
    return out;
  }
  /// Element::GENERATOR * k, compressed
  std::vector<Encoding> mul_generator(const std::vector<Fr>& ks) {
    std::vector<Encoding> out(ks.size());
    check(d377_batch_scalar_mul_base(ctx_, u8(ks), ks.size(), u8m(out)));
    return out;
  }
  /// decompress(P)? * k, compressed (src/min_curve/ops.rs:89-95)
  std::vector<Result<Encoding>> scalar_mul(const std::vector<Encoding>& ps, const std::vector<Fr>& ks) {
    if (ps.size() != ks.size()) throw std::invalid_argument("length mismatch");
    std::vector<Encoding> out(ps.size());
    std::vector<uint8_t> st(ps.size());
    check(d377_batch_scalar_mul_var(ctx_, u8(ps), u8(ks), ps.size(), u8m(out), st.data()));
    return results(out, st);
  }
  /// Element * Fr, Elements in and out (`impl Mul<Fr> for Element`, src/min_curve/ops.rs:89-95)
  std::vector<Element> mul(const std::vector<Element>& ps, const std::vector<Fr>& ks) {
    if (ps.size() != ks.size()) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(ps.size());
    check(d377_batch_scalar_mul_var_element(ctx_, u64(ps), u8(ks), ps.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// Element::GENERATOR * k as Elements
  std::vector<Element> mul_generator_element(const std::vector<Fr>& ks) {
    std::vector<Element> out(ks.size());
    check(d377_batch_scalar_mul_base_element(ctx_, u8(ks), ks.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// Element::encode_to_curve / hash_to_curve as Elements (src/min_curve/element.rs:235-244)
  std::vector<Element> encode_to_curve_element(const std::vector<Fq>& rs) {
    std::vector<Element> out(rs.size());
    check(d377_batch_encode_to_curve_element(ctx_, u8(rs), rs.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  std::vector<Element> hash_to_curve_element(const std::vector<Fq>& r1, const std::vector<Fq>& r2) {
    if (r1.size() != r2.size()) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(r1.size());
    check(d377_batch_hash_to_curve_element(ctx_, u8(r1), u8(r2), r1.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// Element::vartime_compress_to_field (src/min_curve/element.rs:163-181): 4 Montgomery limbs per element
  std::vector<std::array<uint64_t, 4>> vartime_compress_to_field(const std::vector<Element>& els) {
    std::vector<std::array<uint64_t, 4>> out(els.size());
    check(d377_batch_compress_to_field(ctx_, u64(els), els.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// Fr + - * (src/fields/fr/u64/wrapper.rs:88-108); square / neg take b empty; inverse: ok = false for zero (:80-86)
  std::vector<Result<Fr>> fr_op(int op, const std::vector<Fr>& a, const std::vector<Fr>& b = {}) {
    const bool binary = op <= D377_FQ_MUL;
    if (binary && a.size() != b.size()) throw std::invalid_argument("length mismatch");
    std::vector<Fr> out(a.size());
    std::vector<uint8_t> st(a.size());
    check(d377_batch_fr_op(ctx_, op, u8(a), binary ? u8(b) : nullptr, a.size(), u8m(out), st.data()));
    return results(out, st);
  }
  std::vector<Fr> fr_add(const std::vector<Fr>& a, const std::vector<Fr>& b) { return values(fr_op(D377_FQ_ADD, a, b)); }
  std::vector<Fr> fr_sub(const std::vector<Fr>& a, const std::vector<Fr>& b) { return values(fr_op(D377_FQ_SUB, a, b)); }
  std::vector<Fr> fr_mul(const std::vector<Fr>& a, const std::vector<Fr>& b) { return values(fr_op(D377_FQ_MUL, a, b)); }
  /// Fr::from_le_bytes_mod_order on n strings of len = 48 or 64 bytes (src/fields/fr.rs:82-94)
  std::vector<Fr> fr_from_wide_bytes(const std::vector<uint8_t>& bytes, size_t len) {
    if (len == 0 || bytes.size() % len) throw std::invalid_argument("length mismatch");
    std::vector<Fr> out(bytes.size() / len);
    check(d377_batch_fr_from_wide_bytes(ctx_, bytes.data(), len, out.size(), u8m(out)));
    return out;
  }
  /// Fq::sqrt_ratio_zeta (src/ark_curve/invsqrt.rs:75-166; SqrtRoot::MinCurve: src/min_curve/invsqrt.rs:73-95)
  std::vector<std::pair<bool, Fq>> sqrt_ratio_zeta(const std::vector<Fq>& num, const std::vector<Fq>& den,
                                                   SqrtRoot which = SqrtRoot::Ark) {
    if (num.size() != den.size()) throw std::invalid_argument("length mismatch");
    std::vector<Fq> root(num.size());
    std::vector<uint8_t> ws(num.size());
    check(d377_batch_sqrt_ratio_zeta_ex(ctx_, (int)which, u8(num), u8(den), num.size(), u8m(root), ws.data()));
    std::vector<std::pair<bool, Fq>> r(num.size());
    for (size_t i = 0; i < r.size(); ++i) r[i] = {ws[i] != 0, root[i]};
    return r;
  }
  /// Element + Element / double / == (src/min_curve/element.rs:291-340)
  std::vector<Element> add(const std::vector<Element>& p, const std::vector<Element>& q) {
    if (p.size() != q.size()) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(p.size());
    check(d377_batch_add(ctx_, u64(p), u64(q), p.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// Element - Element (src/min_curve/ops.rs:43-87)
  std::vector<Element> sub(const std::vector<Element>& p, const std::vector<Element>& q) {
    if (p.size() != q.size()) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(p.size());
    check(d377_batch_sub(ctx_, u64(p), u64(q), p.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  std::vector<Element> double_(const std::vector<Element>& p) {
    std::vector<Element> out(p.size());
    check(d377_batch_double(ctx_, u64(p), p.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  std::vector<bool> eq(const std::vector<Element>& p, const std::vector<Element>& q) {
    if (p.size() != q.size()) throw std::invalid_argument("length mismatch");
    std::vector<uint8_t> e(p.size());
    check(d377_batch_eq(ctx_, u64(p), u64(q), p.size(), e.data()));
    return std::vector<bool>(e.begin(), e.end());
  }
  /// -Element and Element::is_identity (src/min_curve/element.rs:324-332, 113-117)
  std::vector<Element> neg(const std::vector<Element>& p) {
    std::vector<Element> out(p.size());
    check(d377_batch_neg(ctx_, u64(p), p.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  std::vector<bool> is_identity(const std::vector<Element>& p) {
    std::vector<uint8_t> e(p.size());
    check(d377_batch_is_identity(ctx_, u64(p), p.size(), e.data()));
    return std::vector<bool>(e.begin(), e.end());
  }
  /// Element::vartime_multiscalar_mul(scalars, points) (src/ark_curve/element/projective.rs:99-117)
  Element vartime_multiscalar_mul(const std::vector<Fr>& scalars, const std::vector<Element>& points) {
    if (scalars.size() != points.size()) throw std::invalid_argument("length mismatch");
    Element out;
    Encoding enc;
    check(d377_msm(ctx_, u64(points), u8(scalars), points.size(), enc.b.data(), out.xyzt.data()));
    return out;
  }
  /// The same over Encodings: invalid ones are reported (Err) and left out of the sum.
  std::pair<Element, std::vector<Result<Encoding>>> vartime_multiscalar_mul_encoded(const std::vector<Fr>& scalars,
                                                                                   const std::vector<Encoding>& points) {
    if (scalars.size() != points.size()) throw std::invalid_argument("length mismatch");
    Element out;
    Encoding enc;
    std::vector<uint8_t> st(points.size() ? points.size() : 1);
    check(d377_msm_encoded(ctx_, u8(points), u8(scalars), points.size(), enc.b.data(), out.xyzt.data(), st.data()));
    st.resize(points.size());
    return {out, results(points, st)};
  }
  /// MANY small sums at once: sum i = scalars[i m ..][..m] . points[i m ..][..m], m terms each (1 .. 8) -- the shape of the
  /// crate's own multiscalar test, a 3-term sum per case (tests/operations.rs:44-60).  The sums come back as Elements, as
  /// the crate's function returns them; `encodings`, if given, receives their Encodings, which the same pass computes.
  std::vector<Element> vartime_multiscalar_mul_batch(size_t m, const std::vector<Fr>& scalars, const std::vector<Element>& points,
                                                     std::vector<Encoding>* encodings = nullptr) {
    if (scalars.size() != points.size() || m == 0 || points.size() % m) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(points.size() / m);
    std::vector<Encoding> enc(out.size());
    check(d377_batch_msm_small(ctx_, u64(points), u8(scalars), m, out.size(), u8m(enc), reinterpret_cast<uint64_t*>(out.data())));
    if (encodings) *encodings = std::move(enc);
    return out;
  }
  /// The same over Encodings: an invalid one is reported (Err, per term) and left out of its sum.
  std::pair<std::vector<Element>, std::vector<Result<Encoding>>> vartime_multiscalar_mul_batch_encoded(
      size_t m, const std::vector<Fr>& scalars, const std::vector<Encoding>& points, std::vector<Encoding>* encodings = nullptr) {
    if (scalars.size() != points.size() || m == 0 || points.size() % m) throw std::invalid_argument("length mismatch");
    std::vector<Element> out(points.size() / m);
    std::vector<Encoding> enc(out.size());
    std::vector<uint8_t> st(points.size() ? points.size() : 1);
    check(d377_batch_msm_small_encoded(ctx_, u8(points), u8(scalars), m, out.size(), u8m(enc), reinterpret_cast<uint64_t*>(out.data()), st.data()));
    st.resize(points.size());
    if (encodings) *encodings = std::move(enc);
    return {out, results(points, st)};
  }
  /// CurveGroup::normalize_batch (src/ark_curve/element.rs:74-81): affine (x, y) as 4 Montgomery limbs each
  std::vector<std::array<uint64_t, 8>> normalize_batch(const std::vector<Element>& p) {
    std::vector<std::array<uint64_t, 8>> out(p.size());
    check(d377_batch_to_affine(ctx_, u64(p), p.size(), reinterpret_cast<uint64_t*>(out.data())));
    return out;
  }
  /// decompress then compress: Ok(e) has e == input (tests/encoding.rs:97-107)
  std::vector<Result<Encoding>> roundtrip(const std::vector<Encoding>& encs) {
    std::vector<Encoding> out(encs.size());
    std::vector<uint8_t> st(encs.size());
    check(d377_batch_roundtrip(ctx_, u8(encs), encs.size(), u8m(out), st.data()));
    return results(out, st);
  }
  static Element identity() { Element e; d377_identity(e.xyzt.data()); return e; }     // Element::IDENTITY
  static Element generator() { Element e; d377_generator(e.xyzt.data()); return e; }   // Element::GENERATOR
  d377_ctx* raw() { return ctx_; }

 private:
  template <class T> static const uint8_t* u8(const std::vector<T>& v) { return reinterpret_cast<const uint8_t*>(v.data()); }
  template <class T> static uint8_t* u8m(std::vector<T>& v) { return reinterpret_cast<uint8_t*>(v.data()); }
  static const uint64_t* u64(const std::vector<Element>& v) { return reinterpret_cast<const uint64_t*>(v.data()); }
  static void check(int rc) { if (rc != D377_OK) throw DeviceError(rc); }
  template <class T>
  static std::vector<Result<T>> results(const std::vector<T>& v, const std::vector<uint8_t>& st) {
    std::vector<Result<T>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = {st[i] == 0, v[i], EncodingError::InvalidEncoding};
    return r;
  }
  template <class T>
  static std::vector<T> values(const std::vector<Result<T>>& v) {
    std::vector<T> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = v[i].value;
    return r;
  }
  d377_ctx* ctx_ = nullptr;
};

}  // namespace decaf377

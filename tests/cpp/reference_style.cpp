// The reference's own integration tests (tests/encoding.rs, tests/operations.rs), re-expressed
// against the C++ host mirror (include/decaf377_amd.hpp) and run on the GPU.  Test names follow
// the reference.  Built and run by tests/test_cpp_mirror.py (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "decaf377_amd.hpp"

using namespace decaf377;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static std::array<uint8_t, 32> unhex(const std::string& h) {
  std::array<uint8_t, 32> a{};
  for (int i = 0; i < 32; ++i) a[i] = (uint8_t)std::stoi(h.substr(2 * i, 2), nullptr, 16);
  return a;
}

// tests/encoding.rs:20-26
static void identity_encoding_is_zero(Engine& e) {
  std::vector<Encoding> zero(1);                       // [0; 32]
  auto id = e.vartime_decompress(zero);
  CHECK(id[0].ok);
  auto bytes = e.vartime_compress({id[0].unwrap()});
  CHECK(bytes[0] == zero[0]);
  auto id2 = e.vartime_decompress(bytes);
  CHECK(e.eq({id[0].value}, {id2[0].unwrap()})[0]);
}

// tests/encoding.rs:29-52
static void check_generator(Engine& e) {
  std::vector<Encoding> cand(255);
  for (int b = 1; b <= 255; ++b) cand[b - 1].b[0] = (uint8_t)b;
  auto r = e.vartime_decompress(cand);
  int first = 0;
  for (int b = 1; b <= 255 && !first; ++b) if (r[b - 1].ok) first = b;
  CHECK(first == 8);                                   // the generator [8,0,...] is minimal
  auto enc2 = e.vartime_compress({r[7].unwrap()});
  CHECK(enc2[0] == cand[7]);
  auto g = e.mul_generator({Fr::from_u64(1)});         // Element::GENERATOR
  CHECK(g[0] == cand[7]);
}

// tests/encoding.rs:55-95
static void test_encoding_matches_sage_encoding(Engine& e) {
  const char* expected[16] = {
      "0000000000000000000000000000000000000000000000000000000000000000",
      "0800000000000000000000000000000000000000000000000000000000000000",
      "b2ecf9b9082d6306538be73b0d6ee741141f3222152da78685d6596efc8c1506",
      "2ebd42dd3a2307083c834e79fb9e787e352dd33e0d719f86ae4adb02fe382409",
      "6acd327d70f9588fac373d165f4d9d5300510274dffdfdf2bf0955acd78da50d",
      "460f913e516441c286d95dd30b0a2d2bf14264f325528b06455d7cb93ba13a0b",
      "ec8798bcbb3bf29329549d769f89cf7993e15e2c68ec7aa2a956edf5ec62ae07",
      "48b01e513dd37d94c3b48940dc133b92ccba7f546e99d3fc2e602d284f609f00",
      "a4e85dddd19c80ecf5ef10b9d27b6626ac1a4f90bd10d263c717ecce4da6570a",
      "1a8fea8cbfbc91236d8c7924e3e7e617f9dd544b710ee83827737fe8dc63ae00",
      "0a0f86eaac0c1af30eb138467c49381edb2808904c81a4b81d2b02a2d7816006",
      "588125a8f4e2bab8d16affc4ca60c5f64b50d38d2bb053148021631f72e99b06",
      "f43f4cefbe7326eaab1584722b1b4860de554b23a14490a03f3fd63a089add0b",
      "76c739a33ffd15cf6554a8e705dc573f26490b64de0c5bd4e4ac75ed5af8e60b",
      "200136952d18d3f6c70347032ba3fef4f60c240d706be2950b4f42f1a7087705",
      "bcb0f922df1c7aa9579394020187a2e19e2d8073452c6ab9b0c4b052aa50f505"};
  std::vector<Encoding> encs;
  for (auto h : expected) encs.emplace_back(unhex(h));
  auto pts = e.vartime_decompress(encs);
  std::vector<Element> els;
  for (auto& p : pts) { CHECK(p.ok); els.push_back(p.value); }
  auto back = e.vartime_compress(els);
  std::vector<Element> acc = {els[0]};                 // Element::default()
  const std::vector<Element> basepoint = {els[1]};
  for (int i = 0; i < 16; ++i) {
    CHECK(back[i] == encs[i]);                         // result_hexstr == hexstr
    CHECK(e.eq(acc, {els[i]})[0]);                     // accumulator == point
    acc = e.add(acc, basepoint);                       // accumulator += basepoint
  }
}

// tests/encoding.rs:97-122 (proptests) on seeded inputs
static void round_trips_if_successful(Engine& e) {
  std::mt19937_64 rng(666);
  const size_t n = 20000;
  std::vector<Encoding> raw(n);
  for (auto& x : raw) { for (auto& b : x.b) b = (uint8_t)rng(); x.b[31] &= 0x1f; }
  auto dec = e.vartime_decompress(raw);
  std::vector<Element> ok_el; std::vector<size_t> ok_idx;
  for (size_t i = 0; i < n; ++i) if (dec[i].ok) { ok_el.push_back(dec[i].value); ok_idx.push_back(i); }
  CHECK(ok_el.size() > 1000 && ok_el.size() < n - 1000);
  auto again = e.vartime_compress(ok_el);
  for (size_t j = 0; j < ok_idx.size(); ++j) CHECK(again[j] == raw[ok_idx[j]]);
  // fq / scalar_encoding_round_trip_if_successful: from_bytes_checked accepts exactly the canonical strings
  std::array<uint8_t, 32> ff; ff.fill(0xFF);
  CHECK(Fq::from_bytes_checked(ff).is_err() && Fr::from_bytes_checked(ff).is_err());   // fq.rs:149-152
  std::array<uint8_t, 32> zz{};
  CHECK(Fq::from_bytes_checked(zz).ok && Fr::from_bytes_checked(zz).ok);
  CHECK(Encoding::try_from(ff.data(), 31).err == EncodingError::InvalidSliceLength);
}

// tests/operations.rs:19-43 on seeded inputs: b(aP) = (ab)P needs Fr arithmetic on the host, so the
// commuting form b(aP) == a(bP) is used, plus aP + bP == (a+b)P with small scalars.
static void scalar_mul_properties(Engine& e) {
  std::mt19937_64 rng(667);
  const size_t n = 4096;
  std::vector<Fq> r(n); std::vector<Fr> a(n), b(n);
  for (size_t i = 0; i < n; ++i) { for (auto& x : r[i].b) x = (uint8_t)rng(); for (auto& x : a[i].b) x = (uint8_t)rng(); for (auto& x : b[i].b) x = (uint8_t)rng(); }
  auto P = e.encode_to_curve(r);                       // element_strategy()
  auto unwrap = [](const std::vector<Result<Encoding>>& v) { std::vector<Encoding> o; for (auto& x : v) { CHECK(x.ok); o.push_back(x.value); } return o; };
  auto aP = unwrap(e.scalar_mul(P, a)), bP = unwrap(e.scalar_mul(P, b));
  auto baP = unwrap(e.scalar_mul(aP, b)), abP = unwrap(e.scalar_mul(bP, a));
  for (size_t i = 0; i < n; ++i) CHECK(baP[i] == abP[i]);
  // (a + b) P with 40-bit scalars so the sum is exact on the host
  std::vector<Fr> sa(n), sb(n), sab(n);
  for (size_t i = 0; i < n; ++i) { uint64_t x = rng() >> 24, y = rng() >> 24; sa[i] = Fr::from_u64(x); sb[i] = Fr::from_u64(y); sab[i] = Fr::from_u64(x + y); }
  auto xa = unwrap(e.scalar_mul(P, sa)), xb = unwrap(e.scalar_mul(P, sb)), xab = unwrap(e.scalar_mul(P, sab));
  auto da = e.vartime_decompress(xa), db = e.vartime_decompress(xb);
  std::vector<Element> ea, eb; for (size_t i = 0; i < n; ++i) { ea.push_back(da[i].unwrap()); eb.push_back(db[i].unwrap()); }
  auto sum = e.vartime_compress(e.add(ea, eb));
  for (size_t i = 0; i < n; ++i) CHECK(sum[i] == xab[i]);
  // hash_to_curve = encode_to_curve(r1) + encode_to_curve(r2)   (elligator.rs:67-71)
  std::vector<Fq> r2(r.rbegin(), r.rend());
  auto h = e.hash_to_curve(r, r2);
  auto d1 = e.vartime_decompress(P), d2 = e.vartime_decompress(e.encode_to_curve(r2));
  std::vector<Element> e1, e2; for (size_t i = 0; i < n; ++i) { e1.push_back(d1[i].unwrap()); e2.push_back(d2[i].unwrap()); }
  auto hs = e.vartime_compress(e.add(e1, e2));
  for (size_t i = 0; i < n; ++i) CHECK(h[i] == hs[i]);
}

// tests/operations.rs:44-60: (a*P) + (b*Q) + (c*R) == vartime_multiscalar_mul([a,b,c], [P,Q,R])
static void vartime_multiscalar_mul_matches_scalar_mul(Engine& e) {
  std::mt19937_64 rng(668);
  for (int round = 0; round < 8; ++round) {
    std::vector<Fq> r(3); std::vector<Fr> k(3);
    for (auto& x : r) for (auto& b : x.b) b = (uint8_t)rng();
    for (auto& x : k) for (auto& b : x.b) b = (uint8_t)rng();
    auto enc = e.encode_to_curve(r);
    auto dec = e.vartime_decompress(enc);
    std::vector<Element> pts = {dec[0].unwrap(), dec[1].unwrap(), dec[2].unwrap()};
    auto prods = e.scalar_mul(enc, k);
    auto pd = e.vartime_decompress({prods[0].unwrap(), prods[1].unwrap(), prods[2].unwrap()});
    auto sum = e.add(e.add({pd[0].unwrap()}, {pd[1].unwrap()}), {pd[2].unwrap()});
    Element msm = e.vartime_multiscalar_mul(k, pts);
    CHECK(e.eq(sum, {msm})[0]);
  }
  // the same property for MANY cases in one call (d377_batch_msm_small): 40 three-term sums, Elements and Encodings
  {
    const size_t cases = 40, m = 3;
    std::vector<Fq> r(cases * m); std::vector<Fr> k(cases * m);
    for (auto& x : r) for (auto& b : x.b) b = (uint8_t)rng();
    for (auto& x : k) for (auto& b : x.b) b = (uint8_t)rng();
    auto enc = e.encode_to_curve(r);
    std::vector<Element> pts;
    for (auto& d : e.vartime_decompress(enc)) pts.push_back(d.unwrap());
    std::vector<Encoding> sums_enc, sums_e_enc;
    auto sums = e.vartime_multiscalar_mul_batch(m, k, pts, &sums_enc);                 // Elements, as the crate's function returns
    auto sums_e = e.vartime_multiscalar_mul_batch_encoded(m, k, enc, &sums_e_enc);
    CHECK(sums.size() == cases && sums_e.first.size() == cases && sums_enc.size() == cases && sums_e_enc.size() == cases);
    auto same = e.eq(sums, sums_e.first);
    auto back = e.vartime_compress(sums);
    for (size_t c = 0; c < cases; ++c) {
      std::vector<Fr> kc(k.begin() + c * m, k.begin() + (c + 1) * m);
      std::vector<Element> pc(pts.begin() + c * m, pts.begin() + (c + 1) * m);
      CHECK(e.vartime_compress({e.vartime_multiscalar_mul(kc, pc)})[0] == sums_enc[c]);
      CHECK(same[c] && back[c] == sums_enc[c] && sums_e_enc[c] == sums_enc[c]);
    }
    for (auto& st : sums_e.second) CHECK(st.ok);
  }
  // P + (-P) is the identity; GENERATOR is the decoding of [8, 0, ...]
  auto g = Engine::generator();
  CHECK(e.is_identity(e.add({g}, e.neg({g})))[0]);
  CHECK(e.is_identity(e.sub({g}, {g}))[0]);                // Element - Element, src/min_curve/ops.rs:43-49
  CHECK(e.is_identity({Engine::identity()})[0] && !e.is_identity({g})[0]);
  std::array<uint8_t, 32> eight{}; eight[0] = 8;
  CHECK(e.vartime_compress({g})[0] == Encoding(eight));
}

// src/ark_curve/invsqrt.rs:182-211
static void sqrt_ratio_edge_cases(Engine& e) {
  auto r = e.sqrt_ratio_zeta({Fq::from_u64(0), Fq::from_u64(1)}, {Fq::from_u64(1), Fq::from_u64(0)});
  CHECK(r[0].first == true && r[0].second == Fq::from_u64(0));
  CHECK(r[1].first == false && r[1].second == Fq::from_u64(0));
  // the min_curve backend's root (src/min_curve/invsqrt.rs:73-95): same flag, root equal or negated, and
  // both square to u/v (the reference's own property, invsqrt.rs:182-202)
  std::vector<Fq> u, v;
  for (uint64_t i = 2; i < 66; ++i) { u.push_back(Fq::from_u64(i * i + 7)); v.push_back(Fq::from_u64(3 * i + 1)); }
  auto ra = e.sqrt_ratio_zeta(u, v), rm = e.sqrt_ratio_zeta(u, v, SqrtRoot::MinCurve);
  int differ = 0;
  for (size_t i = 0; i < u.size(); ++i) {
    CHECK(ra[i].first == rm[i].first);
    differ += !(ra[i].second == rm[i].second);
  }
  CHECK(differ > 8 && differ < 56);
}

// tests/operations.rs:19-43 in their literal form, on Elements and with Fr arithmetic on the device:
//   (a * P) + (b * P) == (a + b) * P        b * (a * P) == (a * b) * P
static void operations_proptests_literal(Engine& e) {
  std::mt19937_64 rng(668);
  const size_t n = 2048;
  std::vector<Fq> r(n); std::vector<Fr> a(n), b(n);
  for (size_t i = 0; i < n; ++i) { for (auto& x : r[i].b) x = (uint8_t)rng(); for (auto& x : a[i].b) x = (uint8_t)rng(); for (auto& x : b[i].b) x = (uint8_t)rng(); }
  auto P = e.encode_to_curve_element(r);               // element_strategy()
  auto aP = e.mul(P, a), bP = e.mul(P, b);
  auto lhs = e.add(aP, bP), rhs = e.mul(P, e.fr_add(a, b));
  auto same = e.eq(lhs, rhs);
  for (size_t i = 0; i < n; ++i) CHECK(same[i]);
  auto same2 = e.eq(e.mul(aP, b), e.mul(P, e.fr_mul(a, b)));
  for (size_t i = 0; i < n; ++i) CHECK(same2[i]);
  // Element-form and Encoding-form agree; compress_to_field is the encoding as an Fq; the inverse is an inverse
  auto enc = e.vartime_compress(aP);
  auto enc2 = e.scalar_mul(e.vartime_compress(P), a);
  for (size_t i = 0; i < n; ++i) CHECK(enc2[i].ok && enc2[i].value == enc[i]);
  CHECK(e.vartime_compress_to_field(aP).size() == n);
  auto inv = e.fr_op(D377_FQ_INVERSE, a);
  std::vector<Fr> iv(n); for (size_t i = 0; i < n; ++i) { CHECK(inv[i].ok); iv[i] = inv[i].value; }
  auto one = e.fr_mul(a, iv);
  for (size_t i = 0; i < n; ++i) CHECK(one[i] == Fr::from_u64(1));
  CHECK(!e.fr_op(D377_FQ_INVERSE, {Fr::from_u64(0)})[0].ok);
  auto g1 = e.vartime_compress(e.mul_generator_element({Fr::from_u64(5)}));
  CHECK(g1[0] == e.mul_generator({Fr::from_u64(5)})[0]);
  std::vector<uint8_t> wide(64 * 3, 0xff);
  CHECK(e.fr_from_wide_bytes(wide, 64).size() == 3);
}

// CurveGroup::normalize_batch, Encoding round trip, MSM over encodings
static void widened_entry_points(Engine& e) {
  std::vector<Fr> ks;
  for (uint64_t i = 1; i <= 40; ++i) ks.push_back(Fr::from_u64(i));
  auto encs = e.mul_generator(ks);
  auto rt = e.roundtrip(encs);
  for (size_t i = 0; i < encs.size(); ++i) CHECK(rt[i].ok && rt[i].value == encs[i]);
  std::vector<Element> els;
  for (auto& r : e.vartime_decompress(encs)) els.push_back(r.unwrap());
  auto aff = e.normalize_batch(e.double_(els));
  CHECK(aff.size() == els.size());
  auto both = e.vartime_multiscalar_mul_encoded(ks, encs);
  CHECK(e.eq({both.first}, {e.vartime_multiscalar_mul(ks, els)})[0]);
  for (auto& r : both.second) CHECK(r.ok);
}

int main() {
  Engine e({0});
  identity_encoding_is_zero(e);
  check_generator(e);
  test_encoding_matches_sage_encoding(e);
  round_trips_if_successful(e);
  scalar_mul_properties(e);
  operations_proptests_literal(e);
  vartime_multiscalar_mul_matches_scalar_mul(e);
  sqrt_ratio_edge_cases(e);
  widened_entry_points(e);
  bool threw = false;
  try { Engine bad({99}); } catch (const DeviceError&) { threw = true; }
  CHECK(threw);
  std::printf("CPP_MIRROR_OK\n");
  return 0;
}

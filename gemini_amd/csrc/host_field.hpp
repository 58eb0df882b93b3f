// Host-side BLS12-381 arithmetic used by the product library for the O(1)-sized tails of the
// device algorithms (window Horner over <= 43 window sums, Jacobian normalisation, RCCL-gathered
// partial-point combination, Fiat-Shamir challenge reduction).  64-bit limbs + unsigned __int128.
// Same memory representation as the device code and as ark-ff (Montgomery, R = 2^(64N)).
//
// This is product code (it links into libgemini_hip.so); the test oracle under oracle/ is a
// separate, independent restatement and is never linked here.
#pragma once
#include <cstdint>
#include <cstring>

#include "host_fq_adx.hpp"

namespace gmh {

typedef uint64_t u64;
typedef unsigned __int128 u128;

template <int N>
struct Limbs {
  u64 l[N];
};

template <int N>
static inline bool is_zero(const u64* a) {
  u64 acc = 0;
  for (int i = 0; i < N; i++) acc |= a[i];
  return acc == 0;
}
template <int N>
static inline bool geq(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}
template <int N>
static inline u64 sub_n(u64* r, const u64* a, const u64* b) {
  u64 borrow = 0;
  for (int i = 0; i < N; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}
template <int N>
static inline u64 add_n(u64* r, const u64* a, const u64* b) {
  u64 carry = 0;
  for (int i = 0; i < N; i++) {
    u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (u64)s;
    carry = (u64)(s >> 64);
  }
  return carry;
}

template <int N, class P>
struct Field {
  u64 l[N];

  static Field zero() {
    Field r;
    memset(r.l, 0, sizeof r.l);
    return r;
  }
  static Field one() {
    Field r;
    memcpy(r.l, P::ONE, sizeof r.l);
    return r;
  }
  static Field from_limbs(const u64* p) {
    Field r;
    memcpy(r.l, p, sizeof r.l);
    return r;
  }
  void to_limbs(u64* p) const { memcpy(p, l, sizeof l); }
  bool is_zero() const { return gmh::is_zero<N>(l); }
  bool operator==(const Field& o) const { return memcmp(l, o.l, sizeof l) == 0; }

  // branch-free: the comparisons of a compare-then-subtract form are data-dependent branches that mispredict every other
  // call, and the window Horner of every MSM is a chain of ~2000 dependent field operations
  Field operator+(const Field& o) const {
    Field r, s;
    const u64 c = add_n<N>(r.l, l, o.l);
    const u64 bw = sub_n<N>(s.l, r.l, P::MOD);
    const u64 keep = (u64)0 - (u64)((bw != 0) & (c == 0));  // all ones: r < MOD, keep r
    for (int i = 0; i < N; i++) r.l[i] = (r.l[i] & keep) | (s.l[i] & ~keep);
    return r;
  }
  Field operator-(const Field& o) const {
    Field r, s;
    const u64 bw = sub_n<N>(r.l, l, o.l);
    add_n<N>(s.l, r.l, P::MOD);
    const u64 fix = (u64)0 - (u64)(bw != 0);  // all ones: went negative, take r + MOD
    for (int i = 0; i < N; i++) r.l[i] = (s.l[i] & fix) | (r.l[i] & ~fix);
    return r;
  }
  Field neg() const {
    if (is_zero()) return *this;
    Field r;
    sub_n<N>(r.l, P::MOD, l);
    return r;
  }
  Field dbl() const { return *this + *this; }
  Field operator*(const Field& o) const {
#ifdef GM_HAVE_FQ_ADX
    if constexpr (P::USE_ADX) if (fq_adx_usable()) {  // x86-64 with BMI2 + ADX: host_fq_adx.hpp (Fq only)
      Field r;
      fq_mul_adx(r.l, l, o.l, P::MOD, P::INV);
      return r;
    }
#endif
    return mul_generic(o);
  }
  Field mul_generic(const Field& o) const {
    u64 t[N + 2];
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      u64 c = 0;
      for (int j = 0; j < N; j++) {
        u128 x = (u128)l[j] * o.l[i] + t[j] + c;
        t[j] = (u64)x;
        c = (u64)(x >> 64);
      }
      u128 x = (u128)t[N] + c;
      t[N] = (u64)x;
      t[N + 1] = (u64)(x >> 64);
      u64 m = t[0] * P::INV;
      x = (u128)m * P::MOD[0] + t[0];
      c = (u64)(x >> 64);
      for (int j = 1; j < N; j++) {
        x = (u128)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (u64)x;
        c = (u64)(x >> 64);
      }
      x = (u128)t[N] + c;
      t[N - 1] = (u64)x;
      t[N] = t[N + 1] + (u64)(x >> 64);
    }
    Field r;
    if (t[N] || geq<N>(t, P::MOD)) sub_n<N>(t, t, P::MOD);
    memcpy(r.l, t, sizeof r.l);
    return r;
  }
  Field sqr() const { return *this * *this; }
  Field pow(const u64* e, int n) const {
    Field acc = one();
    for (int i = n * 64 - 1; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
    }
    return acc;
  }
  Field inv() const {  // Fermat; callers never invert zero
    u64 e[N];
    memcpy(e, P::MOD, sizeof e);
    e[0] -= 2;
    return pow(e, N);
  }
  // canonical little-endian integer -> Montgomery and back
  static Field from_canonical(const u64* p) {
    Field a = from_limbs(p), r2;
    memcpy(r2.l, P::R2, sizeof r2.l);
    return a * r2;
  }
  void to_canonical(u64* p) const {
    Field o = zero();
    o.l[0] = 1;
    Field c = *this * o;
    memcpy(p, c.l, sizeof c.l);
  }
};

struct FqP {
  static constexpr bool USE_ADX = true;
  static constexpr u64 MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
  static constexpr u64 INV = 0x89f3fffcfffcfffdULL;
  static constexpr u64 ONE[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                                 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
  static constexpr u64 R2[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                                0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
};
struct FrP {
  static constexpr bool USE_ADX = false;
  static constexpr u64 MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                 0x73eda753299d7d48ULL};
  static constexpr u64 INV = 0xfffffffeffffffffULL;
  static constexpr u64 ONE[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL,
                                 0x1824b159acc5056fULL};
  static constexpr u64 R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL,
                                0x0748d9d99f59ff11ULL};
};
typedef Field<6, FqP> Fq;
typedef Field<4, FrP> Fr;

// Jacobian point, ark-ec `Projective<P>` layout (X, Y, Z), identity Z = 0.
struct G1 {
  Fq x, y, z;
  static G1 identity() {
    G1 r;
    r.x = Fq::one();
    r.y = Fq::one();
    r.z = Fq::zero();
    return r;
  }
  bool is_identity() const { return z.is_zero(); }
  static G1 from_limbs(const u64* p) {
    G1 r;
    r.x = Fq::from_limbs(p);
    r.y = Fq::from_limbs(p + 6);
    r.z = Fq::from_limbs(p + 12);
    return r;
  }
  void to_limbs(u64* p) const {
    x.to_limbs(p);
    y.to_limbs(p + 6);
    z.to_limbs(p + 12);
  }
  G1 dbl() const {  // dbl-2009-l
    if (is_identity()) return *this;
    Fq A = x.sqr(), B = y.sqr(), C = B.sqr();
    Fq t = x + B;
    Fq D = (t.sqr() - A - C).dbl();
    Fq E = A.dbl() + A;
    Fq F = E.sqr();
    G1 r;
    r.x = F - D.dbl();
    r.y = E * (D - r.x) - C.dbl().dbl().dbl();
    r.z = (y * z).dbl();
    return r;
  }
  G1 add(const G1& q) const {  // add-2007-bl with exceptional cases
    if (is_identity()) return q;
    if (q.is_identity()) return *this;
    Fq z1z1 = z.sqr(), z2z2 = q.z.sqr();
    Fq u1 = x * z2z2, u2 = q.x * z1z1;
    Fq s1 = y * q.z * z2z2, s2 = q.y * z * z1z1;
    if (u1 == u2) {
      if (s1 == s2) return dbl();
      return identity();
    }
    Fq h = u2 - u1;
    Fq i = h.dbl().sqr();
    Fq j = h * i;
    Fq rr = (s2 - s1).dbl();
    Fq v = u1 * i;
    G1 r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (s1 * j).dbl();
    r.z = ((z + q.z).sqr() - z1z1 - z2z2) * h;
    return r;
  }
  // (X/Z^2, Y/Z^3, 1): the unique representative, so equal points give equal bytes
  G1 normalized() const {
    if (is_identity()) return identity();
    if (z == Fq::one()) return *this;  // what an MSM hands out already is: the transcript and the proof encoders ask again (an inversion is ~20 us)
    Fq zi = z.inv(), zi2 = zi.sqr();
    G1 r;
    r.x = x * zi2;
    r.y = y * zi2 * zi;
    r.z = Fq::one();
    return r;
  }
};

// Device-resident Fq values are a * 2^390 mod q (g1.cuh, GM_FQ30); ark-ff's form is a * 2^384.
// x_ark = x_dev * 2^-6: as a Montgomery (R = 2^384) operand that constant is 2^378.
#ifndef GM_FQ30
#define GM_FQ30 2
#endif
static inline Fq fq_from_device(const u64* p) {
  Fq v = Fq::from_limbs(p);
#if GM_FQ30
  static const u64 K[6] = {0, 0, 0, 0, 0, 0x0400000000000000ULL};
  return v * Fq::from_limbs(K);
#else
  return v;
#endif
}
// ark-ff form -> device form: x_dev = x_ark * 2^6, i.e. the Montgomery operand 2^390 mod q
static inline void fq_to_device(const Fq& v, u64* out) {
#if GM_FQ30
  static const u64 K[6] = {0x4676000000d1ff2eULL, 0x84b803379b4800acULL, 0x0dd9a7e0e882431cULL, 0xc26c26d0b683dcf8ULL, 0x29f1457663c4a5eeULL, 0x015de9967f3e804bULL};
  (v * Fq::from_limbs(K)).to_limbs(out);
#else
  v.to_limbs(out);
#endif
}
// XYZZ record in device form -> Jacobian in ark-ff form
static inline G1 xyzz_to_jac_dev(const u64* p) {
  Fq X = fq_from_device(p), Y = fq_from_device(p + 6), ZZ = fq_from_device(p + 12), ZZZ = fq_from_device(p + 18);
  if (ZZ.is_zero()) return G1::identity();
  G1 r;
  r.x = X * ZZ;
  r.y = Y * ZZZ;
  r.z = ZZ;
  return r;
}

// extended Jacobian (X, Y, ZZ, ZZZ) -> Jacobian, mirrors device xyzz_to_jac
static inline G1 xyzz_to_jac(const u64* p) {
  Fq X = Fq::from_limbs(p), Y = Fq::from_limbs(p + 6), ZZ = Fq::from_limbs(p + 12), ZZZ = Fq::from_limbs(p + 18);
  if (ZZ.is_zero()) return G1::identity();
  G1 r;
  r.x = X * ZZ;
  r.y = Y * ZZZ;
  r.z = ZZ;
  return r;
}

}  // namespace gmh

// BLS12-381 prime-field arithmetic for gfx950 (CDNA4), written directly for the AMD
// integer pipeline: 32-bit limbs, v_mad_u64_u32 partial products, v_add_co/v_addc carry
// chains.  No MFMA: modular multiplication is a carry-propagating integer recurrence,
// not a dense contraction.
//
// Value representation matches ark-ff 0.4 `Fp<MontBackend<_, N>, N>` in memory
// (little-endian u64 limbs holding a*R mod p, R = 2^(64N)), read here as 2N u32 limbs,
// so Rust `&[Fr]` / `G1Affine` buffers cross the C ABI without conversion
// (SURVEY.md Appendix B).
//
// Replaces: ark-ff Montgomery arithmetic used by every Fr/Fq operator on the Gemini hot
// path, e.g. src/subprotocols/sumcheck/time_prover.rs:105-118, src/misc.rs:52-56.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gm {

#define GM_DEV __device__ __forceinline__
#define GM_HD __host__ __device__ __forceinline__

// ----------------------------------------------------------------------------
// Field parameter packs (limbs are literal constants so they become SGPR/literal operands)
// ----------------------------------------------------------------------------
struct FqParams {
  static constexpr int N = 12;
  // q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
  static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                       0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                       0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  static constexpr uint32_t INV = 0xfffcfffdu;  // -q^{-1} mod 2^32
  // R mod q (Montgomery one)
  static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                       0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                       0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  // R^2 mod q
  static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                      0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                      0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

struct FrParams {
  static constexpr int N = 8;
  // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
  static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                      0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr uint32_t INV = 0xffffffffu;  // -r^{-1} mod 2^32
  static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                      0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

// ----------------------------------------------------------------------------
// Fp<P>: N 32-bit limbs, Montgomery form, always fully reduced (< p) at rest
// ----------------------------------------------------------------------------
template <class P>
struct Fp {
  static constexpr int N = P::N;
  uint32_t l[N];

  static GM_DEV Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  static GM_DEV Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
    return r;
  }
  static GM_DEV Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::R2[i];
    return r;
  }
  GM_DEV bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= l[i];
    return acc == 0;
  }
  GM_DEV bool operator==(const Fp& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
    return acc == 0;
  }
};

// r = a - p if a >= p else a   (a < 2p, `extra` is a possible carry limb above a)
template <class P>
GM_DEV void fp_cond_sub(Fp<P>& a, uint32_t extra) {
  constexpr int N = P::N;
  uint32_t t[N];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t b;
    t[i] = __builtin_subc(a.l[i], P::MOD[i], borrow, &b);
    borrow = b;
  }
  // a >= p  <=>  no final borrow, or the extra limb absorbs it
  bool ge = (extra != 0) | (borrow == 0);
#pragma unroll
  for (int i = 0; i < N; i++) a.l[i] = ge ? t[i] : a.l[i];
}

template <class P>
GM_DEV Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  Fp<P> r;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t c;
    r.l[i] = __builtin_addc(a.l[i], b.l[i], carry, &c);
    carry = c;
  }
  fp_cond_sub<P>(r, carry);  // both moduli leave the top limb < 2^31, so carry is always 0 here
  return r;
}

template <class P>
GM_DEV Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  Fp<P> r;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t c;
    r.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &c);
    borrow = c;
  }
  // add p back if we borrowed
  uint32_t mask = 0u - borrow;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t c;
    r.l[i] = __builtin_addc(r.l[i], P::MOD[i] & mask, carry, &c);
    carry = c;
  }
  return r;
}

template <class P>
GM_DEV Fp<P> fp_neg(const Fp<P>& a) {
  constexpr int N = P::N;
  Fp<P> r;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t c;
    r.l[i] = __builtin_subc(P::MOD[i], a.l[i], borrow, &c);
    borrow = c;
  }
  bool z = a.is_zero();
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = z ? 0u : r.l[i];
  return r;
}

template <class P>
GM_DEV Fp<P> fp_dbl(const Fp<P>& a) {
  return fp_add<P>(a, a);
}

// Montgomery product a*b*R^{-1} mod p.  CIOS over 32-bit limbs; every step is one
// v_mad_u64_u32 (32x32+64) plus a carry fold.  t never exceeds N+1 limbs because
// both moduli satisfy 2p < 2^(32N) ("no-carry" property of the top limb).
template <class P>
GM_DEV Fp<P> fp_mul_cios(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  uint32_t t[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) t[i] = 0;

#pragma unroll
  for (int i = 0; i < N; i++) {
    // t += a * b[i]
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
#pragma unroll
    for (int j = 0; j < N; j++) {
      c = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t top = t[N] + (uint32_t)c;  // cannot overflow (see above)
    // t = (t + m*p) / 2^32
    const uint32_t m = t[0] * P::INV;
    c = (uint64_t)m * P::MOD[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      c = (uint64_t)m * P::MOD[j] + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += top;
    t[N - 1] = (uint32_t)c;
    t[N] = (uint32_t)(c >> 32);
  }
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
  fp_cond_sub<P>(r, t[N]);
  return r;
}

// The production multiplier: product-scanning Montgomery, one v_mad_u64_u32 + v_addc_co_u32 per
// partial product into a 96-bit column accumulator (generated, see gen_field_mul.py).
// fp_mul_cios above is the plain-C form kept as an in-kernel cross-check.
template <class P>
GM_DEV Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b);
#include "field_mul_gen.inc"

template <class P>
GM_DEV Fp<P> fp_sqr(const Fp<P>& a) {
  return fp_mul<P>(a, a);
}

// canonical integer (little-endian limbs) -> Montgomery form and back
template <class P>
GM_DEV Fp<P> fp_to_mont(const Fp<P>& a) {
  return fp_mul<P>(a, Fp<P>::r2());
}
template <class P>
GM_DEV Fp<P> fp_from_mont(const Fp<P>& a) {
  Fp<P> one = Fp<P>::zero();
  one.l[0] = 1;
  return fp_mul<P>(a, one);
}

// a^e for a small public exponent given as a 64-bit integer
template <class P>
GM_DEV Fp<P> fp_pow_u64(const Fp<P>& a, uint64_t e) {
  Fp<P> acc = Fp<P>::one();
  Fp<P> base = a;
  while (e) {
    if (e & 1) acc = fp_mul<P>(acc, base);
    base = fp_sqr<P>(base);
    e >>= 1;
  }
  return acc;
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// 16-byte vector load/store helpers for field elements laid out contiguously (AoS):
// an Fr is 2 x uint4, an Fq 3 x uint4 -> global_load_dwordx4.
template <class P>
GM_DEV Fp<P> fp_load(const void* p) {
  constexpr int N = P::N;
  Fp<P> r;
  const uint4* v = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    uint4 x = v[i];
    r.l[4 * i + 0] = x.x;
    r.l[4 * i + 1] = x.y;
    r.l[4 * i + 2] = x.z;
    r.l[4 * i + 3] = x.w;
  }
  return r;
}
template <class P>
GM_DEV void fp_store(void* p, const Fp<P>& a) {
  constexpr int N = P::N;
  uint4* v = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    v[i] = make_uint4(a.l[4 * i + 0], a.l[4 * i + 1], a.l[4 * i + 2], a.l[4 * i + 3]);
  }
}

}  // namespace gm

// Fq in radix 2^30: 13 limbs, lazy carries, Montgomery factor R' = 2^390.
//
// Why a second representation: on gfx950 v_mad_u64_u32 AND v_addc_co_u32 both issue at half rate
// (tools/ubench_isa.hip), so the 32-bit-limb product-scanning multiplier of field.cuh pays two
// half-rate instructions per partial product (accumulate + carry fold).  With 30-bit limbs thirteen
// 60-bit products fit a 64-bit accumulator, so a partial product is ONE v_mad_u64_u32 and carries
// are extracted once per column with full-rate shifts/ands: 2 x 169 mads instead of 2 x (144 mads
// + 144 addc), written as plain C that hipcc maps 1:1 onto the instruction.
//
// Values are "loose": the integer v = sum l_i 2^(30 i) represents v mod q and is only required to
// stay below 2^386 (32 q).  R' = 2^390 leaves so much headroom that the Montgomery product of two
// loose values is < 3q WITHOUT any conditional subtraction; additions just add limbs and renormalise,
// subtractions add a multiple of q first.  Canonical form (< q) is produced only when a value is
// written to memory (fq30_store), where records keep the 12 x u32 packed layout of field.cuh.
//
// Memory/ABI note: ark-ff keeps a*2^384 mod q.  Device-resident data of the MSM (bases, buckets,
// partials) holds a*2^390 mod q instead; conversion is one multiplication by a constant at the
// boundary (k_pack_bases on the way in, the host on the way out).
#pragma once
#include "field.cuh"

namespace gm {

struct Fq30Params {
  static constexpr int N = 13;
  static constexpr uint32_t MASK = (1u << 30) - 1u;
  // q in radix 2^30
  static constexpr uint32_t MOD[13] = {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x034a83dau, 0x112bf673u, 0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x001a0111u};
  static constexpr uint32_t INV = 0x3ffcfffdu;  // -q^{-1} mod 2^30
};

struct Fq30 {
  uint32_t l[13];
  static GM_DEV Fq30 zero() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = 0;
    return r;
  }
  // exact zero (all limbs): how the identity point is encoded; NOT a test for "== 0 mod q"
  GM_DEV bool is_exact_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) acc |= l[i];
    return acc == 0;
  }
};

// carry propagation: limbs may hold up to 32 bits on entry, < 2^30 on exit (top limb keeps the rest)
GM_DEV void fq30_normalize(Fq30& a) {
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    uint32_t v = a.l[i] + carry;  // callers keep limbs <= 2^32 - 2^3, carry < 4: no wrap
    a.l[i] = v & Fq30Params::MASK;
    carry = v >> 30;
  }
  a.l[12] += carry;
}

// Montgomery product a*b*2^-390 for normalised inputs (every limb < 2^30, value < 2^386); output
// normalised, value < 3q.  13*2^60 + 2^35 < 2^64 bounds every column accumulator.
GM_DEV Fq30 fq30_mul(const Fq30& a, const Fq30& b) {
  using P = Fq30Params;
  uint32_t t[26];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 25; k++) {
#pragma unroll
    for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
    t[k] = (uint32_t)acc & P::MASK;
    acc >>= 30;
  }
  t[25] = (uint32_t)acc;
  // reduction: for k < 13 choose m_k so that column k vanishes mod 2^30
  uint32_t m[13];
  Fq30 r;
  acc = 0;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    acc += t[k];
#pragma unroll
    for (int i = (k > 12 ? k - 12 : 0); i <= (k < 13 ? k - 1 : 12); i++) acc += (uint64_t)m[i] * P::MOD[k - i];
    if (k < 13) {
      m[k] = ((uint32_t)acc * P::INV) & P::MASK;
      acc += (uint64_t)m[k] * P::MOD[0];
    } else {
      r.l[k - 13] = (uint32_t)acc & P::MASK;
    }
    acc >>= 30;
  }
  return r;
}
GM_DEV Fq30 fq30_sqr(const Fq30& a) { return fq30_mul(a, a); }

GM_DEV Fq30 fq30_add(const Fq30& a, const Fq30& b) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + b.l[i];
  fq30_normalize(r);
  return r;
}

// K*q with every limb below the top raised by 2^30 (and the borrow taken from the next limb), so that
// a_i + M_i - b_i never goes negative for a normalised subtrahend b < K*q.  Only the top limb may
// transiently wrap; the carries arriving from below make it non-negative again.
template <int K> struct Fq30SubConst;
template <> struct Fq30SubConst<1> { static constexpr uint32_t M[13] = {0x7fffaaabu, 0x67fbfffeu, 0x553ffffau, 0x6affffabu, 0x70f6241du, 0x434a83d9u, 0x512bf672u, 0x52e13ce0u, 0x6cd76476u, 0x5ed90d2du, 0x69a4b1b9u, 0x7a8e5ff8u, 0x001a0110u}; };
template <> struct Fq30SubConst<2> { static constexpr uint32_t M[13] = {0x7fff5556u, 0x4ff7fffeu, 0x6a7ffff6u, 0x55ffff57u, 0x61ec483cu, 0x469507b4u, 0x6257ece5u, 0x65c279c1u, 0x59aec8edu, 0x7db21a5cu, 0x53496373u, 0x751cbff2u, 0x00340222u}; };
template <> struct Fq30SubConst<4> { static constexpr uint32_t M[13] = {0x7ffeaaacu, 0x5feffffeu, 0x54ffffedu, 0x6bfffeb0u, 0x43d89079u, 0x4d2a0f6au, 0x44afd9cbu, 0x4b84f384u, 0x735d91dcu, 0x7b6434b9u, 0x6692c6e8u, 0x6a397fe5u, 0x00680446u}; };
template <> struct Fq30SubConst<8> { static constexpr uint32_t M[13] = {0x7ffd5558u, 0x7fdffffeu, 0x69ffffdbu, 0x57fffd61u, 0x47b120f4u, 0x5a541ed5u, 0x495fb397u, 0x5709e709u, 0x66bb23b9u, 0x76c86974u, 0x4d258dd2u, 0x5472ffccu, 0x00d0088eu}; };

// a - b + K*q for normalised a, b with b < K*q: value < a + K*q
template <int K>
GM_DEV Fq30 fq30_sub(const Fq30& a, const Fq30& b) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + Fq30SubConst<K>::M[i] - b.l[i];
  fq30_normalize(r);
  return r;
}

struct Fq30Consts {
  // R' mod q (the Montgomery one), 2^396 mod q (ark-ff 2^384-form -> 2^390-form), 2^384 mod q (back)
  static constexpr uint32_t ONE[13] = {0x00d1ff2eu, 0x19d80000u, 0x34800ac4u, 0x2e00cde6u, 0x02431c84u, 0x269f83a2u, 0x3dcf80ddu, 0x09b42da0u, 0x25eec26cu, 0x15d98f12u, 0x04b29f14u, 0x259fcfa0u, 0x00015de9u};
  static constexpr uint32_t CIN[13] = {0x3480cb7fu, 0x3e0c0000u, 0x2042b126u, 0x3f337aafu, 0x3de4b4d1u, 0x1e015cf1u, 0x005c540du, 0x3467b19au, 0x352a6da3u, 0x19d89d19u, 0x2fb9afe6u, 0x3848c817u, 0x0009772fu};
  static constexpr uint32_t COUT[13] = {0x0002fffdu, 0x18240000u, 0x00c00027u, 0x3d0002f1u, 0x0758baebu, 0x22615d4fu, 0x257455f4u, 0x1614dc14u, 0x2c6d77ceu, 0x2a5e895bu, 0x0935c071u, 0x30fea039u, 0x0015f65eu};
  static constexpr uint32_t P0INV = 0x30003u;  // q_0^{-1} mod 2^30
};
GM_DEV Fq30 fq30_const(const uint32_t (&c)[13]) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < 13; i++) r.l[i] = c[i];
  return r;
}

// value == 0 (mod q) for a normalised loose value < 16 q.  If v = k q then k = v_0 q_0^{-1} mod 2^30;
// anything with k >= 16 is rejected after one multiply (all but 2^-26 of the non-zero inputs).
GM_DEV bool fq30_is_zero_modq(const Fq30& a) {
  const uint32_t k = (a.l[0] * Fq30Consts::P0INV) & Fq30Params::MASK;
  if (k >= 16u) return false;
  uint64_t acc = 0;
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    acc += (uint64_t)k * Fq30Params::MOD[i];
    const uint32_t limb = i < 12 ? ((uint32_t)acc & Fq30Params::MASK) : (uint32_t)acc;
    diff |= a.l[i] ^ limb;
    acc >>= 30;
  }
  return diff == 0;
}

// fully reduced representative (< q) of a normalised loose value < 16 q: one product by R' mod q
// brings it below 1.04 q, then a single conditional subtraction
// fq30_canonical_tail takes the product a * ONE (computed by the caller, possibly out of line)
GM_DEV Fq30 fq30_canonical_tail(Fq30 r) {
  uint32_t d[13];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    const uint32_t v = r.l[i] - Fq30Params::MOD[i] - borrow;
    borrow = v >> 31;
    d[i] = i < 12 ? (v & Fq30Params::MASK) : v;
  }
#pragma unroll
  for (int i = 0; i < 13; i++) r.l[i] = borrow ? r.l[i] : d[i];
  return r;
}
GM_DEV Fq30 fq30_canonical(const Fq30& a) { return fq30_canonical_tail(fq30_mul(a, fq30_const(Fq30Consts::ONE))); }

// 12 x u32 packed (the record layout in memory) <-> 13 x 30-bit limbs
GM_DEV Fq30 fq30_unpack(const Fq& x) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    uint32_t v = x.l[w] >> sh;
    if (sh > 2 && w + 1 < 12) v |= x.l[w + 1] << (32 - sh);
    r.l[i] = v & Fq30Params::MASK;
  }
  return r;
}
// requires normalised limbs and a value < 2^384
GM_DEV Fq fq30_pack(const Fq30& a) {
  Fq r;
#pragma unroll
  for (int w = 0; w < 12; w++) {
    // word w holds bits [32w, 32w+32): limb i0 = floor(32w/30) shifted down, plus the next limb(s)
    const int bit = 32 * w, i0 = bit / 30, off = bit - 30 * i0;
    uint32_t v = a.l[i0] >> off;
    v |= a.l[i0 + 1] << (30 - off);
    if (60 - off < 32 && i0 + 2 < 13) v |= a.l[i0 + 2] << (60 - off);
    r.l[w] = v;
  }
  return r;
}

}  // namespace gm

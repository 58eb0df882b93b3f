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
    for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k - 1 : 12); i++) acc += (uint64_t)m[i] * P::MOD[k - i];
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

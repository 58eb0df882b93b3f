// BLS12-381 G1 group arithmetic for the MSM kernels (gfx950).
//
// Buckets are kept in extended Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ with
// ZZ^3 = ZZZ^2.  A bucket += affine base costs 8M + 2S and a bucket += bucket 12M + 2S
// (EFD madd-2008-s / add-2008-s), versus 7M + 4S / 11M + 5S for the Jacobian formulas ark-ec
// uses on the CPU.  All additions are COMPLETE through explicit branches: the benchmark
// inputs of the reference are exactly the degenerate ones (dummy_r1cs makes every scalar
// equal, src/circuit.rs:349-365; the elastic example makes every base the generator,
// examples/snark.rs:59-63), so P + P and P - P do occur inside buckets.
//
// Replaces: `Projective<P>::add_assign(&Affine)` / `add_assign(&Projective)` /
// `double_in_place` of ark-ec 0.4.2 as used by VariableBaseMSM::msm_bigint
// (in-tree statement: src/kzg/msm/variable_base.rs:125-175).
//
// Coordinate arithmetic goes through the `FqE` element layer below.  GM_FQ30 = 0 (default, what ships)
// is the 12 x 32-bit canonical representation of field.cuh.  Two EXPERIMENTAL alternatives are kept,
// parity-green (tools/fq30_check.hip, tests/test_gpu_msm.py) but slower inside the kernels:
//   GM_FQ30 = 1: 13 x 30-bit lazy-carry elements (field30.cuh: one v_mad_u64_u32 per partial product,
//                Montgomery factor 2^390, loose values).  The multiplier is 22 % faster in isolation,
//                but 13-limb XYZZ operands spill (k_acc0 864 B scratch per lane): 4.25 ms vs 3.48 ms.
//   GM_FQ30 = 2: hybrid -- canonical 12 x 32-bit elements, only the product unpacks to 13 x 30 bits.
//                No spills, but the unpack / conditional subtraction / repack cancel the gain: 3.56 ms.
// With GM_FQ30 != 0 every device-resident coordinate (bases, buckets, partials, tables) is stored as
// a * 2^390 mod q in the same 12 x u32 packed, fully reduced record; conversion from / to ark-ff's
// a * 2^384 happens once at the boundary (k_pack_bases, k_export_bases, host plane conversion).
#pragma once
#include "field.cuh"
#include "field30.cuh"

#ifndef GM_FQ30
#define GM_FQ30 2
#endif

namespace gm {

#if GM_FQ30 == 1
// ---- element layer: 13 x 30-bit, loose.  Bounds (multiples of q) are tracked in the comments of the
// group law: products are < 2q whenever bound(a) * bound(b) <= 512, fqe_sub<K> needs b < K q.
using FqE = Fq30;
// The product core is one asm statement on physical registers (gen_field_mul30.py --loose): operands in
// v0..v12 / v13..v25 by the calling convention, result in v0..v12, every temporary a caller-saved register.
#include "field_mul30l_gen.inc"
GM_DEV Fq30 fq30_mul_fn(const Fq30& a, const Fq30& b) {
  return fq30_mul_asm(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12],
                      b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11], b.l[12]);
}
GM_DEV Fq30 fq30_sqr_fn(const Fq30& a) {
  return fq30_sqr_asm(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12]);
}
GM_DEV FqE fq_mul(const FqE& a, const FqE& b) { return fq30_mul_fn(a, b); }
GM_DEV FqE fq_sqr(const FqE& a) { return fq30_sqr_fn(a); }
GM_DEV FqE fq_add(const FqE& a, const FqE& b) { return fq30_add(a, b); }
GM_DEV FqE fq_dbl(const FqE& a) { return fq30_add(a, a); }
template <int K>
GM_DEV FqE fq_sub(const FqE& a, const FqE& b) { return fq30_sub<K>(a, b); }
GM_DEV FqE fq_neg_canonical(const FqE& y) {  // y < q (as loaded); -0 stays the exact zero of the identity record
  FqE r = fq30_sub<1>(FqE::zero(), y);
  const bool z = y.is_exact_zero();
#pragma unroll
  for (int i = 0; i < 13; i++) r.l[i] = z ? 0u : r.l[i];
  return r;
}
GM_DEV bool fq_is_zero_mod(const FqE& a) { return fq30_is_zero_modq(a); }
GM_DEV bool fq_is_exact_zero(const FqE& a) { return a.is_exact_zero(); }
GM_DEV FqE fqe_zero() { return FqE::zero(); }
GM_DEV FqE fqe_one() { return fq30_const(Fq30Consts::ONE); }
GM_DEV FqE fqe_load(const void* p) { return fq30_unpack(fp_load<FqParams>(p)); }
GM_DEV void fqe_store(void* p, const FqE& a) {
  fp_store<FqParams>(p, fq30_pack(fq30_canonical_tail(fq30_mul_fn(a, fq30_const(Fq30Consts::ONE)))));
}
constexpr int FQE_LIMBS = 13;
// ark-ff form (a * 2^384) <-> device form (a * 2^390)
GM_DEV FqE fqe_import(const Fq& ark) { return fq30_mul_fn(fq30_unpack(ark), fq30_const(Fq30Consts::CIN)); }
GM_DEV Fq fqe_export(const FqE& dev) {
  FqE t = fq30_mul_fn(dev, fq30_const(Fq30Consts::COUT));
  return fq30_pack(fq30_canonical_tail(fq30_mul_fn(t, fq30_const(Fq30Consts::ONE))));
}
#elif GM_FQ30 == 2
// ---- hybrid: elements are the canonical 12 x 32-bit records of field.cuh (additions, subtractions,
// comparisons, memory all as in mode 0), only the PRODUCT runs in radix 2^30: unpack both operands to
// 13 x 30-bit limbs, lazy-carry Montgomery product with R' = 2^390 (one v_mad_u64_u32 per partial
// product, no carry instruction), one conditional subtraction (canonical inputs give < 1.002 q), repack.
// Values are therefore a * 2^390 mod q at rest, with the same boundary conversions as mode 1.
using FqE = Fq;
template <int K>
GM_DEV FqE fq_sub(const FqE& a, const FqE& b);
// fq30h_mul_fn / fq30h_sqr_fn: one asm statement each on physical registers (gen_field_mul30.py)
#include "field_mul30_gen.inc"
GM_DEV FqE fq_mul(const FqE& a, const FqE& b) {
  return fq30h_mul_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1], b.l[2], b.l[3],
                      b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
}
GM_DEV FqE fq_sqr(const FqE& a) {
  return fq30h_sqr_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11]);
}
GM_DEV FqE fq_add(const FqE& a, const FqE& b) { return fp_add<FqParams>(a, b); }
GM_DEV FqE fq_dbl(const FqE& a) { return fp_add<FqParams>(a, a); }
template <int K>
GM_DEV FqE fq_sub(const FqE& a, const FqE& b) { return fp_sub<FqParams>(a, b); }
GM_DEV FqE fq_neg_canonical(const FqE& y) { return fp_neg<FqParams>(y); }
GM_DEV bool fq_is_zero_mod(const FqE& a) { return a.is_zero(); }
GM_DEV bool fq_is_exact_zero(const FqE& a) { return a.is_zero(); }
GM_DEV FqE fqe_zero() { return Fq::zero(); }
GM_DEV FqE fqe_one() { return fq30_pack(fq30_const(Fq30Consts::ONE)); }
GM_DEV FqE fqe_load(const void* p) { return fp_load<FqParams>(p); }
GM_DEV void fqe_store(void* p, const FqE& a) { fp_store<FqParams>(p, a); }
constexpr int FQE_LIMBS = 12;
GM_DEV FqE fqe_import(const Fq& ark) { return fq_mul(ark, fq30_pack(fq30_const(Fq30Consts::CIN))); }
GM_DEV Fq fqe_export(const FqE& dev) { return fq_mul(dev, fq30_pack(fq30_const(Fq30Consts::COUT))); }
#else
using FqE = Fq;
// The out-of-line multiplier takes its operands as 24 scalar arguments: hipcc passes a second by-value
// struct through the scratch stack (a 48-byte store + load and a vmcnt drain per call), scalars travel
// in v0..v23.
__device__ __noinline__ Fq fq_mul_fn(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                     uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3,
                                     uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
  Fq a, b;
  a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7; a.l[8] = a8; a.l[9] = a9; a.l[10] = a10; a.l[11] = a11;
  b.l[0] = b0; b.l[1] = b1; b.l[2] = b2; b.l[3] = b3; b.l[4] = b4; b.l[5] = b5; b.l[6] = b6; b.l[7] = b7; b.l[8] = b8; b.l[9] = b9; b.l[10] = b10; b.l[11] = b11;
  return fp_mul<FqParams>(a, b);
}
GM_DEV FqE fq_mul(const FqE& a, const FqE& b) {
  return fq_mul_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1], b.l[2], b.l[3],
                   b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
}
GM_DEV FqE fq_sqr(const FqE& a) { return fq_mul(a, a); }
GM_DEV FqE fq_add(const FqE& a, const FqE& b) { return fp_add<FqParams>(a, b); }
GM_DEV FqE fq_dbl(const FqE& a) { return fp_add<FqParams>(a, a); }
template <int K>
GM_DEV FqE fq_sub(const FqE& a, const FqE& b) { return fp_sub<FqParams>(a, b); }
GM_DEV FqE fq_neg_canonical(const FqE& y) { return fp_neg<FqParams>(y); }
GM_DEV bool fq_is_zero_mod(const FqE& a) { return a.is_zero(); }
GM_DEV bool fq_is_exact_zero(const FqE& a) { return a.is_zero(); }
GM_DEV FqE fqe_zero() { return Fq::zero(); }
GM_DEV FqE fqe_one() { return Fq::one(); }
GM_DEV FqE fqe_load(const void* p) { return fp_load<FqParams>(p); }
GM_DEV void fqe_store(void* p, const FqE& a) { fp_store<FqParams>(p, a); }
constexpr int FQE_LIMBS = 12;
GM_DEV FqE fqe_import(const Fq& ark) { return ark; }
GM_DEV Fq fqe_export(const FqE& dev) { return dev; }
#endif

#if GM_FQ30 == 2
// ---- the bucket accumulator of k_acc0 lives in the product's own representation: 13 x 30-bit LOOSE limbs (field30.cuh),
// the whole mixed addition one asm statement (gen_madd30.py).  Buckets and level-0 partials are written as 208-byte
// records of 4 x 13 limbs (X, Y, ZZ, ZZZ; any representative < 2^386 of the residue; the identity is ZZ == 0 exactly)
// and canonicalised by their consumers (k_merge, k_group_sum) on load: one product by R' mod q per coordinate.
#include "field_mul30l_gen.inc"  // fq30_mul_asm / fq30_sqr_asm: the loose product on registers, out of line
#include "g1_madd30_gen.inc"     // Acc30, g1_madd30_asm
constexpr int XYZZ30_BYTES = 208;

GM_DEV Fq fq30_loose_to_canonical(const Fq30& v) {
  const uint32_t* o = Fq30Consts::ONE;
  Fq30 t = fq30_mul_asm(v.l[0], v.l[1], v.l[2], v.l[3], v.l[4], v.l[5], v.l[6], v.l[7], v.l[8], v.l[9], v.l[10], v.l[11], v.l[12],
                        o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12]);
  return fq30_pack(fq30_canonical_tail(t));  // < q + v q / 2^390 < 1.07 q before the conditional subtraction
}
GM_DEV uint32_t acc30_limb(const Acc30& A, int i) {
  return i < 8 ? A.a0[i] : i < 16 ? A.a1[i - 8] : i < 24 ? A.a2[i - 16] : i < 32 ? A.a3[i - 24] : i < 40 ? A.a4[i - 32] : i < 48 ? A.a5[i - 40] : A.a6[i - 48];
}
GM_DEV void acc30_set_limb(Acc30& A, int i, uint32_t v) {
  if (i < 8) A.a0[i] = v;
  else if (i < 16) A.a1[i - 8] = v;
  else if (i < 24) A.a2[i - 16] = v;
  else if (i < 32) A.a3[i - 24] = v;
  else if (i < 40) A.a4[i - 32] = v;
  else if (i < 48) A.a5[i - 40] = v;
  else A.a6[i - 48] = v;
}
GM_DEV void acc30_zero(Acc30& A) {
  A.a0 = A.a1 = A.a2 = A.a3 = A.a4 = A.a5 = gm_u8v{0, 0, 0, 0, 0, 0, 0, 0};
  A.a6 = gm_u4v{0, 0, 0, 0};
}
// limbs 26..38 are ZZ
GM_DEV void acc30_set_identity(Acc30& A) {
#pragma unroll
  for (int i = 26; i < 39; i++) acc30_set_limb(A, i, 0u);
}
GM_DEV void acc30_store(void* p, const Acc30& A) {
  gm_u4v* o = reinterpret_cast<gm_u4v*>(p);
  o[0] = A.a0.lo; o[1] = A.a0.hi; o[2] = A.a1.lo; o[3] = A.a1.hi; o[4] = A.a2.lo; o[5] = A.a2.hi; o[6] = A.a3.lo;
  o[7] = A.a3.hi; o[8] = A.a4.lo; o[9] = A.a4.hi; o[10] = A.a5.lo; o[11] = A.a5.hi; o[12] = A.a6;
}
#endif

// Affine point, identity encoded as (0, 0) -- not on y^2 = x^3 + 4, hence unambiguous.
// Coordinates of a loaded point are fully reduced (< q).
struct G1Affine {
  FqE x, y;
  GM_DEV bool is_identity() const { return fq_is_exact_zero(x) && fq_is_exact_zero(y); }
};

// Invariants kept by every routine below (GM_FQ30): x < 8q, y < 4q, zz, zzz < 2q; the identity is the
// all-zero record (zz == 0 exactly -- a non-identity point never has zz = 0 mod q).
struct G1Xyzz {
  FqE x, y, zz, zzz;
  GM_DEV bool is_identity() const { return fq_is_exact_zero(zz); }
  static GM_DEV G1Xyzz identity() {
    G1Xyzz r;
    r.x = fqe_zero();
    r.y = fqe_zero();
    r.zz = fqe_zero();
    r.zzz = fqe_zero();
    return r;
  }
  static GM_DEV G1Xyzz from_affine(const G1Affine& p) {
    G1Xyzz r;
    if (p.is_identity()) return identity();
    r.x = p.x;
    r.y = p.y;
    r.zz = fqe_one();
    r.zzz = fqe_one();
    return r;
  }
};

// 2 * (affine p), EFD mdbl-2008-s (a = 0)
GM_DEV G1Xyzz xyzz_dbl_affine(const G1Affine& p) {
  if (p.is_identity() || fq_is_exact_zero(p.y)) return G1Xyzz::identity();
  G1Xyzz r;
  FqE u = fq_dbl(p.y);                                   // < 2q
  FqE v = fq_sqr(u);                                     // < 2q
  FqE w = fq_mul(u, v);                                  // < 2q
  FqE s = fq_mul(p.x, v);                                // < 2q
  FqE xx = fq_sqr(p.x);                                  // < 2q
  FqE m = fq_add(fq_dbl(xx), xx);                        // < 6q
  r.x = fq_sub<4>(fq_sqr(m), fq_dbl(s));                 // 2s < 4q        -> < 6q
  r.y = fq_sub<2>(fq_mul(m, fq_sub<8>(s, r.x)), fq_mul(w, p.y));  // (s - x3) < 10q, 6*10 <= 512 -> < 4q
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2 * p, EFD dbl-2008-s-1 (a = 0)
GM_DEV G1Xyzz xyzz_dbl(const G1Xyzz& p) {
  if (p.is_identity() || fq_is_zero_mod(p.y)) return G1Xyzz::identity();
  G1Xyzz r;
  FqE u = fq_dbl(p.y);                                   // < 8q
  FqE v = fq_sqr(u);                                     // 64 <= 512 -> < 2q
  FqE w = fq_mul(u, v);                                  // < 2q
  FqE s = fq_mul(p.x, v);                                // 8*2 -> < 2q
  FqE xx = fq_sqr(p.x);                                  // 64 -> < 2q
  FqE m = fq_add(fq_dbl(xx), xx);                        // < 6q
  r.x = fq_sub<4>(fq_sqr(m), fq_dbl(s));                 // < 6q
  r.y = fq_sub<2>(fq_mul(m, fq_sub<8>(s, r.x)), fq_mul(w, p.y));  // w*y: 2*4 -> < 2q   -> y3 < 4q
  r.zz = fq_mul(v, p.zz);
  r.zzz = fq_mul(w, p.zzz);
  return r;
}

// acc += q (affine), EFD madd-2008-s with the exceptional cases resolved.
GM_DEV void xyzz_madd(G1Xyzz& acc, const G1Affine& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) {
    acc = G1Xyzz::from_affine(q);
    return;
  }
  FqE u2 = fq_mul(q.x, acc.zz);                          // < 2q
  FqE s2 = fq_mul(q.y, acc.zzz);                         // < 2q
  FqE p = fq_sub<8>(u2, acc.x);                          // acc.x < 8q      -> < 10q
  FqE r = fq_sub<4>(s2, acc.y);                          // acc.y < 4q      -> < 6q
  if (fq_is_zero_mod(p)) {
    if (fq_is_zero_mod(r)) {
      acc = xyzz_dbl_affine(q);
    } else {
      acc = G1Xyzz::identity();
    }
    return;
  }
  // Order chosen for register pressure (every product is an opaque call, so this is the order that runs): at most
  // eight field elements are live at any call -- k_acc0 fits three waves per SIMD only below 168 VGPRs.
  FqE pp = fq_sqr(p);                                    // 100 <= 512 -> < 2q
  acc.zz = fq_mul(acc.zz, pp);
  FqE ppp = fq_mul(p, pp);                               // < 2q            (p dead)
  acc.zzz = fq_mul(acc.zzz, ppp);
  FqE qq = fq_mul(acc.x, pp);                            // 8*2 -> < 2q     (x, pp dead)
  FqE yp = fq_mul(acc.y, ppp);                           // 4*2             (y dead)
  FqE x3 = fq_sub<4>(fq_sub<2>(fq_sqr(r), ppp), fq_dbl(qq));  // (r^2 - ppp) < 4q, 2qq < 4q -> < 8q   (ppp dead)
  acc.y = fq_sub<2>(fq_mul(r, fq_sub<8>(qq, x3)), yp);   // (qq - x3) < 10q, 6*10 -> < 4q
  acc.x = x3;
}

// acc += q (XYZZ), EFD add-2008-s with the exceptional cases resolved.
GM_DEV void xyzz_add(G1Xyzz& acc, const G1Xyzz& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) {
    acc = q;
    return;
  }
  FqE u1 = fq_mul(acc.x, q.zz);                          // 8*2 -> < 2q
  FqE u2 = fq_mul(q.x, acc.zz);
  FqE s1 = fq_mul(acc.y, q.zzz);                         // 4*2
  FqE s2 = fq_mul(q.y, acc.zzz);
  FqE p = fq_sub<2>(u2, u1);                             // < 4q
  FqE r = fq_sub<2>(s2, s1);                             // < 4q
  if (fq_is_zero_mod(p)) {
    if (fq_is_zero_mod(r)) {
      acc = xyzz_dbl(acc);
    } else {
      acc = G1Xyzz::identity();
    }
    return;
  }
  FqE pp = fq_sqr(p);
  FqE ppp = fq_mul(p, pp);
  FqE qq = fq_mul(u1, pp);
  FqE x3 = fq_sub<4>(fq_sub<2>(fq_sqr(r), ppp), fq_dbl(qq));  // < 8q
  FqE y3 = fq_sub<2>(fq_mul(r, fq_sub<8>(qq, x3)), fq_mul(s1, ppp));  // < 4q
  acc.zz = fq_mul(fq_mul(acc.zz, q.zz), pp);
  acc.zzz = fq_mul(fq_mul(acc.zzz, q.zzz), ppp);
  acc.x = x3;
  acc.y = y3;
}

// 96-byte affine / 192-byte XYZZ memory images (AoS, 16-byte aligned -> dwordx4 accesses); values at
// rest are fully reduced
GM_DEV G1Affine g1_load_affine(const void* p) {
  G1Affine r;
  r.x = fqe_load(p);
  r.y = fqe_load(reinterpret_cast<const char*>(p) + 48);
  return r;
}
GM_DEV void g1_store_affine(void* p, const G1Affine& a) {
  fqe_store(p, a.x);
  fqe_store(reinterpret_cast<char*>(p) + 48, a.y);
}
GM_DEV G1Xyzz g1_load_xyzz(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  G1Xyzz r;
  r.x = fqe_load(c);
  r.y = fqe_load(c + 48);
  r.zz = fqe_load(c + 96);
  r.zzz = fqe_load(c + 144);
  return r;
}
#if GM_FQ30 == 1
__device__ __noinline__  // canonicalisation makes the store path long: keep one out-of-line copy
#else
GM_DEV
#endif
void g1_store_xyzz(void* p, const G1Xyzz a) {
  char* c = reinterpret_cast<char*>(p);
  if (a.is_identity()) {  // keep the all-zero encoding exact
    const Fq z = Fq::zero();
    fp_store<FqParams>(c, z);
    fp_store<FqParams>(c + 48, z);
    fp_store<FqParams>(c + 96, z);
    fp_store<FqParams>(c + 144, z);
    return;
  }
  fqe_store(c, a.x);
  fqe_store(c + 48, a.y);
  fqe_store(c + 96, a.zz);
  fqe_store(c + 144, a.zzz);
}

#if GM_FQ30 == 2
// 208-byte loose record (see Acc30) -> canonical XYZZ
GM_DEV G1Xyzz g1_load_xyzz30(const void* p) {
  const gm_u4v* s = reinterpret_cast<const gm_u4v*>(p);
  uint32_t w[52];
#pragma unroll
  for (int i = 0; i < 13; i++) {
    const gm_u4v v = s[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  uint32_t zz = 0;
#pragma unroll
  for (int i = 26; i < 39; i++) zz |= w[i];
  if (zz == 0) return G1Xyzz::identity();
  G1Xyzz r;
  Fq30 t;
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = w[i];
  r.x = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = w[13 + i];
  r.y = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = w[26 + i];
  r.zz = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = w[39 + i];
  r.zzz = fq30_loose_to_canonical(t);
  return r;
}
// canonical XYZZ -> 208-byte record (limbs of the canonical values)
GM_DEV void g1_store_xyzz30(void* p, const G1Xyzz& a) {
  gm_u4v* o = reinterpret_cast<gm_u4v*>(p);
  uint32_t w[52];
  if (a.is_identity()) {
#pragma unroll
    for (int i = 0; i < 52; i++) w[i] = 0;
  } else {
    const Fq30 x = fq30_unpack(a.x), y = fq30_unpack(a.y), zz = fq30_unpack(a.zz), zzz = fq30_unpack(a.zzz);
#pragma unroll
    for (int i = 0; i < 13; i++) {
      w[i] = x.l[i]; w[13 + i] = y.l[i]; w[26 + i] = zz.l[i]; w[39 + i] = zzz.l[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 13; i++) o[i] = gm_u4v{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}
GM_DEV G1Xyzz acc30_to_canonical(const Acc30& A) {
  uint32_t zz = 0;
#pragma unroll
  for (int i = 26; i < 39; i++) zz |= acc30_limb(A, i);
  if (zz == 0) return G1Xyzz::identity();
  G1Xyzz r;
  Fq30 t;
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = acc30_limb(A, i);
  r.x = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = acc30_limb(A, 13 + i);
  r.y = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = acc30_limb(A, 26 + i);
  r.zz = fq30_loose_to_canonical(t);
#pragma unroll
  for (int i = 0; i < 13; i++) t.l[i] = acc30_limb(A, 39 + i);
  r.zzz = fq30_loose_to_canonical(t);
  return r;
}
GM_DEV void acc30_from_canonical(Acc30& A, const G1Xyzz& c) {
  if (c.is_identity()) {
    acc30_zero(A);
    return;
  }
  const Fq30 x = fq30_unpack(c.x), y = fq30_unpack(c.y), zz = fq30_unpack(c.zz), zzz = fq30_unpack(c.zzz);
#pragma unroll
  for (int i = 0; i < 13; i++) {
    acc30_set_limb(A, i, x.l[i]);
    acc30_set_limb(A, 13 + i, y.l[i]);
    acc30_set_limb(A, 26 + i, zz.l[i]);
    acc30_set_limb(A, 39 + i, zzz.l[i]);
  }
}
GM_DEV void acc30_load(Acc30& A, const void* p) {
  const gm_u4v* s = reinterpret_cast<const gm_u4v*>(p);
  const gm_u4v v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3], v4 = s[4], v5 = s[5], v6 = s[6], v7 = s[7], v8 = s[8], v9 = s[9], v10 = s[10], v11 = s[11];
  A.a0 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a1 = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a2 = __builtin_shufflevector(v4, v5, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a3 = __builtin_shufflevector(v6, v7, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a4 = __builtin_shufflevector(v8, v9, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a5 = __builtin_shufflevector(v10, v11, 0, 1, 2, 3, 4, 5, 6, 7);
  A.a6 = s[12];
}
GM_DEV bool acc30_is_identity(const Acc30& A) {
  uint32_t zz = 0;
#pragma unroll
  for (int i = 26; i < 39; i++) zz |= acc30_limb(A, i);
  return zz == 0;
}
GM_DEV void acc30_shfl_xor(Acc30& d, const Acc30& s, int m) {
#pragma unroll
  for (int i = 0; i < 52; i++) acc30_set_limb(d, i, __shfl_xor(acc30_limb(s, i), m));
}
// acc += o, both loose XYZZ (o is consumed).  The statement (gen_madd30.py: gen_add) is complete: identity operands,
// doubling and cancellation are handled inside it.
GM_DEV void acc30_add(Acc30& acc, Acc30& o) { (void)g1_add30_asm(acc, o); }
#endif

}  // namespace gm

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
#pragma once
#include "field.cuh"

namespace gm {

// Fq multiplication is ~700 instructions; inlining ten of them per point addition blows the
// 64 KiB instruction cache, so the group law calls one shared out-of-line copy.
__device__ __noinline__ Fq fq_mul_fn(const Fq a, const Fq b) { return fp_mul<FqParams>(a, b); }

#ifndef GM_FQ_MUL_INLINE
#define GM_FQ_MUL_INLINE 0
#endif
GM_DEV Fq fq_mul(const Fq& a, const Fq& b) {
#if GM_FQ_MUL_INLINE
  return fp_mul<FqParams>(a, b);
#else
  return fq_mul_fn(a, b);
#endif
}
GM_DEV Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }
GM_DEV Fq fq_add(const Fq& a, const Fq& b) { return fp_add<FqParams>(a, b); }
GM_DEV Fq fq_sub(const Fq& a, const Fq& b) { return fp_sub<FqParams>(a, b); }
GM_DEV Fq fq_dbl(const Fq& a) { return fp_add<FqParams>(a, a); }
GM_DEV Fq fq_neg(const Fq& a) { return fp_neg<FqParams>(a); }

// Affine point, identity encoded as (0, 0) -- not on y^2 = x^3 + 4, hence unambiguous.
struct G1Affine {
  Fq x, y;
  GM_DEV bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct G1Xyzz {
  Fq x, y, zz, zzz;
  GM_DEV bool is_identity() const { return zz.is_zero(); }
  static GM_DEV G1Xyzz identity() {
    G1Xyzz r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
  }
  static GM_DEV G1Xyzz from_affine(const G1Affine& p) {
    G1Xyzz r;
    if (p.is_identity()) return identity();
    r.x = p.x;
    r.y = p.y;
    r.zz = Fq::one();
    r.zzz = Fq::one();
    return r;
  }
};

// Jacobian (X, Y, Z): what ark-ec `Projective<P>` holds and what crosses the C ABI.
struct G1Jac {
  Fq x, y, z;
};

GM_DEV G1Affine g1_neg(const G1Affine& p) {
  G1Affine r;
  r.x = p.x;
  r.y = fq_neg(p.y);
  return r;
}

// 2 * (affine p), EFD mdbl-2008-s (a = 0)
GM_DEV G1Xyzz xyzz_dbl_affine(const G1Affine& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  G1Xyzz r;
  Fq u = fq_dbl(p.y);
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s = fq_mul(p.x, v);
  Fq xx = fq_sqr(p.x);
  Fq m = fq_add(fq_dbl(xx), xx);
  r.x = fq_sub(fq_sqr(m), fq_dbl(s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2 * p, EFD dbl-2008-s-1 (a = 0)
GM_DEV G1Xyzz xyzz_dbl(const G1Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  G1Xyzz r;
  Fq u = fq_dbl(p.y);
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s = fq_mul(p.x, v);
  Fq xx = fq_sqr(p.x);
  Fq m = fq_add(fq_dbl(xx), xx);
  r.x = fq_sub(fq_sqr(m), fq_dbl(s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
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
  Fq u2 = fq_mul(q.x, acc.zz);
  Fq s2 = fq_mul(q.y, acc.zzz);
  Fq p = fq_sub(u2, acc.x);
  Fq r = fq_sub(s2, acc.y);
  if (p.is_zero()) {
    if (r.is_zero()) {
      acc = xyzz_dbl_affine(q);
    } else {
      acc = G1Xyzz::identity();
    }
    return;
  }
  Fq pp = fq_sqr(p);
  Fq ppp = fq_mul(p, pp);
  Fq qq = fq_mul(acc.x, pp);
  Fq x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_dbl(qq));
  Fq y3 = fq_sub(fq_mul(r, fq_sub(qq, x3)), fq_mul(acc.y, ppp));
  acc.zz = fq_mul(acc.zz, pp);
  acc.zzz = fq_mul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

// acc += q (XYZZ), EFD add-2008-s with the exceptional cases resolved.
GM_DEV void xyzz_add(G1Xyzz& acc, const G1Xyzz& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) {
    acc = q;
    return;
  }
  Fq u1 = fq_mul(acc.x, q.zz);
  Fq u2 = fq_mul(q.x, acc.zz);
  Fq s1 = fq_mul(acc.y, q.zzz);
  Fq s2 = fq_mul(q.y, acc.zzz);
  Fq p = fq_sub(u2, u1);
  Fq r = fq_sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) {
      acc = xyzz_dbl(acc);
    } else {
      acc = G1Xyzz::identity();
    }
    return;
  }
  Fq pp = fq_sqr(p);
  Fq ppp = fq_mul(p, pp);
  Fq qq = fq_mul(u1, pp);
  Fq x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_dbl(qq));
  Fq y3 = fq_sub(fq_mul(r, fq_sub(qq, x3)), fq_mul(s1, ppp));
  acc.zz = fq_mul(fq_mul(acc.zz, q.zz), pp);
  acc.zzz = fq_mul(fq_mul(acc.zzz, q.zzz), ppp);
  acc.x = x3;
  acc.y = y3;
}

// XYZZ -> Jacobian without inversion: (X*ZZ, Y*ZZZ, ZZ) represents the same point
// (x = X*ZZ / ZZ^2, y = Y*ZZZ / ZZ^3 using ZZ^3 = ZZZ^2).  Identity -> (1, 1, 0) like ark-ec.
GM_DEV G1Jac xyzz_to_jac(const G1Xyzz& p) {
  G1Jac r;
  if (p.is_identity()) {
    r.x = Fq::one();
    r.y = Fq::one();
    r.z = Fq::zero();
    return r;
  }
  r.x = fq_mul(p.x, p.zz);
  r.y = fq_mul(p.y, p.zzz);
  r.z = p.zz;
  return r;
}

// 96-byte affine / 192-byte XYZZ memory images (AoS, 16-byte aligned -> dwordx4 accesses)
GM_DEV G1Affine g1_load_affine(const void* p) {
  G1Affine r;
  r.x = fp_load<FqParams>(p);
  r.y = fp_load<FqParams>(reinterpret_cast<const char*>(p) + 48);
  return r;
}
GM_DEV void g1_store_affine(void* p, const G1Affine& a) {
  fp_store<FqParams>(p, a.x);
  fp_store<FqParams>(reinterpret_cast<char*>(p) + 48, a.y);
}
GM_DEV G1Xyzz g1_load_xyzz(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  G1Xyzz r;
  r.x = fp_load<FqParams>(c);
  r.y = fp_load<FqParams>(c + 48);
  r.zz = fp_load<FqParams>(c + 96);
  r.zzz = fp_load<FqParams>(c + 144);
  return r;
}
GM_DEV void g1_store_xyzz(void* p, const G1Xyzz& a) {
  char* c = reinterpret_cast<char*>(p);
  fp_store<FqParams>(c, a.x);
  fp_store<FqParams>(c + 48, a.y);
  fp_store<FqParams>(c + 96, a.zz);
  fp_store<FqParams>(c + 144, a.zzz);
}

}  // namespace gm

/*
 * gemini_oracle.c -- CPU restatement of the Gemini prover hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Not product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  It restates, in plain C (64-bit limbs, unsigned __int128), the
 * algorithms of the reference path so the HIP implementation has something independent to be
 * bit-compared against, and so the "reference algorithm on the host cores" can be timed.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   - signed-digit Pippenger MSM ........ src/kzg/msm/variable_base.rs:16-19 (window rule),
 *                                         :21-61 (digits), :99-176 (buckets / running sum / Horner).
 *                                         (in-tree copy of ark-ec 0.4.2 VariableBaseMSM::msm_bigint;
 *                                          ark-ec itself is a git dependency that is NOT vendored:
 *                                          arkworks-rs/algebra @ df51425, Cargo.lock:44-46)
 *   - ChunkedPippenger .................. src/kzg/msm/stream_pippenger.rs:209-271
 *   - HashMapPippenger .................. src/kzg/msm/stream_pippenger.rs:143-206
 *   - msm_chunks ........................ src/kzg/space.rs:22-55
 *   - sumcheck TimeProver ............... src/subprotocols/sumcheck/time_prover.rs:75-80, :83-123
 *   - field vector helpers .............. src/misc.rs:37-77, :133-149, :180-218
 *   - KZG open / multi-point quotient ... src/kzg/time.rs:112-145, src/kzg/space.rs:95-166
 *
 * Field/curve arithmetic (ark-ff / ark-ec) is third-party and absent from /root/reference; it is
 * restated from the published definitions: BLS12-381 (q, r, y^2 = x^3 + 4, generator),
 * Montgomery form with R = 2^(64N) exactly as ark-ff's MontBackend keeps values in memory.
 *
 * Parity status: pinned against oracle/pyref.py (independent Python big-int arithmetic), the
 * reference's RNG-free known-answer tests, and the naive definition sum_i s_i * P_i.  Byte-level
 * commitment / transcript parity with a Rust run is UNPINNED (no Rust toolchain in the image and
 * the reference's tests hold no golden bytes).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define NQ 6
#define NR 4

static const u64 Q_MOD[NQ] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 Q_INV = 0x89f3fffcfffcfffdULL;
static const u64 Q_ONE[NQ] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                              0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const u64 Q_R2[NQ] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                             0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};

static const u64 R_MOD[NR] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                              0x73eda753299d7d48ULL};
static const u64 R_INV = 0xfffffffeffffffffULL;
static const u64 R_ONE[NR] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL,
                              0x1824b159acc5056fULL};
static const u64 R_R2[NR] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL,
                             0x0748d9d99f59ff11ULL};

/* --------------------------------------------------------------------------------------------
 * generic Montgomery arithmetic on n 64-bit limbs (n is a compile-time constant at every call
 * site; everything is always_inline so the loops unroll)
 * ------------------------------------------------------------------------------------------ */
#define INL static inline __attribute__((always_inline))

INL int limbs_is_zero(const u64* a, int n) {
  u64 acc = 0;
  for (int i = 0; i < n; i++) acc |= a[i];
  return acc == 0;
}
INL int limbs_eq(const u64* a, const u64* b, int n) {
  u64 acc = 0;
  for (int i = 0; i < n; i++) acc |= a[i] ^ b[i];
  return acc == 0;
}
INL int limbs_geq(const u64* a, const u64* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
INL u64 limbs_sub(u64* r, const u64* a, const u64* b, int n) {
  u64 borrow = 0;
  for (int i = 0; i < n; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}
INL u64 limbs_add(u64* r, const u64* a, const u64* b, int n) {
  u64 carry = 0;
  for (int i = 0; i < n; i++) {
    u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (u64)s;
    carry = (u64)(s >> 64);
  }
  return carry;
}
INL void mod_add(u64* r, const u64* a, const u64* b, const u64* p, int n) {
  u64 t[8];
  u64 c = limbs_add(t, a, b, n);
  if (c || limbs_geq(t, p, n)) limbs_sub(t, t, p, n);
  for (int i = 0; i < n; i++) r[i] = t[i];
}
INL void mod_sub(u64* r, const u64* a, const u64* b, const u64* p, int n) {
  u64 t[8];
  if (limbs_sub(t, a, b, n)) limbs_add(t, t, p, n);
  for (int i = 0; i < n; i++) r[i] = t[i];
}
INL void mod_neg(u64* r, const u64* a, const u64* p, int n) {
  if (limbs_is_zero(a, n)) {
    for (int i = 0; i < n; i++) r[i] = 0;
  } else {
    limbs_sub(r, p, a, n);
  }
}
INL void mont_mul(u64* r, const u64* a, const u64* b, const u64* p, u64 inv, int n) {
  u64 t[10];
  for (int i = 0; i < n + 2; i++) t[i] = 0;
  for (int i = 0; i < n; i++) {
    u64 c = 0;
    for (int j = 0; j < n; j++) {
      u128 x = (u128)a[j] * b[i] + t[j] + c;
      t[j] = (u64)x;
      c = (u64)(x >> 64);
    }
    u128 x = (u128)t[n] + c;
    t[n] = (u64)x;
    t[n + 1] = (u64)(x >> 64);
    u64 m = t[0] * inv;
    x = (u128)m * p[0] + t[0];
    c = (u64)(x >> 64);
    for (int j = 1; j < n; j++) {
      x = (u128)m * p[j] + t[j] + c;
      t[j - 1] = (u64)x;
      c = (u64)(x >> 64);
    }
    x = (u128)t[n] + c;
    t[n - 1] = (u64)x;
    t[n] = t[n + 1] + (u64)(x >> 64);
  }
  if (t[n] || limbs_geq(t, p, n)) limbs_sub(t, t, p, n);
  for (int i = 0; i < n; i++) r[i] = t[i];
}

/* ---- Fq ---- */
typedef struct { u64 l[NQ]; } fq;
#if defined(GO_ADX) && defined(__x86_64__) && defined(__ADX__) && defined(__BMI2__)
#define GO_NATIVE_FQ 1
#include <immintrin.h>
/* timed build: carry-flag chains and branch-free selection (a data-dependent `if (borrow)` mispredicts every other time) */
INL fq fq_add(fq a, fq b) {
  fq t, d, r;
  unsigned long long *tl = (unsigned long long*)t.l, *dl = (unsigned long long*)d.l;
  unsigned char c = 0, bw = 0;
  for (int i = 0; i < NQ; i++) c = _addcarry_u64(c, a.l[i], b.l[i], &tl[i]); /* q < 2^381: no carry out */
  for (int i = 0; i < NQ; i++) bw = _subborrow_u64(bw, t.l[i], Q_MOD[i], &dl[i]);
  const u64 keep = (u64)0 - (u64)bw; /* borrow: t < q, keep t */
  for (int i = 0; i < NQ; i++) r.l[i] = (t.l[i] & keep) | (d.l[i] & ~keep);
  return r;
}
INL fq fq_sub(fq a, fq b) {
  fq d, r;
  unsigned long long *dl = (unsigned long long*)d.l, *rl = (unsigned long long*)r.l;
  unsigned char bw = 0, c = 0;
  for (int i = 0; i < NQ; i++) bw = _subborrow_u64(bw, a.l[i], b.l[i], &dl[i]);
  const u64 m = (u64)0 - (u64)bw;
  for (int i = 0; i < NQ; i++) c = _addcarry_u64(c, d.l[i], Q_MOD[i] & m, &rl[i]);
  return r;
}
#else
INL fq fq_add(fq a, fq b) { fq r; mod_add(r.l, a.l, b.l, Q_MOD, NQ); return r; }
INL fq fq_sub(fq a, fq b) { fq r; mod_sub(r.l, a.l, b.l, Q_MOD, NQ); return r; }
#endif
INL fq fq_neg(fq a) { fq r; mod_neg(r.l, a.l, Q_MOD, NQ); return r; }
#if defined(GO_ADX) && defined(__x86_64__) && defined(__ADX__) && defined(__BMI2__)
/* The TIMED build only (libgemini_oracle_native.so, bench.py's cpu_baseline): the reference runs ark-ff with its `asm` feature
 * (Cargo.toml:77-82: mulx + two carry chains), which the portable __int128 loop above under-represents by 2-3 x.  Same CIOS
 * recurrence, one round per limb of b: t += a b_i on the adox chain with the high halves on the adcx chain, then m = t0 inv and
 * t = (t + m q) / 2^64 the same way; q < 2^381 leaves room for both chains without a seventh word.  The portable library stays
 * the checker; tests/test_oracle_native_cpu.py holds the two equal. */
#define GO_R(OFF)                                                                       \
  "xorl %%eax, %%eax\n\t"                                                               \
  "movq " #OFF "(%[b]), %%rdx\n\t"                                                      \
  "mulxq 0(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r8\n\t adcxq %%r14, %%r9\n\t"         \
  "mulxq 8(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r9\n\t adcxq %%r14, %%r10\n\t"        \
  "mulxq 16(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r10\n\t adcxq %%r14, %%r11\n\t"      \
  "mulxq 24(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r11\n\t adcxq %%r14, %%r12\n\t"      \
  "mulxq 32(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r12\n\t adcxq %%r14, %%r13\n\t"      \
  "mulxq 40(%[a]), %%rax, %%r14\n\t adoxq %%rax, %%r13\n\t"                             \
  "movl $0, %%eax\n\t adcxq %%rax, %%r14\n\t adoxq %%rax, %%r14\n\t"                    \
  "movq %[inv], %%rdx\n\t imulq %%r8, %%rdx\n\t xorl %%eax, %%eax\n\t"                  \
  "mulxq 0(%[q]), %%rax, %%rbx\n\t adcxq %%r8, %%rax\n\t movq %%rbx, %%r8\n\t adcxq %%r9, %%r8\n\t" \
  "mulxq 8(%[q]), %%rax, %%r9\n\t adoxq %%rax, %%r8\n\t adcxq %%r10, %%r9\n\t"          \
  "mulxq 16(%[q]), %%rax, %%r10\n\t adoxq %%rax, %%r9\n\t adcxq %%r11, %%r10\n\t"       \
  "mulxq 24(%[q]), %%rax, %%r11\n\t adoxq %%rax, %%r10\n\t adcxq %%r12, %%r11\n\t"      \
  "mulxq 32(%[q]), %%rax, %%r12\n\t adoxq %%rax, %%r11\n\t adcxq %%r13, %%r12\n\t"      \
  "mulxq 40(%[q]), %%rax, %%r13\n\t adoxq %%rax, %%r12\n\t"                             \
  "movl $0, %%eax\n\t adcxq %%rax, %%r13\n\t adoxq %%r14, %%r13\n\t"
INL fq fq_mul(fq a, fq b) {
  fq t, r;
  u64 inv = Q_INV;
  __asm__ volatile(
      "xorl %%r8d, %%r8d\n\t xorl %%r9d, %%r9d\n\t xorl %%r10d, %%r10d\n\t xorl %%r11d, %%r11d\n\t xorl %%r12d, %%r12d\n\t xorl %%r13d, %%r13d\n\t"
      GO_R(0) GO_R(8) GO_R(16) GO_R(24) GO_R(32) GO_R(40)
      "movq %%r8, 0(%[t])\n\t movq %%r9, 8(%[t])\n\t movq %%r10, 16(%[t])\n\t movq %%r11, 24(%[t])\n\t movq %%r12, 32(%[t])\n\t movq %%r13, 40(%[t])\n\t"
      : "=m"(t)
      : [a] "r"(a.l), [b] "r"(b.l), [q] "r"(Q_MOD), [inv] "m"(inv), [t] "r"(t.l), "m"(a), "m"(b)
      : "rax", "rbx", "rdx", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "cc", "memory");
  { /* t < 2q: one conditional subtraction, branch-free */
    unsigned long long* rl = (unsigned long long*)r.l;
    unsigned char bw = 0;
    for (int i = 0; i < NQ; i++) bw = _subborrow_u64(bw, t.l[i], Q_MOD[i], &rl[i]);
    const u64 keep = (u64)0 - (u64)bw;
    for (int i = 0; i < NQ; i++) r.l[i] = (t.l[i] & keep) | (r.l[i] & ~keep);
  }
  return r;
}
#undef GO_R
#else
INL fq fq_mul(fq a, fq b) { fq r; mont_mul(r.l, a.l, b.l, Q_MOD, Q_INV, NQ); return r; }
#endif
INL fq fq_sqr(fq a) { return fq_mul(a, a); }
INL fq fq_dbl(fq a) { return fq_add(a, a); }
INL int fq_is_zero(fq a) { return limbs_is_zero(a.l, NQ); }
INL int fq_eq(fq a, fq b) { return limbs_eq(a.l, b.l, NQ); }
INL fq fq_one(void) { fq r; memcpy(r.l, Q_ONE, sizeof r.l); return r; }
INL fq fq_zero(void) { fq r; memset(r.l, 0, sizeof r.l); return r; }
static fq fq_pow_limbs(fq a, const u64* e, int n) {
  fq acc = fq_one();
  for (int i = n * 64 - 1; i >= 0; i--) {
    acc = fq_sqr(acc);
    if ((e[i / 64] >> (i % 64)) & 1) acc = fq_mul(acc, a);
  }
  return acc;
}
static fq fq_inv(fq a) { /* a^(q-2) */
  u64 e[NQ];
  memcpy(e, Q_MOD, sizeof e);
  e[0] -= 2; /* low limb of q is ...aaab, no borrow */
  return fq_pow_limbs(a, e, NQ);
}

/* ---- Fr ---- */
typedef struct { u64 l[NR]; } fr;
INL fr fr_add(fr a, fr b) { fr r; mod_add(r.l, a.l, b.l, R_MOD, NR); return r; }
INL fr fr_sub(fr a, fr b) { fr r; mod_sub(r.l, a.l, b.l, R_MOD, NR); return r; }
INL fr fr_mul(fr a, fr b) { fr r; mont_mul(r.l, a.l, b.l, R_MOD, R_INV, NR); return r; }
INL fr fr_sqr(fr a) { return fr_mul(a, a); }
INL int fr_is_zero(fr a) { return limbs_is_zero(a.l, NR); }
INL fr fr_one(void) { fr r; memcpy(r.l, R_ONE, sizeof r.l); return r; }
INL fr fr_zero(void) { fr r; memset(r.l, 0, sizeof r.l); return r; }
INL fr fr_load(const u64* p) { fr r; memcpy(r.l, p, sizeof r.l); return r; }
INL void fr_store(u64* p, fr a) { memcpy(p, a.l, sizeof a.l); }
static fr fr_pow_limbs(fr a, const u64* e, int n) {
  fr acc = fr_one();
  for (int i = n * 64 - 1; i >= 0; i--) {
    acc = fr_sqr(acc);
    if ((e[i / 64] >> (i % 64)) & 1) acc = fr_mul(acc, a);
  }
  return acc;
}

/* canonical <-> Montgomery (exported for tests) */
void go_fr_to_mont(const u64* in, u64* out, size_t n) {
  fr r2; memcpy(r2.l, R_R2, sizeof r2.l);
  for (size_t i = 0; i < n; i++) fr_store(out + 4 * i, fr_mul(fr_load(in + 4 * i), r2));
}
void go_fr_from_mont(const u64* in, u64* out, size_t n) {
  fr one = fr_zero(); one.l[0] = 1;
  for (size_t i = 0; i < n; i++) fr_store(out + 4 * i, fr_mul(fr_load(in + 4 * i), one));
}
void go_fq_to_mont(const u64* in, u64* out, size_t n) {
  fq r2; memcpy(r2.l, Q_R2, sizeof r2.l);
  for (size_t i = 0; i < n; i++) { fq a; memcpy(a.l, in + 6 * i, 48); a = fq_mul(a, r2); memcpy(out + 6 * i, a.l, 48); }
}
void go_fq_from_mont(const u64* in, u64* out, size_t n) {
  fq one = fq_zero(); one.l[0] = 1;
  for (size_t i = 0; i < n; i++) { fq a; memcpy(a.l, in + 6 * i, 48); a = fq_mul(a, one); memcpy(out + 6 * i, a.l, 48); }
}
void go_fr_mul(const u64* a, const u64* b, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_store(out + 4 * i, fr_mul(fr_load(a + 4 * i), fr_load(b + 4 * i)));
}
void go_fq_mul(const u64* a, const u64* b, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) { fq x, y; memcpy(x.l, a + 6 * i, 48); memcpy(y.l, b + 6 * i, 48); x = fq_mul(x, y); memcpy(out + 6 * i, x.l, 48); }
}
void go_fr_inv(const u64* a, u64* out) { /* a^(r-2), Montgomery in/out */
  u64 e[NR]; memcpy(e, R_MOD, sizeof e); e[0] -= 2;
  fr_store(out, fr_pow_limbs(fr_load(a), e, NR));
}

/* --------------------------------------------------------------------------------------------
 * G1 (y^2 = x^3 + 4).  Affine = {x, y} Montgomery, identity encoded as x = y = 0 (not on the
 * curve, so unambiguous).  Jacobian = {X, Y, Z}, identity Z = 0 -- ark-ec `Projective<P>`.
 * ------------------------------------------------------------------------------------------ */
typedef struct { fq x, y; } g1a;
typedef struct { fq x, y, z; } g1j;

INL int g1a_is_inf(const g1a* p) { return fq_is_zero(p->x) && fq_is_zero(p->y); }
INL g1j g1j_zero(void) { g1j r; r.x = fq_one(); r.y = fq_one(); r.z = fq_zero(); return r; }
INL int g1j_is_zero(const g1j* p) { return fq_is_zero(p->z); }

static g1j g1j_double(const g1j* p) { /* dbl-2009-l, a = 0 */
  if (g1j_is_zero(p)) return *p;
  fq A = fq_sqr(p->x), B = fq_sqr(p->y), C = fq_sqr(B);
  fq t = fq_add(p->x, B);
  fq D = fq_dbl(fq_sub(fq_sub(fq_sqr(t), A), C));
  fq E = fq_add(fq_dbl(A), A);
  fq F = fq_sqr(E);
  g1j r;
  r.x = fq_sub(F, fq_dbl(D));
  fq c8 = fq_dbl(fq_dbl(fq_dbl(C)));
  r.y = fq_sub(fq_mul(E, fq_sub(D, r.x)), c8);
  r.z = fq_dbl(fq_mul(p->y, p->z));
  return r;
}
static g1j g1j_add(const g1j* p, const g1j* q) { /* add-2007-bl, complete via branches */
  if (g1j_is_zero(p)) return *q;
  if (g1j_is_zero(q)) return *p;
  fq z1z1 = fq_sqr(p->z), z2z2 = fq_sqr(q->z);
  fq u1 = fq_mul(p->x, z2z2), u2 = fq_mul(q->x, z1z1);
  fq s1 = fq_mul(fq_mul(p->y, q->z), z2z2), s2 = fq_mul(fq_mul(q->y, p->z), z1z1);
  if (fq_eq(u1, u2)) {
    if (fq_eq(s1, s2)) return g1j_double(p);
    return g1j_zero();
  }
  fq h = fq_sub(u2, u1);
  fq i = fq_sqr(fq_dbl(h));
  fq j = fq_mul(h, i);
  fq rr = fq_dbl(fq_sub(s2, s1));
  fq v = fq_mul(u1, i);
  g1j r;
  r.x = fq_sub(fq_sub(fq_sqr(rr), j), fq_dbl(v));
  r.y = fq_sub(fq_mul(rr, fq_sub(v, r.x)), fq_dbl(fq_mul(s1, j)));
  fq zz = fq_add(p->z, q->z);
  r.z = fq_mul(fq_sub(fq_sub(fq_sqr(zz), z1z1), z2z2), h);
  return r;
}
static g1j g1j_add_mixed(const g1j* p, const g1a* q) { /* madd-2007-bl */
  if (g1a_is_inf(q)) return *p;
  if (g1j_is_zero(p)) { g1j r; r.x = q->x; r.y = q->y; r.z = fq_one(); return r; }
  fq z1z1 = fq_sqr(p->z);
  fq u2 = fq_mul(q->x, z1z1);
  fq s2 = fq_mul(fq_mul(q->y, p->z), z1z1);
  if (fq_eq(p->x, u2)) {
    if (fq_eq(p->y, s2)) return g1j_double(p);
    return g1j_zero();
  }
  fq h = fq_sub(u2, p->x);
  fq hh = fq_sqr(h);
  fq i = fq_dbl(fq_dbl(hh));
  fq j = fq_mul(h, i);
  fq rr = fq_dbl(fq_sub(s2, p->y));
  fq v = fq_mul(p->x, i);
  g1j r;
  r.x = fq_sub(fq_sub(fq_sqr(rr), j), fq_dbl(v));
  r.y = fq_sub(fq_mul(rr, fq_sub(v, r.x)), fq_dbl(fq_mul(p->y, j)));
  fq zz = fq_add(p->z, h);
  r.z = fq_sub(fq_sub(fq_sqr(zz), z1z1), hh);
  return r;
}
INL g1a g1a_neg(const g1a* p) { g1a r; r.x = p->x; r.y = fq_neg(p->y); return r; }
static g1a g1j_to_affine(const g1j* p) {
  g1a r;
  if (g1j_is_zero(p)) { r.x = fq_zero(); r.y = fq_zero(); return r; }
  fq zi = fq_inv(p->z), zi2 = fq_sqr(zi);
  r.x = fq_mul(p->x, zi2);
  r.y = fq_mul(p->y, fq_mul(zi2, zi));
  return r;
}

/* exported point utilities */
void go_g1_to_affine(const u64* jac18, u64* aff12) {
  g1j p; memcpy(&p, jac18, sizeof p);
  g1a a = g1j_to_affine(&p);
  memcpy(aff12, &a, sizeof a);
}
void go_g1_add(const u64* a18, const u64* b18, u64* out18) {
  g1j p, q; memcpy(&p, a18, sizeof p); memcpy(&q, b18, sizeof q);
  g1j r = g1j_add(&p, &q);
  memcpy(out18, &r, sizeof r);
}
int go_g1_is_on_curve(const u64* aff12) {
  g1a a; memcpy(&a, aff12, sizeof a);
  if (g1a_is_inf(&a)) return 1;
  fq four = fq_dbl(fq_dbl(fq_one()));
  fq lhs = fq_sqr(a.y), rhs = fq_add(fq_mul(fq_sqr(a.x), a.x), four);
  return fq_eq(lhs, rhs);
}
/* projective equality: X1 Z2^2 == X2 Z1^2 and Y1 Z2^3 == Y2 Z1^3 (ark-ec Projective::eq) */
int go_g1_jac_eq(const u64* a18, const u64* b18) {
  g1j p, q; memcpy(&p, a18, sizeof p); memcpy(&q, b18, sizeof q);
  if (g1j_is_zero(&p)) return g1j_is_zero(&q);
  if (g1j_is_zero(&q)) return 0;
  fq z1z1 = fq_sqr(p.z), z2z2 = fq_sqr(q.z);
  if (!fq_eq(fq_mul(p.x, z2z2), fq_mul(q.x, z1z1))) return 0;
  return fq_eq(fq_mul(p.y, fq_mul(z2z2, q.z)), fq_mul(q.y, fq_mul(z1z1, p.z)));
}

/* scalar (canonical, 4 limbs) times affine point: double-and-add = `mul_bigint` */
static g1j g1_mul_bigint(const g1a* p, const u64* k) {
  g1j acc = g1j_zero();
  int started = 0;
  for (int i = 255; i >= 0; i--) {
    if (started) acc = g1j_double(&acc);
    if ((k[i / 64] >> (i % 64)) & 1) { acc = g1j_add_mixed(&acc, p); started = 1; }
  }
  return acc;
}
void go_g1_mul(const u64* aff12, const u64* k4, u64* out18) {
  g1a a; memcpy(&a, aff12, sizeof a);
  g1j r = g1_mul_bigint(&a, k4);
  memcpy(out18, &r, sizeof r);
}

/* naive definition: sum_i s_i * P_i  (src/kzg/msm/variable_base.rs:182-194) */
void go_msm_naive(const u64* bases, const u64* scalars, size_t n, u64* out18) {
  g1j acc = g1j_zero();
  for (size_t i = 0; i < n; i++) {
    g1a p; memcpy(&p, bases + 12 * i, sizeof p);
    g1j t = g1_mul_bigint(&p, scalars + 4 * i);
    acc = g1j_add(&acc, &t);
  }
  memcpy(out18, &acc, sizeof acc);
}

/* --------------------------------------------------------------------------------------------
 * Pippenger, restated from src/kzg/msm/variable_base.rs
 * ------------------------------------------------------------------------------------------ */
static unsigned ceil_log2(size_t n) { /* ark_std::log2 */
  if (n <= 1) return 0;
  unsigned b = 0; size_t v = n - 1;
  while (v) { b++; v >>= 1; }
  return b;
}
/* variable_base.rs:16-19 */
static size_t ln_without_floats(size_t a) { return (size_t)(ceil_log2(a) * 69 / 100); }
/* variable_base.rs:105-109 */
size_t go_msm_window(size_t size) { return size < 32 ? 3 : ln_without_floats(size) + 2; }

/* variable_base.rs:21-61: signed c-bit digits of a canonical scalar */
void go_signed_digits(const u64* scalar, size_t w, size_t num_bits, int64_t* digits, size_t digits_count) {
  u64 radix = (u64)1 << w, window_mask = radix - 1, carry = 0;
  (void)num_bits;
  for (size_t i = 0; i < digits_count; i++) {
    size_t bit_offset = i * w, u64_idx = bit_offset / 64, bit_idx = bit_offset % 64;
    u64 bit_buf;
    if (bit_idx < 64 - w || u64_idx == NR - 1) bit_buf = scalar[u64_idx] >> bit_idx;
    else bit_buf = (scalar[u64_idx] >> bit_idx) | (scalar[1 + u64_idx] << (64 - bit_idx));
    u64 coef = carry + (bit_buf & window_mask);
    carry = (coef + radix / 2) >> w;
    digits[i] = (int64_t)coef - (int64_t)(carry << w);
  }
  digits[digits_count - 1] += (int64_t)(carry << w);
}

/* variable_base.rs:99-176.  One task per window (`parallel` feature of ark-ec) when built with
 * OpenMP; `threads` <= 0 means "all".  bases: n x 12 u64 (affine Montgomery, (0,0)=identity);
 * scalars: n x 4 u64 canonical.  c_override = 0 uses the reference's window rule. */
void go_msm_pippenger_c(const u64* bases, const u64* scalars, size_t n, u64* out18, int threads, size_t c_override) {
  g1j zero = g1j_zero();
  if (n == 0) { memcpy(out18, &zero, sizeof zero); return; }
  size_t c = c_override ? c_override : go_msm_window(n);
  size_t num_bits = 255;
  size_t digits_count = (num_bits + c - 1) / c;
  int64_t* digits = (int64_t*)malloc(sizeof(int64_t) * n * digits_count);
  for (size_t i = 0; i < n; i++) go_signed_digits(scalars + 4 * i, c, num_bits, digits + i * digits_count, digits_count);
  g1j* window_sums = (g1j*)malloc(sizeof(g1j) * digits_count);
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
  for (size_t w = 0; w < digits_count; w++) {
    size_t nb = (size_t)1 << c; /* "We only have 2^(c-1) buckets" but the reference allocates 1<<c */
    g1j* buckets = (g1j*)malloc(sizeof(g1j) * nb);
    for (size_t b = 0; b < nb; b++) buckets[b] = zero;
    for (size_t i = 0; i < n; i++) {
      int64_t s = digits[i * digits_count + w];
      const g1a* base = (const g1a*)(bases + 12 * i);
      if (s > 0) {
        buckets[s - 1] = g1j_add_mixed(&buckets[s - 1], base);
      } else if (s < 0) {
        g1a nb_ = g1a_neg(base);
        buckets[-s - 1] = g1j_add_mixed(&buckets[-s - 1], &nb_);
      }
    }
    g1j running = zero, res = zero;
    for (size_t b = nb; b-- > 0;) {
      running = g1j_add(&running, &buckets[b]);
      res = g1j_add(&res, &running);
    }
    window_sums[w] = res;
    free(buckets);
  }
  g1j total = window_sums[digits_count - 1];
  for (size_t w = digits_count - 1; w-- > 0;) {
    for (size_t k = 0; k < c; k++) total = g1j_double(&total);
    total = g1j_add(&total, &window_sums[w]);
  }
  free(window_sums);
  free(digits);
  memcpy(out18, &total, sizeof total);
}
void go_msm_pippenger(const u64* bases, const u64* scalars, size_t n, u64* out18, int threads) {
  go_msm_pippenger_c(bases, scalars, n, out18, threads, 0);
}

/* ChunkedPippenger (stream_pippenger.rs:209-271): flush every buf_size pairs, result += msm. */
void go_chunked_pippenger(const u64* bases, const u64* scalars, size_t n, size_t buf_size, u64* out18, int threads) {
  g1j result = g1j_zero();
  size_t off = 0;
  while (off < n) {
    size_t m = n - off < buf_size ? n - off : buf_size;
    g1j part;
    go_msm_pippenger(bases + 12 * off, scalars + 4 * off, m, (u64*)&part, threads);
    result = g1j_add(&result, &part);
    off += m;
  }
  memcpy(out18, &result, sizeof result);
}

/* msm_chunks (src/kzg/space.rs:22-55): align by skipping len(bases) - len(scalars) bases, then
 * 2^20-pair steps.  Streams are passed as arrays in stream order. */
void go_msm_chunks(const u64* bases, size_t nbases, const u64* scalars, size_t nscalars, u64* out18, int threads) {
  const u64* b = bases + 12 * (nbases - nscalars);
  go_chunked_pippenger(b, scalars, nscalars, (size_t)1 << 20, out18, threads);
}

/* HashMapPippenger (stream_pippenger.rs:143-206): scalars of equal bases are added in Fr before
 * the MSM; flush when `cap` distinct bases are held.  `scalars` are Montgomery Fr (the reference
 * takes `ScalarField`, :162-172) and are converted with into_bigint at flush (:176-180).
 * Open-addressing table keyed by the 96 base bytes; iteration order differs from hashbrown's,
 * which cannot change the sum. */
static u64 hash96(const u64* k) {
  u64 h = 0xcbf29ce484222325ULL;
  for (int i = 0; i < 12; i++) { h ^= k[i]; h *= 0x100000001b3ULL; h ^= h >> 29; }
  return h;
}
void go_hashmap_pippenger(const u64* bases, const u64* scalars_mont, size_t n, size_t cap, u64* out18, int threads) {
  size_t tsize = 1;
  while (tsize < 2 * cap + 2) tsize <<= 1;
  u64* keys = (u64*)malloc(96 * cap);
  fr* vals = (fr*)malloc(sizeof(fr) * cap);
  int64_t* table = (int64_t*)malloc(sizeof(int64_t) * tsize);
  u64* flush_scalars = (u64*)malloc(32 * cap);
  g1j result = g1j_zero();
  size_t used = 0;
  for (size_t t = 0; t < tsize; t++) table[t] = -1;
  fr one_raw = fr_zero(); one_raw.l[0] = 1;
  for (size_t i = 0; i <= n; i++) {
    if (i < n) {
      const u64* k = bases + 12 * i;
      size_t h = hash96(k) & (tsize - 1);
      while (table[h] >= 0 && memcmp(keys + 12 * table[h], k, 96) != 0) h = (h + 1) & (tsize - 1);
      if (table[h] < 0) {
        table[h] = (int64_t)used;
        memcpy(keys + 12 * used, k, 96);
        vals[used] = fr_zero();
        used++;
      }
      vals[table[h]] = fr_add(vals[table[h]], fr_load(scalars_mont + 4 * i));
    }
    if ((i < n && used == cap) || (i == n && used > 0)) {
      for (size_t j = 0; j < used; j++) fr_store(flush_scalars + 4 * j, fr_mul(vals[j], one_raw));
      g1j part;
      go_msm_pippenger(keys, flush_scalars, used, (u64*)&part, threads);
      result = g1j_add(&result, &part);
      used = 0;
      for (size_t t = 0; t < tsize; t++) table[t] = -1;
    }
  }
  free(keys); free(vals); free(table); free(flush_scalars);
  memcpy(out18, &result, sizeof result);
}

/* --------------------------------------------------------------------------------------------
 * Fixed-base helpers used to build test inputs / an SRS (src/kzg/time.rs:49-59 computes
 * powers_of_g[i] = tau^i * g with ark-ec FixedBase; any correct scalar multiplication yields the
 * same points).  out: n x 12 u64 affine Montgomery.  Scalars canonical.
 * ------------------------------------------------------------------------------------------ */
void go_g1_fixed_base_mul(const u64* base_aff12, const u64* scalars, size_t n, u64* out_aff) {
  /* 8-bit windows: table[w][d] = d * 2^(8w) * base */
  enum { W = 32, D = 256 };
  g1a* table = (g1a*)malloc(sizeof(g1a) * W * D);
  g1a b; memcpy(&b, base_aff12, sizeof b);
  g1j cur; cur.x = b.x; cur.y = b.y; cur.z = fq_one();
  if (g1a_is_inf(&b)) cur = g1j_zero();
  for (int w = 0; w < W; w++) {
    g1j acc = g1j_zero();
    g1a cura = g1j_to_affine(&cur);
    for (int d = 0; d < D; d++) {
      table[w * D + d] = g1j_to_affine(&acc);
      acc = g1j_add_mixed(&acc, &cura);
    }
    cur = acc; /* 256 * cur */
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (size_t i = 0; i < n; i++) {
    g1j acc = g1j_zero();
    for (int w = 0; w < W; w++) {
      unsigned d = (unsigned)((scalars[4 * i + w / 8] >> (8 * (w % 8))) & 0xff);
      if (d) acc = g1j_add_mixed(&acc, &table[w * D + d]);
    }
    g1a a = g1j_to_affine(&acc);
    memcpy(out_aff + 12 * i, &a, sizeof a);
  }
  free(table);
}
void go_g1_generator(u64* aff12) {
  static const u64 gx[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                            0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
  static const u64 gy[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                            0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
  go_fq_to_mont(gx, aff12, 1);
  go_fq_to_mont(gy, aff12 + 6, 1);
}

/* --------------------------------------------------------------------------------------------
 * Field vector helpers (src/misc.rs), Montgomery Fr arrays
 * ------------------------------------------------------------------------------------------ */
/* misc.rs:52-56; returns output length ceil(n/2) */
size_t go_fold_polynomial(const u64* f, size_t n, const u64* r, u64* out) {
  fr rr = fr_load(r);
  size_t m = (n + 1) / 2;
  for (size_t i = 0; i < m; i++) {
    fr e = fr_load(f + 4 * (2 * i));
    fr o = (2 * i + 1 < n) ? fr_load(f + 4 * (2 * i + 1)) : fr_zero();
    fr_store(out + 4 * i, fr_add(e, fr_mul(rr, o)));
  }
  return m;
}
/* misc.rs:59-65 */
void go_powers(const u64* x, size_t n, u64* out) {
  fr e = fr_load(x), cur = fr_one();
  for (size_t i = 0; i < n; i++) { fr_store(out + 4 * i, cur); cur = fr_mul(cur, e); }
}
/* misc.rs:68-77 */
void go_powers2(const u64* x, size_t n, u64* out) {
  fr cur = fr_load(x);
  for (size_t i = 0; i < n; i++) { fr_store(out + 4 * i, cur); cur = fr_sqr(cur); }
}
/* misc.rs:133-149; out has 2^k elements */
void go_tensor(const u64* elements, size_t k, u64* out) {
  fr_store(out, fr_one());
  for (size_t i = 0; i < k; i++) {
    fr e = fr_load(elements + 4 * i);
    for (size_t j = 0; j < ((size_t)1 << i); j++)
      fr_store(out + 4 * (((size_t)1 << i) + j), fr_mul(fr_load(out + 4 * j), e));
  }
}
/* misc.rs:180-199: evaluate_le by Horner from the top */
void go_evaluate_le(const u64* poly, size_t n, const u64* x, u64* out) {
  fr e = fr_load(x), acc = fr_zero();
  for (size_t i = n; i-- > 0;) acc = fr_add(fr_mul(acc, e), fr_load(poly + 4 * i));
  fr_store(out, acc);
}
/* misc.rs:205-208 */
void go_hadamard(const u64* a, const u64* b, size_t n, u64* out) {
  for (size_t i = 0; i < n; i++) fr_store(out + 4 * i, fr_mul(fr_load(a + 4 * i), fr_load(b + 4 * i)));
}
/* misc.rs:215-218 */
void go_ip(const u64* a, const u64* b, size_t n, u64* out) {
  fr acc = fr_zero();
  for (size_t i = 0; i < n; i++) acc = fr_add(acc, fr_mul(fr_load(a + 4 * i), fr_load(b + 4 * i)));
  fr_store(out, acc);
}
/* misc.rs:37-48: out[i] = sum_j c_j p_j[i]; polys given as (ptr,len) arrays; returns the length
 * after stripping high zero coefficients (DensePolynomial::from_coefficients_vec). */
size_t go_linear_combination(const u64* const* polys, const size_t* lens, size_t k, const u64* challenges, u64* out, size_t out_cap) {
  size_t n = 0;
  for (size_t j = 0; j < k; j++) if (lens[j] > n) n = lens[j];
  if (n > out_cap) return (size_t)-1;
  for (size_t i = 0; i < n; i++) fr_store(out + 4 * i, fr_zero());
  for (size_t j = 0; j < k; j++) {
    fr c = fr_load(challenges + 4 * j);
    for (size_t i = 0; i < lens[j]; i++)
      fr_store(out + 4 * i, fr_add(fr_load(out + 4 * i), fr_mul(fr_load(polys[j] + 4 * i), c)));
  }
  while (n > 0 && fr_is_zero(fr_load(out + 4 * (n - 1)))) n--;
  return n;
}
/* quotient of f (len n, little-endian) by the monic vanishing polynomial z (len d+1):
 * what `DensePolynomial::div` yields in src/kzg/time.rs:134-145.  q has n-d entries, rem d. */
void go_poly_div_monic(const u64* f, size_t n, const u64* z, size_t d, u64* q, u64* rem) {
  if (n <= d) { for (size_t i = 0; i < n; i++) fr_store(rem + 4 * i, fr_load(f + 4 * i)); for (size_t i = n; i < d; i++) fr_store(rem + 4 * i, fr_zero()); return; }
  fr* w = (fr*)malloc(sizeof(fr) * n);
  for (size_t i = 0; i < n; i++) w[i] = fr_load(f + 4 * i);
  for (size_t i = n; i-- > d;) {
    fr c = w[i];
    fr_store(q + 4 * (i - d), c);
    for (size_t j = 0; j < d; j++) w[i - d + j] = fr_sub(w[i - d + j], fr_mul(c, fr_load(z + 4 * j)));
  }
  for (size_t i = 0; i < d; i++) fr_store(rem + 4 * i, w[i]);
  free(w);
}

/* --------------------------------------------------------------------------------------------
 * Sumcheck time prover (src/subprotocols/sumcheck/time_prover.rs)
 * ------------------------------------------------------------------------------------------ */
/* :83-123 message part: a = sum f_e g_e t, b = sum (f_e g_o + g_e f_o twist) t, t *= twist^2.
 * Iterates min(ceil(nf/2), ceil(ng/2)) pairs (chunks(2).zip). */
void go_sumcheck_message(const u64* f, size_t nf, const u64* g, size_t ng, const u64* twist, u64* a_out, u64* b_out) {
  fr tw = fr_load(twist), tw2 = fr_sqr(tw), runner = fr_one();
  fr a = fr_zero(), b = fr_zero();
  size_t pf = (nf + 1) / 2, pg = (ng + 1) / 2, np = pf < pg ? pf : pg;
  for (size_t i = 0; i < np; i++) {
    fr fe = fr_load(f + 4 * (2 * i)), ge = fr_load(g + 4 * (2 * i));
    fr fo = (2 * i + 1 < nf) ? fr_load(f + 4 * (2 * i + 1)) : fr_zero();
    fr go = (2 * i + 1 < ng) ? fr_load(g + 4 * (2 * i + 1)) : fr_zero();
    a = fr_add(a, fr_mul(fr_mul(fe, ge), runner));
    fr t = fr_add(fr_mul(fe, go), fr_mul(fr_mul(ge, fo), tw));
    b = fr_add(b, fr_mul(t, runner));
    runner = fr_mul(runner, tw2);
  }
  fr_store(a_out, a);
  fr_store(b_out, b);
}
/* :75-80 fold in place: f <- fold(f, r*twist), g <- fold(g, r), twist <- twist^2.
 * Buffers are overwritten; new lengths returned through *nf, *ng. */
void go_sumcheck_fold(u64* f, size_t* nf, u64* g, size_t* ng, u64* twist, const u64* r) {
  fr rr = fr_load(r), tw = fr_load(twist);
  fr rt = fr_mul(rr, tw);
  *nf = go_fold_polynomial(f, *nf, rt.l, f);
  *ng = go_fold_polynomial(g, *ng, rr.l, g);
  fr_store(twist, fr_sqr(tw));
}
size_t go_sumcheck_rounds(size_t nf, size_t ng) { return ceil_log2(nf > ng ? nf : ng); }

/* --------------------------------------------------------------------------------------------
 * SplitMix64-based deterministic inputs shared with tests/bench (see pyref.SplitMix64)
 * ------------------------------------------------------------------------------------------ */
static u64 splitmix_next(u64* s) {
  *s += 0x9E3779B97F4A7C15ULL;
  u64 z = *s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* n canonical Fr scalars, rejection sampled from 255-bit draws */
void go_random_fr(u64 seed, size_t n, u64* out) {
  u64 s = seed;
  for (size_t i = 0; i < n; i++) {
    for (;;) {
      u64 v[4];
      for (int k = 0; k < 4; k++) v[k] = splitmix_next(&s);
      v[3] &= 0x7fffffffffffffffULL;
      if (!limbs_geq(v, R_MOD, NR)) { memcpy(out + 4 * i, v, 32); break; }
    }
  }
}

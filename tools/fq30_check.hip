// Unit checks of the Fq element layer of gemini_amd/csrc/g1.cuh against the host field arithmetic
// (gemini_amd/csrc/host_field.hpp), for either representation:
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -DGM_FQ30=1 -I gemini_amd/csrc tools/fq30_check.hip -o /tmp/fq30_check && /tmp/fq30_check
// Field level: mul / add / sub<K> on canonical and on loose operands, is_zero_mod on multiples of q,
// import / export.  Group level: xyzz_madd / xyzz_add / xyzz_dbl chains incl. P + P, P - P, identity
// operands, stored through g1_store_xyzz and read back with the host's xyzz_to_jac_dev.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "g1.cuh"
#include "host_field.hpp"

using namespace gm;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

constexpr int NF = 14;  // field results per test vector
__global__ void k_field(const uint8_t* a_in, const uint8_t* b_in, int n, uint8_t* out, uint32_t* flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FqE A = fqe_import(fp_load<FqParams>(a_in + i * 48)), B = fqe_import(fp_load<FqParams>(b_in + i * 48));
  FqE r[NF];
  r[0] = fq_mul(A, B);
  r[1] = fq_add(A, B);
  r[2] = fq_sub<1>(A, B);
  FqE L2 = fq_add(A, B);   // < 2q
  FqE L4 = fq_dbl(L2);     // < 4q
  FqE L8 = fq_dbl(L4);     // < 8q
  r[3] = fq_sub<2>(A, L2);   // -b
  r[4] = fq_sub<4>(A, L4);   // a - 2a - 2b
  r[5] = fq_sub<8>(A, L8);
  r[6] = fq_mul(L8, L4);     // 32 (a+b)^2
  r[7] = fq_sqr(fq_sub<8>(L2, L8));  // (a+b-4a-4b)^2 = 9(a+b)^2, operand < 10q
  r[8] = fq_mul(fq_sub<4>(fq_sub<2>(fq_sqr(L4), r[0]), fq_dbl(r[0])), fq_sub<8>(r[0], L8));  // the x3 / y3 shape
  r[9] = fq_neg_canonical(fqe_load(a_in + i * 48));  // memory image read as device form
  r[10] = fqe_load(a_in + i * 48);
  r[11] = fq_mul(fqe_one(), A);
  r[12] = fq_dbl(fq_dbl(fq_dbl(A)));  // 8a
  r[13] = fq_sub<8>(fq_mul(A, B), r[12]);
  for (int k = 0; k < NF; k++) fp_store<FqParams>(out + ((size_t)i * NF + k) * 48, fqe_export(r[k]));
  uint32_t f = 0;
  f |= fq_is_zero_mod(fq_sub<1>(A, A)) ? 1u : 0u;
  f |= fq_is_zero_mod(fq_sub<2>(L2, L2)) ? 2u : 0u;
  f |= fq_is_zero_mod(fq_sub<4>(L4, L4)) ? 4u : 0u;
  f |= fq_is_zero_mod(fq_sub<8>(L8, L8)) ? 8u : 0u;
  f |= fq_is_zero_mod(fq_sub<8>(fq_add(L8, A), L8)) ? 0u : 16u;  // = a != 0
  f |= fq_is_zero_mod(fq_sub<4>(fq_mul(A, B), fq_dbl(fq_mul(B, A)))) ? 0u : 32u;  // -ab != 0
  f |= fq_is_zero_mod(fq_sub<2>(fq_mul(A, B), fq_mul(B, A))) ? 64u : 0u;
  f |= fq_is_zero_mod(fqe_zero()) ? 128u : 0u;
  flags[i] = f;
}

// group checks: per vector, points P, Q (affine, ark form); results in XYZZ device memory form
constexpr int NG = 8;
__global__ void k_group(const uint8_t* p_in, const uint8_t* q_in, int n, uint8_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine P, Q;
  P.x = fqe_import(fp_load<FqParams>(p_in + i * 96));
  P.y = fqe_import(fp_load<FqParams>(p_in + i * 96 + 48));
  Q.x = fqe_import(fp_load<FqParams>(q_in + i * 96));
  Q.y = fqe_import(fp_load<FqParams>(q_in + i * 96 + 48));
  // round-trip the operands through memory like the MSM does (canonical device form)
  __shared__ uint8_t scratch[64 * 192];
  uint8_t* mine = scratch + (threadIdx.x % 64) * 192;
  fqe_store(mine, P.x);
  fqe_store(mine + 48, P.y);
  fqe_store(mine + 96, Q.x);
  fqe_store(mine + 144, Q.y);
  P = g1_load_affine(mine);
  Q = g1_load_affine(mine + 96);
  G1Affine nQ = Q;
  nQ.y = fq_neg_canonical(Q.y);
  G1Xyzz r[NG];
  r[0] = G1Xyzz::from_affine(P);
  xyzz_madd(r[0], Q);  // P + Q
  r[1] = G1Xyzz::from_affine(P);
  xyzz_madd(r[1], P);  // 2P
  r[2] = G1Xyzz::from_affine(Q);
  xyzz_madd(r[2], nQ);  // identity
  r[3] = G1Xyzz::identity();
  for (int k = 0; k < 9; k++) xyzz_madd(r[3], (k & 1) ? Q : P);  // 5P + 4Q: bound growth over a chain
  r[4] = r[3];
  xyzz_add(r[4], r[0]);  // 6P + 5Q
  r[5] = xyzz_dbl(r[3]);  // 10P + 8Q
  r[6] = r[3];
  xyzz_add(r[6], r[3]);  // same by the addition path (equal operands)
  r[7] = r[0];
  for (int k = 0; k < 3; k++) xyzz_madd(r[7], nQ);  // P - 2Q
  for (int k = 0; k < NG; k++) g1_store_xyzz(out + ((size_t)i * NG + k) * 192, r[k]);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static gmh::Fq rand_fq(int kind) {
  uint64_t l[6];
  for (int k = 0; k < 6; k++) l[k] = rnd();
  l[5] &= 0x0fffffffffffffffull;  // < 2^380 < q
  gmh::Fq v = gmh::Fq::from_canonical(l);
  if (kind == 1) return gmh::Fq::zero();
  if (kind == 2) return gmh::Fq::one();
  if (kind == 3) return gmh::Fq::zero() - gmh::Fq::one();  // q - 1
  return v;
}

int main() {
  const int n = 2048;
  std::vector<uint64_t> a(n * 6), b(n * 6);
  std::vector<gmh::Fq> A(n), B(n);
  for (int i = 0; i < n; i++) {
    A[i] = rand_fq(i < 16 ? i % 4 : 0);
    B[i] = rand_fq(i < 16 ? (i / 4) % 4 : 0);
    A[i].to_limbs(&a[i * 6]);
    B[i].to_limbs(&b[i * 6]);
  }
  uint8_t *da, *db, *dout;
  uint32_t* dfl;
  CK(hipMalloc(&da, n * 48));
  CK(hipMalloc(&db, n * 48));
  CK(hipMalloc(&dout, (size_t)n * NF * 48));
  CK(hipMalloc(&dfl, n * 4));
  CK(hipMemcpy(da, a.data(), n * 48, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), n * 48, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_field, dim3((n + 63) / 64), dim3(64), 0, 0, da, db, n, dout, dfl);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> out((size_t)n * NF * 6);
  std::vector<uint32_t> fl(n);
  CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(fl.data(), dfl, n * 4, hipMemcpyDeviceToHost));
  int bad[NF] = {0}, badflags = 0;
  // the memory image read directly as device form means value * 2^-6 (GM_FQ30) or the value itself
  for (int i = 0; i < n; i++) {
    gmh::Fq x = A[i], y = B[i];
    gmh::Fq two = gmh::Fq::one() + gmh::Fq::one();
    gmh::Fq s = x + y;
    gmh::Fq e[NF];
    e[0] = x * y;
    e[1] = s;
    e[2] = x - y;
    e[3] = x - s;
    e[4] = x - (s + s);
    e[5] = x - (s + s + s + s);
    gmh::Fq s4 = s + s, s8 = s4 + s4;  // L4 = 2s (< 4q), L8 = 4s (< 8q)
    e[6] = s8 * s4;
    e[7] = (s - s8) * (s - s8);
    e[8] = (s4 * s4 - e[0] - (e[0] + e[0])) * (e[0] - s8);
    gmh::Fq raw = gmh::fq_from_device(&a[i * 6]);  // how the library reads a device-form record
    e[9] = raw.neg();
    e[10] = raw;
    e[11] = x;
    e[12] = (x + x + x + x) + (x + x + x + x);
    e[13] = x * y - e[12];
    (void)two;
    for (int k = 0; k < NF; k++) {
      uint64_t w[6];
      e[k].to_limbs(w);
      if (memcmp(w, &out[((size_t)i * NF + k) * 6], 48) != 0) {
        if (bad[k]++ < 2) printf("field mismatch test %d vector %d\n", k, i);
      }
    }
    uint32_t want = 1 | 2 | 4 | 8 | 64 | 128;
    if (!x.is_zero()) want |= 16;
    if (!(x * y).is_zero()) want |= 32;
    if (fl[i] != want) {
      if (badflags++ < 4) printf("flag mismatch vector %d: got %x want %x\n", i, fl[i], want);
    }
  }
  int total_bad = badflags;
  for (int k = 0; k < NF; k++) {
    total_bad += bad[k];
    if (bad[k]) printf("field test %d: %d / %d mismatches\n", k, bad[k], n);
  }
  printf("field: %s (GM_FQ30=%d)\n", total_bad ? "FAIL" : "ok", (int)GM_FQ30);

  // ---- group ----
  const int m = 512;
  gmh::G1 G = gmh::G1::identity();
  {
    // generator in Montgomery form
    const uint64_t gx[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
    const uint64_t gy[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
    G.x = gmh::Fq::from_limbs(gx);
    G.y = gmh::Fq::from_limbs(gy);
    G.z = gmh::Fq::one();
  }
  std::vector<gmh::G1> P(m), Q(m);
  std::vector<uint64_t> pin(m * 12), qin(m * 12);
  gmh::G1 cur = G, cur2 = G.dbl().add(G);
  for (int i = 0; i < m; i++) {
    cur = cur.dbl().add(G);
    cur2 = cur2.add(cur).dbl();
    P[i] = cur.normalized();
    Q[i] = (i % 7 == 3) ? P[i] : cur2.normalized();  // some P == Q vectors
    P[i].x.to_limbs(&pin[i * 12]);
    P[i].y.to_limbs(&pin[i * 12 + 6]);
    Q[i].x.to_limbs(&qin[i * 12]);
    Q[i].y.to_limbs(&qin[i * 12 + 6]);
  }
  uint8_t *dp, *dq, *dg;
  CK(hipMalloc(&dp, m * 96));
  CK(hipMalloc(&dq, m * 96));
  CK(hipMalloc(&dg, (size_t)m * NG * 192));
  CK(hipMemcpy(dp, pin.data(), m * 96, hipMemcpyHostToDevice));
  CK(hipMemcpy(dq, qin.data(), m * 96, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_group, dim3((m + 63) / 64), dim3(64), 0, 0, dp, dq, m, dg);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> gout((size_t)m * NG * 24);
  CK(hipMemcpy(gout.data(), dg, gout.size() * 8, hipMemcpyDeviceToHost));
  int gbad[NG] = {0}, gtot = 0;
  for (int i = 0; i < m; i++) {
    gmh::G1 e[NG];
    gmh::G1 nQ = Q[i];
    nQ.y = nQ.y.neg();
    e[0] = P[i].add(Q[i]);
    e[1] = P[i].dbl();
    e[2] = gmh::G1::identity();
    gmh::G1 c = gmh::G1::identity();
    for (int k = 0; k < 9; k++) c = c.add((k & 1) ? Q[i] : P[i]);
    e[3] = c;
    e[4] = c.add(e[0]);
    e[5] = c.dbl();
    e[6] = c.add(c);
    e[7] = e[0].add(nQ).add(nQ).add(nQ);
    for (int k = 0; k < NG; k++) {
      gmh::G1 got = gmh::xyzz_to_jac_dev(&gout[((size_t)i * NG + k) * 24]).normalized();
      gmh::G1 want = e[k].normalized();
      uint64_t wa[18], wb[18];
      got.to_limbs(wa);
      want.to_limbs(wb);
      if (memcmp(wa, wb, 144) != 0) {
        if (gbad[k]++ < 2) printf("group mismatch test %d vector %d (P==Q: %d)\n", k, i, (int)(i % 7 == 3));
      }
    }
  }
  for (int k = 0; k < NG; k++) {
    gtot += gbad[k];
    if (gbad[k]) printf("group test %d: %d / %d mismatches\n", k, gbad[k], m);
  }
  printf("group: %s\n", gtot ? "FAIL" : "ok");
  return (total_bad || gtot) ? 1 : 0;
}

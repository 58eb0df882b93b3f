// Device check + throughput of the radix-2^30 asm Fq multiplier / squarer (gen_field_mul30.py) against the
// 32-bit-limb product-scanning multiplier (gen_field_mul.py).  Dev tool, not product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fqmul_check.hip -o tools/_build/fqmul_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../gemini_amd/csrc/field.cuh"
namespace gm {
#include "../gemini_amd/csrc/field_mul30_gen.inc"
}
using namespace gm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ Fq mul30(const Fq& a, const Fq& b) {
  return fq30h_mul_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1], b.l[2], b.l[3],
                      b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
}
__device__ Fq sqr30(const Fq& a) {
  return fq30h_sqr_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11]);
}
// the 32-bit multiplier out of line with the same calling sequence, for a like-for-like rate
__device__ __noinline__ Fq mul32_fn(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8,
                                    uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5,
                                    uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
  Fq a, b;
  a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7; a.l[8] = a8; a.l[9] = a9; a.l[10] = a10; a.l[11] = a11;
  b.l[0] = b0; b.l[1] = b1; b.l[2] = b2; b.l[3] = b3; b.l[4] = b4; b.l[5] = b5; b.l[6] = b6; b.l[7] = b7; b.l[8] = b8; b.l[9] = b9; b.l[10] = b10; b.l[11] = b11;
  return fp_mul<FqParams>(a, b);
}
__device__ Fq mul32(const Fq& a, const Fq& b) {
  return mul32_fn(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4],
                  b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
}

// a*b*2^-390 == (a*b*2^-384) * 2^378 * 2^-384
__global__ void k_check(const uint32_t* a, const uint32_t* b, uint32_t* ref_mul, uint32_t* new_mul, uint32_t* ref_sqr, uint32_t* new_sqr) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  Fq c378 = Fq::zero();
  c378.l[11] = 0x4000000u;
  fp_store<FqParams>(ref_mul + 12 * t, fp_mul<FqParams>(fp_mul<FqParams>(x, y), c378));
  fp_store<FqParams>(new_mul + 12 * t, mul30(x, y));
  fp_store<FqParams>(ref_sqr + 12 * t, fp_mul<FqParams>(fp_mul<FqParams>(x, x), c378));
  fp_store<FqParams>(new_sqr + 12 * t, sqr30(x));
}
template <int WHICH>
__global__ void k_rate(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  for (int i = 0; i < iters; i++) {
    if (WHICH == 0) x = mul32(x, y);
    if (WHICH == 1) x = mul30(x, y);
    if (WHICH == 2) x = sqr30(x);
    if (WHICH == 3) x = fp_mul<FqParams>(x, y);
  }
  fp_store<FqParams>(out + 12 * t, x);
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main(int argc, char** argv) {
  const uint32_t Qw[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  int bad = 0;
  for (int wps : {8, 4, 3, 2, 1}) {
    const int blocks = 256 * wps, threads = 256, n = blocks * threads;
    std::vector<uint32_t> ha(12 * n), hb(12 * n);
    uint64_t s = 88172645463325252ull;
    auto lt_q = [&](const uint32_t* v) { for (int i = 11; i >= 0; i--) { if (v[i] != Qw[i]) return v[i] < Qw[i]; } return false; };
    for (auto* v : {&ha, &hb})
      for (int e = 0; e < n; e++) {
        uint32_t* p = v->data() + 12 * e;
        do {
          for (int i = 0; i < 12; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; p[i] = (uint32_t)s; }
          p[11] &= 0x1fffffffu;
        } while (!lt_q(p));
      }
    // edge values in the first lanes: 0, 1, q - 1, all-ones 30-bit limbs below q
    for (int i = 0; i < 12; i++) { ha[i] = 0; ha[12 + i] = i == 0; ha[24 + i] = Qw[i] - (i == 0); hb[24 + i] = Qw[i] - (i == 0); ha[36 + i] = hb[36 + i] = i < 11 ? 0xffffffffu : 0x19ffffffu; }
    uint32_t *da, *db, *o[4];
    CK(hipMalloc(&da, 48 * n)); CK(hipMalloc(&db, 48 * n));
    for (auto& p : o) CK(hipMalloc(&p, 48 * n));
    CK(hipMemcpy(da, ha.data(), 48 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 48 * n, hipMemcpyHostToDevice));
    k_check<<<blocks, threads>>>(da, db, o[0], o[1], o[2], o[3]);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h[4];
    for (int i = 0; i < 4; i++) { h[i].resize(12 * n); CK(hipMemcpy(h[i].data(), o[i], 48 * n, hipMemcpyDeviceToHost)); }
    printf("waves/SIMD %d: mul30 == mul32 * 2^-6: %s   sqr30 == sqr32 * 2^-6: %s   (%d products each)\n", wps, h[0] == h[1] ? "yes" : "NO", h[2] == h[3] ? "yes" : "NO", n);
    bad += h[0] != h[1];
    bad += h[2] != h[3];
    const int fi = 200;
    float ms;
    ms = timeit([&] { k_rate<3><<<blocks, threads>>>(da, db, o[0], fi); });
    printf("  fq_mul 12x32 inline    %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    ms = timeit([&] { k_rate<0><<<blocks, threads>>>(da, db, o[0], fi); });
    printf("  fq_mul 12x32 call      %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    ms = timeit([&] { k_rate<1><<<blocks, threads>>>(da, db, o[0], fi); });
    printf("  fq_mul radix-2^30 asm  %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    ms = timeit([&] { k_rate<2><<<blocks, threads>>>(da, db, o[0], fi); });
    printf("  fq_sqr radix-2^30 asm  %8.3f ms  %8.2f Gsqr/s\n", ms, (double)n * fi / ms / 1e6);
    CK(hipFree(da)); CK(hipFree(db));
    for (auto& p : o) CK(hipFree(p));
  }
  return bad ? 1 : 0;
}

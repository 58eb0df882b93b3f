// Instruction-rate microbenchmarks for the integer pipeline of gfx950 (dev tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_isa.hip -o tools/ubench_isa
// Each asm statement holds 8 independent instructions so hipcc's one-state pad after
// ;;#ASMEND is amortised (a lone instruction per statement measures instr + s_nop).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../gemini_amd/csrc/field.cuh"
#include "../gemini_amd/csrc/field30.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define ACC8 "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
#define I8S(pre, post, suf) pre "%0" post "%0" suf "\n\t" pre "%1" post "%1" suf "\n\t" pre "%2" post "%2" suf "\n\t" pre "%3" post "%3" suf "\n\t" pre "%4" post "%4" suf "\n\t" pre "%5" post "%5" suf "\n\t" pre "%6" post "%6" suf "\n\t" pre "%7" post "%7" suf
#define I8(pre, post) I8S(pre, post, "")

#define KERNEL(NAME, TYPE, INIT, ASM8, XT, XV, YV)                               \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {        \
    TYPE acc[8];                                                                 \
    XT x = XV, y = YV;                                                           \
    for (int i = 0; i < 8; i++) acc[i] = INIT;                                   \
    for (int it = 0; it < iters; it++) {                                         \
      asm volatile(ASM8 : ACC8 : "v"(x), "v"(y) : "vcc");                        \
      asm volatile(ASM8 : ACC8 : "v"(x), "v"(y) : "vcc");                        \
      asm volatile(ASM8 : ACC8 : "v"(x), "v"(y) : "vcc");                        \
      asm volatile(ASM8 : ACC8 : "v"(x), "v"(y) : "vcc");                        \
    }                                                                            \
    uint64_t s = 0;                                                              \
    for (int i = 0; i < 8; i++) s ^= (uint64_t)acc[i];                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
  }

KERNEL(k_mad64, uint64_t, (i + threadIdx.x), I8("v_mad_u64_u32 ", ", vcc, %8, %9, "), uint32_t, a + threadIdx.x, b)
KERNEL(k_mullo, uint32_t, (a + i + threadIdx.x), I8("v_mul_lo_u32 ", ", %9, "), uint32_t, a, b | 1)
KERNEL(k_mulhi, uint32_t, (a + i + threadIdx.x), I8("v_mul_hi_u32 ", ", %9, "), uint32_t, a, b | 0x80000001u)
KERNEL(k_addc, uint32_t, (a + i + threadIdx.x), I8S("v_addc_co_u32 ", ", vcc, %9, ", ", vcc"), uint32_t, a, b)
KERNEL(k_add, uint32_t, (a + i + threadIdx.x), I8("v_add_u32 ", ", %9, "), uint32_t, a, b)
KERNEL(k_lshladd64, uint64_t, (a + i + threadIdx.x), I8("v_lshl_add_u64 ", ", %9, 1, "), uint64_t, a, b)
KERNEL(k_fma64, double, (double)(i + threadIdx.x), I8("v_fma_f64 ", ", %8, %9, "), double, 1.0 + 1e-9 * a, 1e-9 * b)
KERNEL(k_fma32, float, (float)(i + threadIdx.x), I8("v_fma_f32 ", ", %8, %9, "), float, 1.0f + 1e-9f * a, 1e-9f * b)
KERNEL(k_mad24, uint32_t, (a + i + threadIdx.x), I8("v_mad_u32_u24 ", ", %9, %8, "), uint32_t, a, b)

// mad + addc pair on a 96-bit accumulator (the Montgomery inner step)
__global__ void k_mad64_addc(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[4]; uint32_t hi[4];
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < 4; i++) { acc[i] = i + threadIdx.x; hi[i] = 0; }
#define P(k, h) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n\tv_addc_co_u32 %" #h ", vcc, 0, %" #h ", vcc\n\t"
  for (int it = 0; it < iters; it++) {
    asm volatile(P(0, 4) P(1, 5) P(2, 6) P(3, 7) P(0, 4) P(1, 5) P(2, 6) P(3, 7) "s_nop 0"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]) : "v"(x), "v"(y) : "vcc");
    asm volatile(P(0, 4) P(1, 5) P(2, 6) P(3, 7) P(0, 4) P(1, 5) P(2, 6) P(3, 7) "s_nop 0"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]) : "v"(x), "v"(y) : "vcc");
  }
  uint64_t s = 0; for (int i = 0; i < 4; i++) s ^= acc[i] + hi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same, single dependent chain (what one column of a Montgomery product looks like)
__global__ void k_mad64_addc_dep(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t hi = 0;
  uint32_t x = a + threadIdx.x, y = b;
#define Q "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
  for (int it = 0; it < iters; it++) {
    asm volatile(Q Q Q Q Q Q Q Q "s_nop 0" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc");
    asm volatile(Q Q Q Q Q Q Q Q "s_nop 0" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hi;
}

using namespace gm;
__global__ void k_fq_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  for (int i = 0; i < iters; i++) { x = fp_mul<FqParams>(x, y); }
  fp_store<FqParams>(out + 12 * t, x);
}
__global__ void k_fq_mul_cios(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  for (int i = 0; i < iters; i++) { x = fp_mul_cios<FqParams>(x, y); }
  fp_store<FqParams>(out + 12 * t, x);
}
__global__ void k_fr_mul_cios(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = fp_load<FrParams>(a + 8 * t), y = fp_load<FrParams>(b + 8 * t);
  for (int i = 0; i < iters; i++) { x = fp_mul_cios<FrParams>(x, y); }
  fp_store<FrParams>(out + 8 * t, x);
}
__global__ void k_fq30_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq30 x = fq30_unpack(fp_load<FqParams>(a + 12 * t)), y = fq30_unpack(fp_load<FqParams>(b + 12 * t));
  for (int i = 0; i < iters; i++) { x = fq30_mul(x, y); }
  fp_store<FqParams>(out + 12 * t, fq30_pack(x));
}
// one product both ways: out32 = a*b*2^-384 (canonical), out30 = (a*b*2^-390) * 2^6 canonicalised
__global__ void k_fq30_check(const uint32_t* a, const uint32_t* b, uint32_t* out32, uint32_t* out30) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  fp_cond_sub<FqParams>(x, 0); fp_cond_sub<FqParams>(y, 0);
  fp_store<FqParams>(out32 + 12 * t, fp_mul<FqParams>(x, y));
  Fq30 r = fq30_mul(fq30_unpack(x), fq30_unpack(y));
  const Fq30 c396 = {{0x3480cb7fu, 0x3e0c0000u, 0x2042b126u, 0x3f337aafu, 0x3de4b4d1u, 0x1e015cf1u, 0x005c540du, 0x3467b19au, 0x352a6da3u, 0x19d89d19u, 0x2fb9afe6u, 0x3848c817u, 0x0009772fu}};
  r = fq30_mul(r, c396);
  Fq p = fq30_pack(r);
  fp_cond_sub<FqParams>(p, 0); fp_cond_sub<FqParams>(p, 0);
  fp_store<FqParams>(out30 + 12 * t, p);
}
__global__ void k_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = fp_load<FrParams>(a + 8 * t), y = fp_load<FrParams>(b + 8 * t);
  for (int i = 0; i < iters; i++) { x = fp_mul<FrParams>(x, y); }
  fp_store<FrParams>(out + 8 * t, x);
}
__global__ void k_fq_add(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x = fp_load<FqParams>(a + 12 * t), y = fp_load<FqParams>(b + 12 * t);
  for (int i = 0; i < iters; i++) { x = fp_add<FqParams>(x, y); y = fp_sub<FqParams>(y, x); }
  fp_store<FqParams>(out + 12 * t, x);
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main(int argc, char** argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 8;  // waves per SIMD
  const int blocks = 256 * wps, threads = 256, iters = 2000;
  uint64_t* out; CK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
  printf("waves/SIMD = %d\n", wps);
#define RUN(name, k, per_iter) { double nops = (double)blocks * threads * iters * per_iter; float ms = timeit([&] { k<<<blocks, threads>>>(out, 12345u, 67891u, iters); }); \
    printf("%-18s %8.3f ms  %8.2f Gop/s/lane  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", name, ms, nops / ms / 1e6, 1024.0 * 2.4e9 / (nops / 64 / (ms * 1e-3))); }
  RUN("add_u32", k_add, 32)
  RUN("addc_co_u32", k_addc, 32)
  RUN("fma_f32", k_fma32, 32)
  RUN("mad_u32_u24", k_mad24, 32)
  RUN("lshl_add_u64", k_lshladd64, 32)
  RUN("fma_f64", k_fma64, 32)
  RUN("mul_lo_u32", k_mullo, 32)
  RUN("mul_hi_u32", k_mulhi, 32)
  RUN("mad_u64_u32", k_mad64, 32)
  RUN("mad64+addc x4chain", k_mad64_addc, 32)
  RUN("mad64+addc 1chain", k_mad64_addc_dep, 32)

  {
    int n = blocks * threads;
    std::vector<uint32_t> ha(12 * n), hb(12 * n);
    uint64_t s = 88172645463325252ull;
    for (auto* v : {&ha, &hb}) for (size_t i = 0; i < v->size(); i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; (*v)[i] = (uint32_t)s; if (i % 12 == 11) (*v)[i] &= 0x0fffffffu; }
    uint32_t *da, *db, *dc; CK(hipMalloc(&da, 48 * n)); CK(hipMalloc(&db, 48 * n)); CK(hipMalloc(&dc, 48 * n));
    CK(hipMemcpy(da, ha.data(), 48 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 48 * n, hipMemcpyHostToDevice));
    int fi = 200;
    uint32_t* dd; CK(hipMalloc(&dd, 48 * n));
    std::vector<uint32_t> h1(12 * n), h2(12 * n);
    float ms = timeit([&] { k_fq_mul<<<blocks, threads>>>(da, db, dc, fi); });
    printf("fq_mul (asm FIPS) %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    ms = timeit([&] { k_fq_mul_cios<<<blocks, threads>>>(da, db, dd, fi); });
    printf("fq_mul (C CIOS)   %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    CK(hipMemcpy(h1.data(), dc, 48 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dd, 48 * n, hipMemcpyDeviceToHost));
    printf("fq asm == cios: %s\n", h1 == h2 ? "yes" : "NO");
    ms = timeit([&] { k_fq30_mul<<<blocks, threads>>>(da, db, dc, fi); });
    printf("fq30_mul (13x30)  %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    k_fq30_check<<<blocks, threads>>>(da, db, dc, dd);
    CK(hipMemcpy(h1.data(), dc, 48 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dd, 48 * n, hipMemcpyDeviceToHost));
    printf("fq30 * 2^6 == fq32: %s\n", h1 == h2 ? "yes" : "NO");
    ms = timeit([&] { k_fr_mul<<<blocks, threads>>>(da, db, dc, fi); });
    printf("fr_mul (asm FIPS) %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    ms = timeit([&] { k_fr_mul_cios<<<blocks, threads>>>(da, db, dd, fi); });
    printf("fr_mul (C CIOS)   %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * fi / ms / 1e6);
    CK(hipMemcpy(h1.data(), dc, 32 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dd, 32 * n, hipMemcpyDeviceToHost));
    printf("fr asm == cios: %s\n", std::equal(h1.begin(), h1.begin() + 8 * n, h2.begin()) ? "yes" : "NO");
    ms = timeit([&] { k_fq_add<<<blocks, threads>>>(da, db, dc, fi); });
    printf("fq_add+sub        %8.3f ms  %8.2f Gpair/s\n", ms, (double)n * fi / ms / 1e6);
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  return 0;
}

// Shader clock of the MI355X UNDER the load k_acc0 puts on it: every SIMD issues back-to-back v_mad_u64_u32 from two waves
// (the half-rate integer multiply-add that is 80 % of the mixed addition).  clock64() counts shader cycles, wall_clock64()
// ticks at a constant 100 MHz; their ratio over a ~3 ms kernel is the clock the issue bound of DESIGN section 4.1 has to
// be priced at (the bound there assumes the nominal 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_load(uint64_t* out, uint32_t iters, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t x = (uint32_t)a0 | 1u, y = (uint32_t)a1 | 3u;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      asm volatile(
          "v_mad_u64_u32 %0, s[60:61], %8, %9, %0\n\t"
          "v_mad_u64_u32 %1, s[60:61], %8, %9, %1\n\t"
          "v_mad_u64_u32 %2, s[60:61], %8, %9, %2\n\t"
          "v_mad_u64_u32 %3, s[60:61], %8, %9, %3\n\t"
          "v_mad_u64_u32 %4, s[60:61], %8, %9, %4\n\t"
          "v_mad_u64_u32 %5, s[60:61], %8, %9, %5\n\t"
          "v_mad_u64_u32 %6, s[60:61], %8, %9, %6\n\t"
          "v_mad_u64_u32 %7, s[60:61], %8, %9, %7"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(x), "v"(y)
          : "s60", "s61");
    }
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = c1 - c0;
    out[3 * blockIdx.x + 1] = w1 - w0;
    out[3 * blockIdx.x + 2] = s;
  }
}

// dependent chains: CH accumulators per wave, 64 multiply-adds per iteration in round-robin order.  CH = 1 is what a column sum of
// the Montgomery product looks like (every multiply-add waits for the one before it)
template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain(uint64_t* out, uint32_t iters, uint32_t seed) {
  uint64_t a[8];
  for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x * (2 * k + 1);
  uint32_t x = (uint32_t)a[0] | 1u, y = (uint32_t)a[1] | 3u;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 64; k++) {
      asm volatile("v_mad_u64_u32 %0, s[60:61], %1, %2, %0" : "+v"(a[k % CH]) : "v"(x), "v"(y) : "s60", "s61");
    }
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  uint64_t s = 0;
  for (int k = 0; k < 8; k++) s ^= a[k];
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = c1 - c0;
    out[3 * blockIdx.x + 1] = w1 - w0;
    out[3 * blockIdx.x + 2] = s;
  }
}

// the same inside ONE asm statement (no compiler-inserted s_nop between the instructions, as in the generated group law)
template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain1(uint64_t* out, uint32_t iters, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  uint32_t x = (uint32_t)a0 | 1u, y = (uint32_t)a1 | 3u;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (CH == 1)
        asm volatile(
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\t"
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "s60", "s61");
      else if (CH == 2)
        asm volatile(
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1\n\t"
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1\n\tv_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "s60", "s61");
      else
        asm volatile(
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1\n\tv_mad_u64_u32 %2, s[60:61], %4, %5, %2\n\tv_mad_u64_u32 %3, s[60:61], %4, %5, %3\n\t"
            "v_mad_u64_u32 %0, s[60:61], %4, %5, %0\n\tv_mad_u64_u32 %1, s[60:61], %4, %5, %1\n\tv_mad_u64_u32 %2, s[60:61], %4, %5, %2\n\tv_mad_u64_u32 %3, s[60:61], %4, %5, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "s60", "s61");
    }
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = c1 - c0;
    out[3 * blockIdx.x + 1] = w1 - w0;
    out[3 * blockIdx.x + 2] = a0 ^ a1 ^ a2 ^ a3;
  }
}
template <int CH>
static void run_chain1(uint64_t* d, int blocks) {
  std::vector<uint64_t> h(blocks * 3);
  const uint32_t iters = 6000;
  hipLaunchKernelGGL(k_chain1<CH>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int b = 0; b < blocks; b++) {
    cyc += (double)h[3 * b];
    wall += (double)h[3 * b + 1];
  }
  printf("one asm statement, %d chain(s) per wave, %d blocks: shader clock %.0f MHz, %.2f cycles per v_mad_u64_u32 per SIMD\n", CH, blocks,
         cyc / wall * 100.0, cyc / blocks / ((double)iters * 64.0 * (blocks >= 512 ? 2.0 : 1.0)));
}

template <int CH>
static void run_chain(uint64_t* d, int blocks) {
  std::vector<uint64_t> h(blocks * 3);
  const uint32_t iters = 6000;
  hipLaunchKernelGGL(k_chain<CH>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int b = 0; b < blocks; b++) {
    cyc += (double)h[3 * b];
    wall += (double)h[3 * b + 1];
  }
  printf("%d chain(s) per wave, %d blocks: shader clock %.0f MHz, %.2f cycles per v_mad_u64_u32 per SIMD\n", CH, blocks, cyc / wall * 100.0,
         cyc / blocks / ((double)iters * 64.0 * (blocks >= 512 ? 2.0 : 1.0)));
}

int main() {
  const int blocks = 512;  // 2 waves on each of the 1024 SIMDs
  uint64_t* d;
  (void)hipMalloc(&d, blocks * 24);
  std::vector<uint64_t> h(blocks * 3);
  for (int rep = 0; rep < 6; rep++) {
    const uint32_t iters = rep < 2 ? 2000 : 12000;  // ~64 mads per iteration and wave
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, 0, d, iters, (uint32_t)rep);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; b++) {
      cyc += (double)h[3 * b];
      wall += (double)h[3 * b + 1];
    }
    const double mhz = cyc / wall * 100.0;  // wall_clock64: 100 MHz
    const double mads = (double)iters * 64.0;
    printf("rep %d: kernel %.3f ms, shader clock %.0f MHz, %.2f cycles per v_mad_u64_u32 per SIMD (2 waves)\n", rep, ms, mhz,
           cyc / blocks / (mads * 2.0));
  }
  run_chain<1>(d, blocks);
  run_chain<2>(d, blocks);
  run_chain<4>(d, blocks);
  run_chain<8>(d, blocks);
  run_chain1<1>(d, blocks);
  run_chain1<2>(d, blocks);
  run_chain1<4>(d, blocks);
  run_chain1<1>(d, 256);
  run_chain1<4>(d, 256);
  run_chain<1>(d, 256);  // one wave per SIMD
  run_chain<2>(d, 256);
  run_chain<8>(d, 256);
  return 0;
}

// Real issue cycles of the generated group law (g1_madd30_asm / g1_add30_asm, gen_madd30.py) with operands in registers:
// clock64() around N statements, two waves per SIMD as in k_acc0, against the constant-rate wall clock for the MHz.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGM_FQ30=2 tools/madd_cycles.hip -o /tmp/madd_cycles && /tmp/madd_cycles
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <utility>
#include <vector>
#include "../gemini_amd/csrc/g1.cuh"
using namespace gm;
constexpr size_t AFF_BYTES = 96;

template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_madd(const uint8_t* base_pts, uint64_t* out, uint32_t iters) {
  Acc30 acc;
  acc30_set_identity(acc);
  const gm_u4v* bp = reinterpret_cast<const gm_u4v*>(base_pts + (size_t)(threadIdx.x & 63) * AFF_BYTES);
  const gm_u4v b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3], b4 = bp[4], b5 = bp[5];
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (uint32_t i = 0; i < iters; i++) {
    gm_u4v x0 = b0, x1 = b1, x2 = b2, y0 = b3, y1 = b4, y2 = b5;
    x0.x ^= i;  // not a curve point any more: the arithmetic does not care, the doubling test never fires
    (void)g1_madd30_asm(acc, x0, x1, x2, y0, y1, y2, i & 1u);
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x] = c1 - c0;
    out[4 * blockIdx.x + 1] = w1 - w0;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[4 * blockIdx.x + 2] = w0;
    out[4 * blockIdx.x + 3] = ((uint64_t)xcc << 32) | hw;
  }
  if (acc30_limb(acc, 0) == 0xdeadbeefu) out[4 * blockIdx.x + 1] = acc30_limb(acc, 1);
}

// the same with the base GATHERED per iteration from a table of `npts` points at a pseudo-random index (k_acc0's access pattern),
// loaded right in front of the statement (PF = 0) or one iteration ahead (PF = 1, asm loads + explicit wait as in k_acc0_pf)
template <int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_madd_gather(const uint8_t* table, uint32_t mask, uint64_t* out, uint32_t iters) {
  Acc30 acc;
  acc30_set_identity(acc);
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  const gm_u4v* bp = reinterpret_cast<const gm_u4v*>(table + (size_t)(idx & mask) * AFF_BYTES);
  gm_u4v x0 = bp[0], x1 = bp[1], x2 = bp[2], y0 = bp[3], y1 = bp[4], y2 = bp[5];
  for (uint32_t i = 0; i < iters; i++) {
    idx = idx * 1664525u + 1013904223u;
    const gm_u4v* np = reinterpret_cast<const gm_u4v*>(table + (size_t)((idx >> 4) & mask) * AFF_BYTES);
    gm_u4v n0, n1, n2, n3, n4, n5;
    if (PF) {
      asm volatile(
          "global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\tglobal_load_dwordx4 %2, %6, off offset:32\n\t"
          "global_load_dwordx4 %3, %6, off offset:48\n\tglobal_load_dwordx4 %4, %6, off offset:64\n\tglobal_load_dwordx4 %5, %6, off offset:80"
          : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5)
          : "v"(np));
    } else {
      x0 = np[0]; x1 = np[1]; x2 = np[2]; y0 = np[3]; y1 = np[4]; y2 = np[5];
    }
    (void)g1_madd30_asm(acc, x0, x1, x2, y0, y1, y2, i & 1u);
    if (PF) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5));
      x0 = n0; x1 = n1; x2 = n2; y0 = n3; y1 = n4; y2 = n5;
    }
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x] = c1 - c0;
    out[4 * blockIdx.x + 1] = w1 - w0;
  }
  if (acc30_limb(acc, 0) == 0xdeadbeefu) out[4 * blockIdx.x + 2] = acc30_limb(acc, 1);
}

template <int PF>
static void run_gather(const uint8_t* d_tab, uint32_t lognpts, uint64_t* d_o) {
  const int blocks = 512;
  const uint32_t iters = 300;
  (void)hipMemset(d_o, 0, 1024 * 32);
  hipLaunchKernelGGL(k_madd_gather<PF>, dim3(blocks), dim3(256), 0, 0, d_tab, (1u << lognpts) - 1u, d_o, iters);
  (void)hipDeviceSynchronize();
  std::vector<uint64_t> o(blocks * 4);
  (void)hipMemcpy(o.data(), d_o, blocks * 32, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int b = 0; b < blocks; b++) {
    cyc += (double)o[4 * b];
    wall += (double)o[4 * b + 1];
  }
  printf("gather from 2^%u points (%.0f MB), %s: %.0f cycles per iteration and wave (2 waves per SIMD), clock %.0f MHz\n", lognpts,
         96.0 * (1u << lognpts) / 1e6, PF ? "next base in flight during the addition" : "base loaded in front of the addition", cyc / blocks / iters,
         cyc / wall * 100.0);
}

int main() {
  std::vector<uint8_t> h(64 * AFF_BYTES);
  uint64_t st = 88172645463325252ull;
  for (auto& b : h) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    b = (uint8_t)st;
  }
  // keep the 12-word values below q: clear the top bits of the top word of x and y
  for (int i = 0; i < 64; i++) {
    h[i * AFF_BYTES + 47] &= 0x0f;
    h[i * AFF_BYTES + 95] &= 0x0f;
  }
  uint8_t* d_b;
  uint64_t* d_o;
  (void)hipMalloc(&d_b, h.size());
  (void)hipMemcpy(d_b, h.data(), h.size(), hipMemcpyHostToDevice);
  (void)hipMalloc(&d_o, 1024 * 32);
  const uint32_t iters = 300;
  for (int rep = 0; rep < 3; rep++)
    for (int blocks : {512, 256}) {
      (void)hipMemset(d_o, 0, 1024 * 32);
      hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), 0, 0, d_b, d_o, iters);
      (void)hipDeviceSynchronize();
      std::vector<uint64_t> o(blocks * 4);
      (void)hipMemcpy(o.data(), d_o, blocks * 32, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int b = 0; b < blocks; b++) {
        cyc += (double)o[4 * b];
        wall += (double)o[4 * b + 1];
      }
      const double per = cyc / blocks / iters;  // cycles one wave spends per statement
      printf("mixed addition, %d wave(s) per SIMD: %.0f cycles per statement and wave = %.0f SIMD cycles per wave-addition, clock %.0f MHz\n",
             blocks >= 512 ? 2 : 1, per, blocks >= 512 ? per / 2 : per, cyc / wall * 100.0);
    }
  // rounds: the same total work as 1 / 2 / 4 rounds of blocks (k_acc0 at 2^20 pairs: 1024 blocks of 64 iterations = 2 rounds);
  // kernel time against the time a wave is resident
  for (int rounds : {1, 2, 4, 2, 1}) {
    const int blocks = 512 * rounds;
    const uint32_t it = 128 / rounds;
    uint64_t* d_o2;
    (void)hipMalloc(&d_o2, (size_t)blocks * 32);
    (void)hipMemset(d_o2, 0, (size_t)blocks * 32);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), 0, 0, d_b, d_o2, it);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> o((size_t)blocks * 4);
    (void)hipMemcpy(o.data(), d_o2, (size_t)blocks * 32, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; b++) {
      cyc += (double)o[4 * b];
      wall += (double)o[4 * b + 1];
    }
    {  // where the blocks ran: HW_ID without its wave / SIMD / queue fields, per XCC
      std::vector<std::pair<uint64_t, int>> ids;
      for (int b = 0; b < blocks; b++) {
        const uint64_t key = o[4 * b + 3] & 0xf0000ff00ull;  // XCC id | SE, SH, CU fields of HW_ID
        bool found = false;
        for (auto& kv : ids)
          if (kv.first == key) {
            kv.second++;
            found = true;
            break;
          }
        if (!found) ids.push_back({key, 1});
      }
      int hist[16] = {0};
      for (auto& kv : ids) hist[std::min(kv.second, 15)]++;
      printf("  [%d blocks] distinct CU ids %zu; blocks per CU id:", blocks, ids.size());
      for (int k = 1; k < 16; k++)
        if (hist[k]) printf(" %d CUs x %d", hist[k], k);
      printf("\n");
      // timeline per XCC (the 100 MHz counter is shared inside one): first start, last start, last end
      for (uint32_t x = 0; x < 8; x++) {
        uint64_t s_min = ~0ull, s_max = 0, e_max = 0;
        int cnt = 0;
        for (int b = 0; b < blocks; b++) {
          if (((o[4 * b + 3] >> 32) & 15u) != x) continue;
          cnt++;
          s_min = std::min(s_min, o[4 * b + 2]);
          s_max = std::max(s_max, o[4 * b + 2]);
          e_max = std::max(e_max, o[4 * b + 2] + o[4 * b + 1]);
        }
        if (cnt) printf("    XCC %u: %d blocks, last start %.3f ms after the first, last end %.3f ms after the first start\n", x, cnt, (s_max - s_min) / 1e5, (e_max - s_min) / 1e5);
      }
    }
    const double resident_ms = wall / blocks / 100.0 / 1000.0;  // 100 MHz ticks -> ms
    printf("%d round(s) of 512 blocks x %u iterations: kernel %.3f ms, wave resident %.3f ms each (x rounds = %.3f ms), %.0f cycles per iteration, clock %.0f MHz\n",
           rounds, it, ms, resident_ms, resident_ms * rounds, cyc / blocks / it, cyc / wall * 100.0);
    (void)hipFree(d_o2);
  }
  // TRUE cost: SIMD cycles per wave-addition = (last end - first start) x clock / (additions per SIMD), one round of W waves per SIMD.
  // (the per-wave averages above are NOT that: co-resident waves progress unequally -- the older one issues first -- and finish
  // at different times)
  for (int W : {1, 2, 3, 2, 1, 3}) {
    const int blocks = 256 * W;
    const uint32_t it = 96;
    uint64_t* d_o2;
    (void)hipMalloc(&d_o2, (size_t)blocks * 32);
    (void)hipMemset(d_o2, 0, (size_t)blocks * 32);
    if (W == 3)
      hipLaunchKernelGGL(k_madd<3>, dim3(blocks), dim3(256), 0, 0, d_b, d_o2, it);
    else
      hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), 0, 0, d_b, d_o2, it);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> o((size_t)blocks * 4);
    (void)hipMemcpy(o.data(), d_o2, (size_t)blocks * 32, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0, span = 0;
    int nx = 0;
    for (uint32_t x = 0; x < 8; x++) {
      uint64_t s_min = ~0ull, e_max = 0;
      for (int b = 0; b < blocks; b++)
        if (((o[4 * b + 3] >> 32) & 15u) == x) {
          s_min = std::min(s_min, o[4 * b + 2]);
          e_max = std::max(e_max, o[4 * b + 2] + o[4 * b + 1]);
        }
      if (e_max) {
        span += (double)(e_max - s_min);
        nx++;
      }
    }
    uint64_t fastest = ~0ull, slowest = 0;
    for (int b = 0; b < blocks; b++) {
      cyc += (double)o[4 * b];
      wall += (double)o[4 * b + 1];
      fastest = std::min(fastest, o[4 * b + 1]);
      slowest = std::max(slowest, o[4 * b + 1]);
    }
    const double mhz = cyc / wall * 100.0;
    const double span_cycles = span / nx / 100.0 * mhz;  // ticks -> us -> cycles
    printf("%d wave(s) per SIMD, one round: %.0f SIMD cycles per wave-addition (span %.3f ms at %.0f MHz); a wave is resident %.3f .. %.3f ms\n", W,
           span_cycles / (it * W), span / nx / 1e5, mhz, fastest / 1e5, slowest / 1e5);
    (void)hipFree(d_o2);
  }
  for (uint32_t lg : {10u, 20u, 24u}) {
    uint8_t* d_tab;
    const size_t bytes = ((size_t)1 << lg) * AFF_BYTES;
    (void)hipMalloc(&d_tab, bytes);
    std::vector<uint8_t> big(std::min<size_t>(bytes, (size_t)1 << 26));
    for (size_t i = 0; i < big.size(); i++) big[i] = h[i % h.size()];
    for (size_t off = 0; off < bytes; off += big.size()) (void)hipMemcpy(d_tab + off, big.data(), std::min(big.size(), bytes - off), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
      run_gather<0>(d_tab, lg, d_o);
      run_gather<1>(d_tab, lg, d_o);
    }
    (void)hipFree(d_tab);
  }
  return 0;
}

// G1 multi-scalar multiplication for gfx950: signed-digit bucket method (Pippenger).
//
// Replaces ark-ec 0.4.2 `VariableBaseMSM::msm_bigint` -- in-tree statement of the algorithm:
// src/kzg/msm/variable_base.rs:21-61 (signed digits), :99-176 (buckets, running sum, Horner).
// The result is the group element sum_i s_i P_i, which does not depend on the window width,
// bucket order or summation order, so it is bit-exact against the reference after
// normalisation.  The decomposition below is GPU-first, not a translation of the CPU loop:
//
//   1. k_msm_hist     scalars -> signed c-bit digits; histogram of (window, |digit|) keys
//   2. k_scan         exclusive scan -> bucket offsets
//   3. k_msm_scatter  counting sort: entries[] = (key, sign, pair index) grouped by key
//   4. k_acc0         the hot kernel.  Thread t sums the affine bases of entries
//                     [t*L, (t+1)*L) into an XYZZ accumulator, run by run.  Work per lane is
//                     exactly L mixed additions whatever the digit distribution is -- the
//                     reference's own benchmark inputs put EVERY point of a window in one bucket
//                     (src/circuit.rs:349-365), which serialises a thread-per-bucket kernel.
//                     Runs interior to a chunk are complete buckets; the first and last run
//                     of a chunk are emitted as keyed partials.
//   5. k_merge        wave-cooperative segmented reduction of keyed partials (128 slots ->
//                     <= 2 per wave per level, values staged in LDS, keys in registers, 64-wide
//                     shuffles), repeated until one wave remains.
//   6. k_group_sum    bucket reduction by plain sums only: sum_b b*B_b = sum_j 2^j Z_j + Tot
//                     with Z_j = sum of buckets whose index has bit j set, computed as row /
//                     column sums followed by bit-plane sums (16 lanes per output, tree depth
//                     instead of the 2*2^(c-1) sequential additions of the CPU running sum).
//   7. host           Horner over <= 256 bit positions on the (W x c) plane sums
//                     (src/kzg/msm/variable_base.rs:168-175 is the window Horner).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sched.h>
#include <cstdio>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "g1.cuh"
#include "host_field.hpp"

namespace gm {

constexpr uint32_t KEY_INV = 0xffffffffu;
constexpr int XYZZ_BYTES = 192;
constexpr int AFF_BYTES = 96;

// ------------------------------------------------------------------------------------------
// digits
// ------------------------------------------------------------------------------------------
struct DigitIter {
  uint32_t s[8];
  uint32_t carry;
  bool bad;  // the scalar was not a canonical Fr image (>= 2^255): top digit clamped, call rejected
  GM_DEV void init(const uint32_t* p, bool active, int mont) {
    if (active) {
      Fr v = fp_load<FrParams>(p);
      if (mont) v = fp_from_mont<FrParams>(v);  // Fr::into_bigint
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = v.l[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = 0;
    }
    carry = 0;
    bad = (s[7] >> 31) != 0;  // >= 2^255: not the BigInt image of an Fr element (< r < 2^255)
  }
  GM_DEV void init_mag(const uint32_t m4[4], bool active, bool bad_in) {
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (active && i < 4) ? m4[i] : 0u;
    carry = 0;
    bad = bad_in;
  }
  // top-window digit with the final carry folded back in (variable_base.rs:58: digits[last] += carry << w).
  // For a canonical scalar (< r < 2^255) |d| <= 2^(c-1); a larger value would index past the bucket array
  // (the reference panics there), so it is clamped and flagged instead.
  GM_DEV int32_t last(int c) {
    int32_t d = next(c) + (int32_t)(carry << c);
    const int32_t B = 1 << (c - 1);
    if (d > B) {
      d = B;
      bad = true;
    }
    return d;
  }
  // next signed digit in [-2^(c-1), 2^(c-1)]; semantics of variable_base.rs:21-61.  With
  // c * W >= 256 the top window's raw value is < 2^(c-1), so after the caller folds the final
  // carry back in (variable_base.rs:58) the top digit is still within the bucket range.
  GM_DEV int32_t next(int c) {
    uint32_t raw = s[0] & ((1u << c) - 1u);
#pragma unroll
    for (int i = 0; i < 7; i++) s[i] = (s[i] >> c) | (s[i + 1] << (32 - c));
    s[7] >>= c;
    uint32_t coef = raw + carry;
    carry = (coef + (1u << (c - 1))) >> c;
    return (int32_t)coef - (int32_t)(carry << c);
  }
};

// GLV addressing (two half-length digit strings per scalar, the second one on phi(P)) and the window-group split of one-call
// MSMs are measured-negative experiments: their code is in the hot translation unit only with
// -DGM_EXPERIMENTS; otherwise GM_GLV(x) is the constant 0 and the compiler drops every branch that mentions it.
#ifdef GM_EXPERIMENTS
#include "msm_glv.inc"
#define GM_GLV(x) (x)
#else
#define GM_GLV(x) 0
#endif

constexpr int ENTRY_HALF_SHIFT = 30;  // entry idx field, GLV calls: pair index in bits 0..25, bit 30 = the phi(P) half
// the digit strings of one scalar: one string over the whole scalar, or -- GLV -- two strings over |v1|, |v2|
struct ScalarDigits {
  DigitIter it;
  bool bad;
#ifdef GM_EXPERIMENTS
  GlvHalves h;
  GM_DEV void load(const uint32_t* p, bool active, int mont, int glv) {
    it.init(p, active, mont);
    bad = it.bad;
    if (glv) h = glv_split(it.s);
  }
  GM_DEV void start(int half, bool active, int glv) {
    if (glv) it.init_mag(h.m[half], active, bad);
  }
  GM_DEV bool neg(int half) const { return h.neg[half]; }
#else
  GM_DEV void load(const uint32_t* p, bool active, int mont, int) {
    it.init(p, active, mont);
    bad = it.bad;
  }
  GM_DEV void start(int, bool, int) {}
  GM_DEV bool neg(int) const { return false; }
#endif
};

// One atomic per wave when every participating lane has the same key (the all-equal-scalars
// case would otherwise serialise 64 same-address atomics per instruction).
GM_DEV uint32_t wave_atomic_inc(uint32_t* arr, uint32_t key) {
  const int lane = threadIdx.x & 63;
  uint64_t valid = __ballot(key != KEY_INV);
  if (valid == 0) return 0;
  int first_lane = __ffsll((unsigned long long)valid) - 1;
  uint32_t first = __shfl(key, first_lane);
  uint64_t same = __ballot(key == first);
  uint32_t pos = 0;
  if (same == valid) {
    uint32_t base = 0;
    if (lane == first_lane) base = atomicAdd(arr + first, (uint32_t)__popcll((unsigned long long)same));
    base = __shfl(base, first_lane);
    pos = base + (uint32_t)__popcll((unsigned long long)(same & ((1ull << lane) - 1ull)));
  } else if (key != KEY_INV) {
    pos = atomicAdd(arr + key, 1u);
  }
  return pos;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_msm_digits(const uint32_t* __restrict__ scalars, uint32_t n, int mont, int c,
                                                    int W, uint32_t B, uint32_t* __restrict__ counts_or_cursor,
                                                    uint64_t* __restrict__ entries, uint32_t* __restrict__ err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool active = i < n;
  DigitIter it;
  it.init(scalars + 8 * (size_t)(active ? i : 0), active, mont);
  for (int w = 0; w < W; w++) {
    int32_t d = w == W - 1 ? it.last(c) : it.next(c);
    uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    uint32_t key = (active && d != 0) ? (uint32_t)w * B + (mag - 1u) : KEY_INV;
    uint32_t pos = wave_atomic_inc(counts_or_cursor, key);
    if (SCATTER && key != KEY_INV) {
      entries[pos] = ((uint64_t)key << 32) | ((uint64_t)(d < 0 ? 1u : 0u) << 31) | (uint64_t)i;
    }
  }
  if (!SCATTER && it.bad) atomicOr(err, 1u);
}

// ------------------------------------------------------------------------------------------
// Two-pass MSD counting sort with block-local aggregation in LDS.
//
// The per-entry global atomics of k_msm_digits cost one L2 atomic + one scattered 8-byte write per
// (scalar, window) pair.  Here every block first counts its tile in LDS, reserves contiguous
// output ranges with ONE global atomic per non-empty bin, and ranks its entries with LDS atomics:
//   pass 1  bin = key >> FB  (window, high bucket bits; G = W * 2^(c-1-FB) <= 4096 bins)
//   pass 2  bin = key & (2^FB - 1) inside one coarse group, blocks own (group, chunk) pairs
// Global atomics drop from n*W to ~(n*W / tile) * bins; writes of a block to one bin are contiguous.
// Skewed inputs (all-equal scalars) only serialise LDS atomics inside a block.
// ------------------------------------------------------------------------------------------
constexpr uint32_t SORT_TS = 1024;     // scalars per block in pass 1
constexpr uint32_t SORT_CH = 16384;    // entries per block in pass 2
constexpr uint32_t SORT_GMAX = 4096;   // coarse bins
constexpr uint32_t SORT_FMAX = 4096;   // fine bins (10 key bits by default, up to 12 for the widest windows)

struct SortGeom {
  int c, W;      // W: windows of the whole scalar (the signed-digit recurrence always runs from window 0)
  int w_lo, Wg;  // this call sorts windows [w_lo, w_lo + Wg) only; keys are relative to w_lo
  int glv;       // 1: every scalar contributes two half-length digit strings (v1 on P, v2 on phi(P)) over the same W windows
  uint32_t B, FB, G;
  int shared;  // 1: fixed-base tables in use -> one bucket set for all windows, window index rides in the entry
};
constexpr int ENTRY_W_SHIFT = 26;  // entry idx field: pair index in bits 0..25, window in bits 26..30 (tables only)

// SEVERAL calls as the LEVELS of one block-sorted pass (the folding commitments of a tensor check: 2^20, 2^19, ... pairs against
// slices of one key).  Element i of the concatenation belongs to level l = the range start[l] <= i < start[l + 1]; its scalar
// is scal[l][i - start[l]], its base index base0[l] + step (i - start[l]) -- ABSOLUTE in the entry: k_acc0 then runs with first = 0,
// step = 1 --, its bucket set (l, window) (tables: one set per level, the window rides in the entry as for one call).
// levels = 0: one call, pair i is scalar i and the entry carries i.
constexpr int MULTI_MAX_LEVELS = 16;
struct LevelGeom {
  int levels, step;
  uint32_t start[MULTI_MAX_LEVELS + 1];
  const uint32_t* scal[MULTI_MAX_LEVELS];
  long long base0[MULTI_MAX_LEVELS];
};
struct LevelOf {
  const uint32_t* sp;
  uint32_t keyoff, idx;
};
// where pair i of a (possibly levelled) pass finds its scalar, which bucket sets it feeds, what its entry carries
GM_DEV LevelOf level_of(const LevelGeom& lg, const SortGeom& sg, const uint32_t* scalars, uint32_t i, bool active) {
  LevelOf r;
  if (lg.levels == 0 || !active) {
    r.sp = (lg.levels == 0 ? scalars : lg.scal[0]) + 8 * (size_t)(active ? i : 0);
    r.keyoff = 0;
    r.idx = i;
    return r;
  }
  int l = 0;
  while (l + 1 < lg.levels && i >= lg.start[l + 1]) l++;
  const uint32_t il = i - lg.start[l];
  r.sp = lg.scal[l] + 8 * (size_t)il;
  r.keyoff = (uint32_t)l * (sg.shared ? sg.B : (uint32_t)sg.Wg * sg.B);
  r.idx = (uint32_t)(lg.base0[l] + (long long)lg.step * (long long)il);
  return r;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_sort1(const uint32_t* __restrict__ scalars, uint32_t n, int mont, SortGeom sg, LevelGeom lg,
                                               uint32_t* __restrict__ gcount_or_cursor, uint64_t* __restrict__ tmp, uint32_t* __restrict__ err) {
  __shared__ uint32_t cnt[SORT_GMAX];
  __shared__ uint32_t base[SCATTER ? SORT_GMAX : 1];
  for (uint32_t g = threadIdx.x; g < sg.G; g += 256) cnt[g] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * SORT_TS;
  for (uint32_t s = threadIdx.x; s < SORT_TS; s += 256) {
    const uint32_t i = first + s;
    const bool active = i < n;
    const LevelOf lv = level_of(lg, sg, scalars, i, active);
    ScalarDigits sd;
    sd.load(lv.sp, active, mont, GM_GLV(sg.glv));
    DigitIter& it = sd.it;
    for (int half = 0; half <= GM_GLV(sg.glv); half++) {
      sd.start(half, active, GM_GLV(sg.glv));
      for (int w = 0; w < sg.w_lo + sg.Wg; w++) {
        int32_t d = w == sg.W - 1 ? it.last(sg.c) : it.next(sg.c);
        if (w < sg.w_lo) continue;  // wave-uniform
        // (wave_atomic_inc: one LDS atomic per wave when all lanes hit the same bin -- the all-equal-scalars
        // instance of the reference's benchmark would otherwise serialise 64 same-address atomics)
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        wave_atomic_inc(cnt, (active && d != 0) ? ((lv.keyoff + (sg.shared ? 0u : (uint32_t)(w - sg.w_lo) * sg.B) + (mag - 1u)) >> sg.FB) : KEY_INV);
      }
    }
    if (!SCATTER && sd.bad) atomicOr(err, 1u);  // a scalar >= 2^255 (not an Fr image): the call fails with GM_EINVAL
  }
  __syncthreads();
  if (!SCATTER) {
    for (uint32_t g = threadIdx.x; g < sg.G; g += 256)
      if (cnt[g]) atomicAdd(gcount_or_cursor + g, cnt[g]);
    return;
  }
  for (uint32_t g = threadIdx.x; g < sg.G; g += 256) {
    const uint32_t k = cnt[g];
    if (SCATTER) base[g] = k ? atomicAdd(gcount_or_cursor + g, k) : 0u;
    cnt[g] = 0;
  }
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < SORT_TS; s += 256) {
    const uint32_t i = first + s;
    const bool active = i < n;
    const LevelOf lv = level_of(lg, sg, scalars, i, active);
    ScalarDigits sd;
    sd.load(lv.sp, active, mont, GM_GLV(sg.glv));
    DigitIter& it = sd.it;
    for (int half = 0; half <= GM_GLV(sg.glv); half++) {
      sd.start(half, active, GM_GLV(sg.glv));
      const bool hneg = GM_GLV(sg.glv) && sd.neg(half);
      for (int w = 0; w < sg.w_lo + sg.Wg; w++) {
        int32_t d = w == sg.W - 1 ? it.last(sg.c) : it.next(sg.c);
        if (w < sg.w_lo) continue;
        const bool live = active && d != 0;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint32_t key = lv.keyoff + (sg.shared ? 0u : (uint32_t)(w - sg.w_lo) * sg.B) + (mag - 1u);
        const uint32_t g = key >> sg.FB;
        const uint32_t r = wave_atomic_inc(cnt, live ? g : KEY_INV);
        if (live) {
          const uint32_t idx = sg.shared ? (lv.idx | ((uint32_t)w << ENTRY_W_SHIFT)) : (lv.idx | ((uint32_t)half << ENTRY_HALF_SHIFT));
          tmp[(SCATTER ? base[g] : 0u) + r] = ((uint64_t)key << 32) | ((uint64_t)(((d < 0) != hneg) ? 1u : 0u) << 31) | (uint64_t)idx;
        }
      }
    }
  }
}

// Scatter half of pass 1, staged in LDS like k_sort2_staged: one thread per scalar (1024 per block) keeps
// its <= 16 entries in registers, the block counting-sorts them by coarse bin inside LDS and streams the
// image out, so a wave writes runs of neighbouring addresses instead of 64 scattered 8-byte words.
// Dynamic LDS: SORT_TS * W entries + 3 G counters + the scan array (<= 160 KiB for c >= 16).
constexpr int SORT1_STAGE_WMAX = 16;
__global__ __launch_bounds__(1024) void k_sort1_staged(const uint32_t* __restrict__ scalars, uint32_t n, int mont, SortGeom sg, LevelGeom lg,
                                                       uint32_t* __restrict__ gcursor, uint64_t* __restrict__ tmp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(buf + (size_t)SORT_TS * sg.Wg * (1 + GM_GLV(sg.glv)));
  uint32_t* base = cnt + sg.G;
  uint32_t* lst = base + sg.G;
  uint32_t* scan = lst + sg.G;
  const uint32_t tid = threadIdx.x;
  for (uint32_t g = tid; g < sg.G; g += 1024) cnt[g] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * SORT_TS + tid;
  const bool active = i < n;
  uint64_t e[SORT1_STAGE_WMAX];
  {
    const LevelOf lv = level_of(lg, sg, scalars, i, active);
    ScalarDigits sd;
    sd.load(lv.sp, active, mont, GM_GLV(sg.glv));
    DigitIter& it = sd.it;
    // GLV: the register slots are split between the halves (Wg <= SORT1_STAGE_WMAX / 2 windows each)
    constexpr int HS = SORT1_STAGE_WMAX / 2;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (half <= GM_GLV(sg.glv)) {
        sd.start(half, active, GM_GLV(sg.glv));
        for (int w = 0; w < sg.w_lo; w++) it.next(sg.c);  // the recurrence starts at window 0
      }
      const bool hneg = GM_GLV(sg.glv) && half <= GM_GLV(sg.glv) && sd.neg(half);
#pragma unroll
      for (int jj = 0; jj < HS; jj++) {
        // without GLV the one string fills all SORT1_STAGE_WMAX slots: "half 1" continues it at window w_lo + HS
        const int j = half * HS + jj;
        const int jw = GM_GLV(sg.glv) ? jj : j;  // window of this slot, relative to w_lo
        const int w = sg.w_lo + jw;
        e[j] = ~0ull;
        if (jw < sg.Wg && (GM_GLV(sg.glv) || half == 0 || sg.Wg > HS)) {
          int32_t d = w == sg.W - 1 ? it.last(sg.c) : it.next(sg.c);
          if (active && d != 0) {
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            const uint32_t key = lv.keyoff + (sg.shared ? 0u : (uint32_t)jw * sg.B) + (mag - 1u);
            const uint32_t idx = sg.shared ? (lv.idx | ((uint32_t)w << ENTRY_W_SHIFT)) : (lv.idx | ((uint32_t)(GM_GLV(sg.glv) ? half : 0) << ENTRY_HALF_SHIFT));
            e[j] = ((uint64_t)key << 32) | ((uint64_t)(((d < 0) != hneg) ? 1u : 0u) << 31) | (uint64_t)idx;
          }
        }
        wave_atomic_inc(cnt, e[j] != ~0ull ? ((uint32_t)(e[j] >> 32) >> sg.FB) : KEY_INV);
      }
    }
  }
  __syncthreads();
  const uint32_t per = (sg.G + 1023u) / 1024u;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t g = tid * per + k;
    if (g < sg.G) {
      const uint32_t c = cnt[g];
      base[g] = c ? atomicAdd(gcursor + g, c) : 0u;
      s += c;
    }
  }
  scan[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t x = tid >= d ? scan[tid - d] : 0;
    __syncthreads();
    scan[tid] += x;
    __syncthreads();
  }
  const uint32_t m = scan[1023];
  uint32_t run = scan[tid] - s;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t g = tid * per + k;
    if (g < sg.G) {
      lst[g] = run;
      run += cnt[g];
    }
  }
  __syncthreads();
  for (uint32_t g = tid; g < sg.G; g += 1024) cnt[g] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < SORT1_STAGE_WMAX; w++) {
    const bool live = e[w] != ~0ull;
    const uint32_t g = (uint32_t)(e[w] >> 32) >> sg.FB;
    const uint32_t r = wave_atomic_inc(cnt, live ? g : KEY_INV);
    if (live) buf[lst[g] + r] = e[w];
  }
  __syncthreads();
  for (uint32_t j = tid; j < m; j += 1024) {
    const uint64_t v = buf[j];
    const uint32_t g = (uint32_t)(v >> 32) >> sg.FB;
    tmp[base[g] + (j - lst[g])] = v;
  }
}

// exclusive scan of the G coarse counters + prefix of the pass-2 block counts (one block)
__global__ __launch_bounds__(1024) void k_sort1_scan(const uint32_t* __restrict__ gcount, uint32_t G,
                                                     uint32_t* __restrict__ goff, uint32_t* __restrict__ gcursor,
                                                     uint32_t* __restrict__ blkoff) {
  __shared__ uint32_t a[1024], b[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (G + 1023u) / 1024u;  // <= 4
  uint32_t v[4], nb[4], sa = 0, sb = 0;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t g = tid * per + k;
    v[k] = g < G ? gcount[g] : 0u;
    nb[k] = (v[k] + SORT_CH - 1u) / SORT_CH;
    sa += v[k];
    sb += nb[k];
  }
  a[tid] = sa;
  b[tid] = sb;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t xa = tid >= d ? a[tid - d] : 0, xb = tid >= d ? b[tid - d] : 0;
    __syncthreads();
    a[tid] += xa;
    b[tid] += xb;
    __syncthreads();
  }
  uint32_t ra = a[tid] - sa, rb = b[tid] - sb;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t g = tid * per + k;
    if (g < G) {
      goff[g] = ra;
      gcursor[g] = ra;
      blkoff[g] = rb;
    }
    ra += v[k];
    rb += nb[k];
  }
  if (tid == 1023) {
    goff[G] = a[1023];
    blkoff[G] = b[1023];
  }
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_sort2(const uint64_t* __restrict__ tmp, const uint32_t* __restrict__ goff,
                                               const uint32_t* __restrict__ blkoff, SortGeom sg,
                                               uint32_t* __restrict__ counts_or_cursor, uint64_t* __restrict__ entries) {
  __shared__ uint32_t cnt[SORT_FMAX];
  __shared__ uint32_t base[SCATTER ? SORT_FMAX : 1];
  __shared__ uint32_t s_g;
  const uint32_t nf = 1u << sg.FB;
  if (blockIdx.x >= blkoff[sg.G]) return;
  if (threadIdx.x == 0) {  // largest g with blkoff[g] <= blockIdx.x
    uint32_t lo = 0, hi = sg.G;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (blkoff[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    s_g = lo;
  }
  for (uint32_t f = threadIdx.x; f < nf; f += 256) cnt[f] = 0;
  __syncthreads();
  const uint32_t g = s_g;
  const uint32_t chunk = blockIdx.x - blkoff[g];
  const uint32_t lo = goff[g] + chunk * SORT_CH;
  const uint32_t hi = min(lo + SORT_CH, goff[g + 1]);
  const uint32_t mask = nf - 1u;
  for (uint32_t i0 = lo; i0 < hi; i0 += 256) {  // whole waves enter wave_atomic_inc together
    const uint32_t i = i0 + threadIdx.x;
    wave_atomic_inc(cnt, i < hi ? ((uint32_t)(tmp[i] >> 32) & mask) : KEY_INV);
  }
  __syncthreads();
  if (!SCATTER) {
    for (uint32_t f = threadIdx.x; f < nf; f += 256)
      if (cnt[f]) atomicAdd(counts_or_cursor + ((size_t)g << sg.FB) + f, cnt[f]);
    return;
  }
  for (uint32_t f = threadIdx.x; f < nf; f += 256) {
    const uint32_t k = cnt[f];
    if (SCATTER) base[f] = k ? atomicAdd(counts_or_cursor + ((size_t)g << sg.FB) + f, k) : 0u;
    cnt[f] = 0;
  }
  __syncthreads();
  for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
    const uint32_t i = i0 + threadIdx.x;
    const bool live = i < hi;
    const uint64_t e = live ? tmp[i] : 0;
    const uint32_t f = (uint32_t)(e >> 32) & mask;
    const uint32_t r = wave_atomic_inc(cnt, live ? f : KEY_INV);
    if (live) entries[(SCATTER ? base[f] : 0u) + r] = e;
  }
}

// Scatter half of pass 2 with the block's chunk staged in LDS.  k_sort2<true> writes each 8-byte entry
// straight to entries[base[f] + rank]: neighbouring lanes hold random fine bins, so a wave's store is 64
// separate 8-byte writes (a 64-byte sector each): 1.9 GB of entries cost 4.2 ms at 2^24 pairs, 0.9 TB/s.
// Here the block (1024 threads, the whole 160 KiB of LDS) first counting-sorts its 16 Ki entries by fine
// bin INSIDE LDS and then streams the LDS image out: lane i writes the i-th locally sorted entry, so
// neighbouring lanes write neighbouring addresses of the same bin (runs of chunk / bins = 8-16 entries).
constexpr uint32_t SORT_STAGE_FMAX = 2048;
__global__ __launch_bounds__(1024) void k_sort2_staged(const uint64_t* __restrict__ tmp, const uint32_t* __restrict__ goff,
                                                       const uint32_t* __restrict__ blkoff, SortGeom sg, uint32_t* __restrict__ cursor,
                                                       uint64_t* __restrict__ entries) {
  __shared__ uint64_t buf[SORT_CH];
  __shared__ uint32_t cnt[SORT_STAGE_FMAX], base[SORT_STAGE_FMAX], lst[SORT_STAGE_FMAX];
  __shared__ uint32_t scan[1024];
  __shared__ uint32_t s_g;
  const uint32_t tid = threadIdx.x;
  const uint32_t nf = 1u << sg.FB;
  if (blockIdx.x >= blkoff[sg.G]) return;
  if (tid == 0) {
    uint32_t lo = 0, hi = sg.G;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (blkoff[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    s_g = lo;
  }
  for (uint32_t f = tid; f < nf; f += 1024) cnt[f] = 0;
  __syncthreads();
  const uint32_t g = s_g;
  const uint32_t chunk = blockIdx.x - blkoff[g];
  const uint32_t lo = goff[g] + chunk * SORT_CH;
  const uint32_t hi = min(lo + SORT_CH, goff[g + 1]);
  const uint32_t m = hi - lo;
  const uint32_t mask = nf - 1u;
  constexpr uint32_t PER = SORT_CH / 1024;
  uint64_t e[PER];
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t i = tid + k * 1024;
    e[k] = i < m ? tmp[lo + i] : 0;
    wave_atomic_inc(cnt, i < m ? ((uint32_t)(e[k] >> 32) & mask) : KEY_INV);
  }
  __syncthreads();
  // global ranges of this block's entries per bin + exclusive scan of the local counts
  const uint32_t per = nf > 1024 ? nf / 1024 : 1;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t f = tid * per + k;
    if (f < nf) {
      const uint32_t c = cnt[f];
      base[f] = c ? atomicAdd(cursor + ((size_t)g << sg.FB) + f, c) : 0u;
      s += c;
    }
  }
  scan[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t x = tid >= d ? scan[tid - d] : 0;
    __syncthreads();
    scan[tid] += x;
    __syncthreads();
  }
  uint32_t run = scan[tid] - s;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t f = tid * per + k;
    if (f < nf) {
      lst[f] = run;
      run += cnt[f];
    }
  }
  __syncthreads();
  for (uint32_t f = tid; f < nf; f += 1024) cnt[f] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t i = tid + k * 1024;
    const uint32_t f = (uint32_t)(e[k] >> 32) & mask;
    const uint32_t r = wave_atomic_inc(cnt, i < m ? f : KEY_INV);
    if (i < m) buf[lst[f] + r] = e[k];
  }
  __syncthreads();
  for (uint32_t i = tid; i < m; i += 1024) {
    const uint64_t v = buf[i];
    const uint32_t f = (uint32_t)(v >> 32) & mask;
    entries[base[f] + (i - lst[f])] = v;
  }
}

// exclusive scan of m counters in three launches: per-block sums (4096 counters per block),
// a single-block scan of the block sums, then the local scans with the block offsets applied
constexpr uint32_t SCAN_PER_THREAD = 16;
constexpr uint32_t SCAN_PER_BLOCK = 256 * SCAN_PER_THREAD;

GM_DEV uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* lds /* 256 */, uint32_t* total) {
  const uint32_t tid = threadIdx.x;
  lds[tid] = v;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t x = tid >= d ? lds[tid - d] : 0;
    __syncthreads();
    lds[tid] += x;
    __syncthreads();
  }
  if (total) *total = lds[255];
  return lds[tid] - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t* __restrict__ counts, uint32_t m,
                                                         uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t lds[256];
  const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  uint32_t s = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) s += base + k < m ? counts[base + k] : 0u;
  uint32_t total;
  block_exclusive_scan_256(s, lds, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* __restrict__ block_sums, uint32_t nb,
                                                   uint32_t* __restrict__ total_out) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (nb + 1023u) / 1024u;
  const uint32_t lo = min(tid * chunk, nb), hi = min(lo + chunk, nb);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += block_sums[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
  for (uint32_t i = lo; i < hi; i++) {
    uint32_t c = block_sums[i];
    block_sums[i] = run;
    run += c;
  }
  if (tid == 1023) *total_out = part[1023];
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ counts, uint32_t m,
                                                    const uint32_t* __restrict__ block_sums,
                                                    uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor) {
  __shared__ uint32_t lds[256];
  const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  uint32_t v[SCAN_PER_THREAD];
  uint32_t s = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
    v[k] = base + k < m ? counts[base + k] : 0u;
    s += v[k];
  }
  uint32_t run = block_exclusive_scan_256(s, lds, nullptr) + block_sums[blockIdx.x];
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
    if (base + k < m) {
      offsets[base + k] = run;
      cursor[base + k] = run;
    }
    run += v[k];
  }
}

// ------------------------------------------------------------------------------------------
// Small calls (n * W <= 2^18 entries): one thread per (scalar, window).  The signed-digit recurrence of
// variable_base.rs:21-61 carries from window to window, but the carry INTO window w is just the carry of the
// integer addition s + K out of the low c*w bits, K = sum_{j < W-1} 2^(c j + c - 1) (half a window in every
// window but the top one): coef_j >= 2^(c-1)  <=>  raw_j + carry_j + 2^(c-1) >= 2^c.  So digit_w = window w of
// (s + K) minus 2^(c-1), and the top window, which absorbs the final carry (:58), is window W-1 of s + K as it is.
// That makes every digit independent: a 2^10-pair call has 44 k threads instead of the 4 blocks of the
// per-scalar kernels (whose ~170 sequential digit extractions per thread run at lone-wave speed: 0.14 ms of a
// 0.68 ms call), and one counting pass with global atomics (wave-aggregated: lanes of a wave hold consecutive
// scalars of ONE window, so the all-equal-scalars instance costs one atomic per wave) replaces the two-pass
// block-local sort -- 3 launches instead of 9.
// ------------------------------------------------------------------------------------------
struct FlatGeom {
  int c, W;
  int glv;  // grid.y = 2 W: rows [0, W) are the v1 half, rows [W, 2 W) the v2 half (on phi(P))
  uint32_t B;
  uint32_t K[8];  // sum_{j < W-1} 2^(c j + c - 1), little-endian limbs
};
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_digits_flat(const uint32_t* __restrict__ scalars, uint32_t n, int mont, FlatGeom fg,
                                                     uint32_t* __restrict__ counts_or_cursor, uint64_t* __restrict__ entries,
                                                     uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = (int)blockIdx.y >= fg.W ? 1 : 0;
  const int w = (int)blockIdx.y - half * fg.W;
  const bool active = i < n;
  uint32_t key = KEY_INV;
  bool neg = false;
  if (active) {
    Fr v = fp_load<FrParams>(scalars + 8 * (size_t)i);
    if (mont) v = fp_from_mont<FrParams>(v);
    if (!SCATTER && blockIdx.y == 0 && (v.l[7] >> 31)) atomicOr(err, 1u);  // >= 2^255: not an Fr image
    bool hneg = false;
#ifdef GM_EXPERIMENTS
    if (fg.glv) {
      const GlvHalves h = glv_split(v.l);
      hneg = h.neg[half];
#pragma unroll
      for (int l = 0; l < 8; l++) v.l[l] = l < 4 ? h.m[half][l] : 0u;
    }
#endif
    uint32_t t[9];
    uint32_t carry = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      uint32_t cy;
      t[l] = __builtin_addc(v.l[l], fg.K[l], carry, &cy);
      carry = cy;
    }
    t[8] = carry;
    const uint32_t bit = (uint32_t)(fg.c * w), lo = bit >> 5, sh = bit & 31u;
    uint64_t two = 0;
#pragma unroll
    for (int l = 0; l < 9; l++) {  // limbs lo, lo + 1 without dynamic register indexing
      if ((uint32_t)l == lo) two |= (uint64_t)t[l];
      if ((uint32_t)l == lo + 1) two |= (uint64_t)t[l] << 32;
    }
    uint32_t u = (uint32_t)(two >> sh);
    int32_t d;
    if (w == fg.W - 1) {
      // the top window keeps every remaining bit (c * W >= 256, so they fit 32 bits only for a canonical scalar;
      // anything beyond the bucket range is clamped -- the call fails through the flag above)
      d = (int32_t)min(u, fg.B);
    } else {
      u &= (1u << fg.c) - 1u;
      d = (int32_t)u - (int32_t)(1u << (fg.c - 1));
    }
    if (d != 0) {
      key = (uint32_t)w * fg.B + ((uint32_t)(d < 0 ? -d : d) - 1u);
      neg = (d < 0) != hneg;
    }
  }
  const uint32_t pos = wave_atomic_inc(counts_or_cursor, key);
  if (SCATTER && key != KEY_INV)
    entries[pos] = ((uint64_t)key << 32) | ((uint64_t)(neg ? 1u : 0u) << 31) | (uint64_t)(i | ((uint32_t)half << ENTRY_HALF_SHIFT));
}

// Several SMALL MSMs in one pass (the folding commitments of a tensor check end in a dozen calls of 2^13 ... 2 pairs, each a
// latency-bound chain of ~15 launches that runs in ~0.3-0.5 ms whatever its size): the calls become "levels" of ONE call whose
// bucket sets are (level, window) pairs.  Element i of the concatenation belongs to level l = the range start[l] <= i < start[l + 1],
// its scalar is scal[l][i - start[l]], its base index base0[l] + step (i - start[l]) (absolute: k_acc0 runs with first = 0,
// step = 1), its key (l W + w) B + (|digit| - 1).  Digits as in k_digits_flat.
struct MultiGeom {
  int c, W, levels, step;
  uint32_t B;
  uint32_t K[8];
  uint32_t start[MULTI_MAX_LEVELS + 1];
  const uint32_t* scal[MULTI_MAX_LEVELS];
  long long base0[MULTI_MAX_LEVELS];
};
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_digits_multi(MultiGeom mg, int mont, uint32_t* __restrict__ counts_or_cursor, uint64_t* __restrict__ entries,
                                                      uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = (int)blockIdx.y;
  const bool active = i < mg.start[mg.levels];
  uint32_t key = KEY_INV, idx = 0;
  bool neg = false;
  if (active) {
    int l = 0;
    while (l + 1 < mg.levels && i >= mg.start[l + 1]) l++;
    const uint32_t il = i - mg.start[l];
    Fr v = fp_load<FrParams>(mg.scal[l] + 8 * (size_t)il);
    if (mont) v = fp_from_mont<FrParams>(v);
    if (!SCATTER && blockIdx.y == 0 && (v.l[7] >> 31)) atomicOr(err, 1u);  // >= 2^255: not an Fr image
    uint32_t t[9];
    uint32_t carry = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      uint32_t cy;
      t[q] = __builtin_addc(v.l[q], mg.K[q], carry, &cy);
      carry = cy;
    }
    t[8] = carry;
    const uint32_t bit = (uint32_t)(mg.c * w), lo = bit >> 5, sh = bit & 31u;
    uint64_t two = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) {
      if ((uint32_t)q == lo) two |= (uint64_t)t[q];
      if ((uint32_t)q == lo + 1) two |= (uint64_t)t[q] << 32;
    }
    uint32_t u = (uint32_t)(two >> sh);
    int32_t d;
    if (w == mg.W - 1) {
      d = (int32_t)min(u, mg.B);
    } else {
      u &= (1u << mg.c) - 1u;
      d = (int32_t)u - (int32_t)(1u << (mg.c - 1));
    }
    if (d != 0) {
      key = (uint32_t)(l * mg.W + w) * mg.B + ((uint32_t)(d < 0 ? -d : d) - 1u);
      neg = d < 0;
      idx = (uint32_t)(mg.base0[l] + (long long)mg.step * (long long)il);
    }
  }
  const uint32_t pos = wave_atomic_inc(counts_or_cursor, key);
  if (SCATTER && key != KEY_INV) entries[pos] = ((uint64_t)key << 32) | ((uint64_t)(neg ? 1u : 0u) << 31) | (uint64_t)idx;
}


// exclusive scan of m <= 2^18 counters in one block (the three-launch scan is for the millions of buckets of big calls)
__global__ __launch_bounds__(1024) void k_scan_small(const uint32_t* __restrict__ counts, uint32_t m, uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ cursor) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (m + 1023u) / 1024u;
  const uint32_t lo = min(tid * per, m), hi = min(lo + per, m);
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += counts[k];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t x = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t c = counts[k];
    offsets[k] = run;
    cursor[k] = run;
    run += c;
  }
  if (tid == 1023) offsets[m] = part[1023];
}

// ------------------------------------------------------------------------------------------
// level 0: chunk-per-thread accumulation of sorted entries
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// k_acc0: chunk-per-lane accumulation of sorted entries with the accumulator in the product's own representation
// (13 x 30-bit loose limbs, 52 pinned VGPRs) and the whole mixed addition ONE asm statement (gen_madd30.py,
// g1_madd30_gen.inc): no unpack / repack / conditional subtraction per product, no calls, one reduction for
// Y3 = R (Q - X3) - Y1 PPP.  Runs are written as 208-byte loose records (buckets, level-0 partials) that the
// consumers add in the same representation.  The statement is COMPLETE: an identity accumulator, a negated base,
// doubling and cancellation (every base is equal in the reference's elastic benchmark) are handled inside it.
// (Round 2's kernel kept a canonical 12 x 32-bit accumulator and called an out-of-line multiplier: 2.58 ms against 2.13 ms
// at 2^20 pairs, profiles/r3_ab_acc0_*.json.)
// ------------------------------------------------------------------------------------------
#ifdef GM_ACC0_CYCLES
__device__ unsigned long long gm_acc0_dbg[4];
#endif
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_acc0(const uint64_t* __restrict__ entries,
                                              const uint32_t* __restrict__ total_ptr,
                                              const uint8_t* __restrict__ bases, long long first, long long step,
                                              long long tab_stride, uint32_t L, uint32_t* __restrict__ pk,
                                              uint8_t* __restrict__ pp, uint8_t* __restrict__ buckets, const uint8_t* __restrict__ phi,
                                              unsigned long long* __restrict__ clk) {
  __shared__ uint64_t ebuf[8][256];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t clk_c0 = clock64(), clk_w0 = wall_clock64();  // the shader clock this launch ran at (lane 0 reports it)
  const uint64_t total = *total_ptr;
  const uint64_t start = (uint64_t)t * L;
  uint32_t head_key = KEY_INV, tail_key = KEY_INV;
#ifdef GM_ACC0_CYCLES
  const uint64_t dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif
  if (start < total) {
    const uint64_t end = min(start + (uint64_t)L, total);
    Acc30 acc;
    acc30_zero(acc);
    uint32_t cur = KEY_INV;
    bool first_run = true;
    for (uint64_t i = start; i < end; i++) {
      // Entries are staged eight at a time through LDS: a lane walks its own chunk, so lane t reads entries[t L + i]
      // -- 8 bytes out of a different 128-byte line per lane and iteration, and with one addition (~20 k cycles)
      // between two reads of a line it is long gone from L1 / L2 by then.  Four 16-byte loads per eight iterations fetch
      // each line twice; the transposed LDS image [k][lane] is written and read conflict-free and only by its own lane.
      const uint32_t k8 = (uint32_t)(i - start) & 7u;
      if (k8 == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(entries + i);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint4 v = src[q];
          ebuf[2 * q][threadIdx.x] = ((uint64_t)v.y << 32) | v.x;
          ebuf[2 * q + 1][threadIdx.x] = ((uint64_t)v.w << 32) | v.z;
        }
      }
      const uint64_t e = ebuf[k8][threadIdx.x];
      long long idx;
      if (tab_stride) {
        const uint32_t lo = (uint32_t)e & 0x7fffffffu;
        idx = (long long)(lo >> ENTRY_W_SHIFT) * tab_stride + first + step * (long long)(lo & ((1u << ENTRY_W_SHIFT) - 1u));
      } else {
        idx = first + step * (long long)(e & 0x3fffffffull);
      }
      const uint8_t* src = (phi != nullptr && ((e >> ENTRY_HALF_SHIFT) & 1ull)) ? phi : bases;
      const gm_u4v* bp = reinterpret_cast<const gm_u4v*>(src + (size_t)idx * AFF_BYTES);
      const gm_u4v x0 = bp[0], x1 = bp[1], x2 = bp[2], y0 = bp[3], y1 = bp[4], y2 = bp[5];
      const uint32_t key = (uint32_t)(e >> 32);
      if (key != cur) {
        if (cur != KEY_INV) {
          if (first_run) {
            head_key = cur;
            acc30_store(pp + (size_t)(2 * (size_t)t) * XYZZ30_BYTES, acc);
            first_run = false;
          } else {
            acc30_store(buckets + (size_t)cur * XYZZ30_BYTES, acc);  // interior run = whole bucket
          }
        }
        cur = key;
        acc30_set_identity(acc);
      }
      const uint32_t nz = x0.x | x0.y | x0.z | x0.w | x1.x | x1.y | x1.z | x1.w | x2.x | x2.y | x2.z | x2.w | y0.x | y0.y | y0.z | y0.w |
                          y1.x | y1.y | y1.z | y1.w | y2.x | y2.y | y2.z | y2.w;
      if (nz == 0) continue;  // the identity base (0, 0)
      const uint32_t neg = (uint32_t)(e >> 31) & 1u;
      // complete: identity accumulator, doubling and cancellation are handled inside the statement (gen_madd30.py)
      (void)g1_madd30_asm(acc, x0, x1, x2, y0, y1, y2, neg);
    }
    if (first_run) {
      head_key = cur;
      acc30_store(pp + (size_t)(2 * (size_t)t) * XYZZ30_BYTES, acc);
    } else {
      tail_key = cur;
      acc30_store(pp + (size_t)(2 * (size_t)t + 1) * XYZZ30_BYTES, acc);
    }
  }
  pk[2 * (size_t)t] = head_key;
  pk[2 * (size_t)t + 1] = tail_key;
  if (t == 0 && clk != nullptr) {
    clk[0] = (unsigned long long)(clock64() - clk_c0);
    clk[1] = (unsigned long long)(wall_clock64() - clk_w0);
  }
#ifdef GM_ACC0_CYCLES
  if ((threadIdx.x & 63) == 0) {  // per wave: shader cycles, 100 MHz ticks, waves (make EXTRA=-DGM_ACC0_CYCLES; tools/acc0_cycles.py)
    atomicAdd(&gm_acc0_dbg[0], (unsigned long long)(clock64() - dbg_c0));
    atomicAdd(&gm_acc0_dbg[1], (unsigned long long)(wall_clock64() - dbg_w0));
    atomicAdd(&gm_acc0_dbg[2], 1ull);
  }
#endif
}

// The same accumulation with the gather of entry i + 1 IN FLIGHT during the addition of entry i (24 more registers).  On the
// plain path the 96 B x n bases sit in the Infinity Cache and the wait is short; the fixed-base tables are 13 x that (1.3 GB
// at 2^20 points, 38 GB at 2^25) and every gather is a random HBM access of ~1-2 us, a tenth of the addition it feeds.
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_acc0_pf(const uint64_t* __restrict__ entries,
                                              const uint32_t* __restrict__ total_ptr,
                                              const uint8_t* __restrict__ bases, long long first, long long step,
                                              long long tab_stride, uint32_t L, uint32_t* __restrict__ pk,
                                              uint8_t* __restrict__ pp, uint8_t* __restrict__ buckets, const uint8_t* __restrict__ phi,
                                              unsigned long long* __restrict__ clk) {
  __shared__ uint64_t ebuf[8][256];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t clk_c0 = clock64(), clk_w0 = wall_clock64();  // the shader clock this launch ran at (lane 0 reports it)
  const uint64_t total = *total_ptr;
  const uint64_t start = (uint64_t)t * L;
  uint32_t head_key = KEY_INV, tail_key = KEY_INV;
  if (start < total) {
    const uint64_t end = min(start + (uint64_t)L, total);
    Acc30 acc;
    acc30_zero(acc);
    uint32_t cur = KEY_INV;
    bool first_run = true;
    // entry i of this lane (staged eight at a time through the transposed LDS image, as in k_acc0)
    auto entry_at = [&](uint64_t i) {
      const uint32_t k8 = (uint32_t)(i - start) & 7u;
      if (k8 == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(entries + i);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint4 v = src[q];
          ebuf[2 * q][threadIdx.x] = ((uint64_t)v.y << 32) | v.x;
          ebuf[2 * q + 1][threadIdx.x] = ((uint64_t)v.w << 32) | v.z;
        }
      }
      return ebuf[k8][threadIdx.x];
    };
    auto base_of = [&](uint64_t e) {
      long long idx;
      if (tab_stride) {
        const uint32_t lo = (uint32_t)e & 0x7fffffffu;
        idx = (long long)(lo >> ENTRY_W_SHIFT) * tab_stride + first + step * (long long)(lo & ((1u << ENTRY_W_SHIFT) - 1u));
      } else {
        idx = first + step * (long long)(e & 0x3fffffffull);
      }
      const uint8_t* src = (phi != nullptr && ((e >> ENTRY_HALF_SHIFT) & 1ull)) ? phi : bases;
      return reinterpret_cast<const gm_u4v*>(src + (size_t)idx * AFF_BYTES);
    };
    uint64_t e = entry_at(start);
    const gm_u4v* bp = base_of(e);
    gm_u4v x0 = bp[0], x1 = bp[1], x2 = bp[2], y0 = bp[3], y1 = bp[4], y2 = bp[5];
    for (uint64_t i = start; i < end; i++) {
      uint64_t en = 0;
      gm_u4v n0 = x0, n1 = x1, n2 = x2, n3 = y0, n4 = y1, n5 = y2;
      if (i + 1 < end) {
        // issued here, waited for after the addition.  The loads are asm so that the compiler can neither sink them below the
        // statement nor wait for them in front of it; its own s_waitcnt bookkeeping stays safe (extra outstanding loads only
        // make a vmcnt(k) wait longer, never shorter)
        en = entry_at(i + 1);
        const gm_u4v* np = base_of(en);
        asm volatile(
            "global_load_dwordx4 %0, %6, off\n\t"
            "global_load_dwordx4 %1, %6, off offset:16\n\t"
            "global_load_dwordx4 %2, %6, off offset:32\n\t"
            "global_load_dwordx4 %3, %6, off offset:48\n\t"
            "global_load_dwordx4 %4, %6, off offset:64\n\t"
            "global_load_dwordx4 %5, %6, off offset:80"
            : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5)
            : "v"(np));
      }
      const uint32_t key = (uint32_t)(e >> 32);
      if (key != cur) {
        if (cur != KEY_INV) {
          if (first_run) {
            head_key = cur;
            acc30_store(pp + (size_t)(2 * (size_t)t) * XYZZ30_BYTES, acc);
            first_run = false;
          } else {
            acc30_store(buckets + (size_t)cur * XYZZ30_BYTES, acc);  // interior run = whole bucket
          }
        }
        cur = key;
        acc30_set_identity(acc);
      }
      const uint32_t nz = x0.x | x0.y | x0.z | x0.w | x1.x | x1.y | x1.z | x1.w | x2.x | x2.y | x2.z | x2.w | y0.x | y0.y | y0.z | y0.w |
                          y1.x | y1.y | y1.z | y1.w | y2.x | y2.y | y2.z | y2.w;
      if (nz != 0) {  // not the identity base (0, 0)
        const uint32_t neg = (uint32_t)(e >> 31) & 1u;
        (void)g1_madd30_asm(acc, x0, x1, x2, y0, y1, y2, neg);
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5));
      e = en;
      x0 = n0; x1 = n1; x2 = n2; y0 = n3; y1 = n4; y2 = n5;
    }
    if (first_run) {
      head_key = cur;
      acc30_store(pp + (size_t)(2 * (size_t)t) * XYZZ30_BYTES, acc);
    } else {
      tail_key = cur;
      acc30_store(pp + (size_t)(2 * (size_t)t + 1) * XYZZ30_BYTES, acc);
    }
  }
  pk[2 * (size_t)t] = head_key;
  pk[2 * (size_t)t + 1] = tail_key;
  if (t == 0 && clk != nullptr) {
    clk[0] = (unsigned long long)(clock64() - clk_c0);
    clk[1] = (unsigned long long)(wall_clock64() - clk_w0);
  }
}

#ifdef GM_EXPERIMENTS
#include "msm_levels.inc"
#endif

// ------------------------------------------------------------------------------------------
// merge levels: one wave reduces 128 keyed slots (sorted by key, holes allowed) to <= 2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_merge(const uint32_t* __restrict__ keys_in, const uint8_t* __restrict__ pts_in,
                                              uint32_t E, uint32_t* __restrict__ keys_out,
                                              uint8_t* __restrict__ pts_out, uint8_t* __restrict__ buckets,
                                              int final_level) {
  // every record -- partials of all levels, buckets -- is a 208-byte loose XYZZ record (g1.cuh: Acc30); additions are
  // the asm statement of gen_madd30.py (acc30_add)
  __shared__ __attribute__((aligned(16))) uint8_t lds[128 * XYZZ30_BYTES];
  const int lane = threadIdx.x;
  const uint32_t wave = blockIdx.x;
  const uint32_t s0 = wave * 128u + 2u * lane;
  uint32_t hk = s0 < E ? keys_in[s0] : KEY_INV;
  uint32_t tk = s0 + 1 < E ? keys_in[s0 + 1] : KEY_INV;
  uint32_t hs = 2 * lane, ts = 2 * lane + 1;  // LDS slot ids
  auto copy_rec = [](void* dst, const void* src) {
    const gm_u4v* s = reinterpret_cast<const gm_u4v*>(src);
    gm_u4v* d = reinterpret_cast<gm_u4v*>(dst);
#pragma unroll
    for (int i = 0; i < 13; i++) d[i] = s[i];
  };
  auto add_slots = [&](uint32_t dst_slot, uint32_t src_slot, Acc30& a) {  // a = lds[dst] + lds[src]
    Acc30 o;
    acc30_load(a, lds + dst_slot * XYZZ30_BYTES);
    acc30_load(o, lds + src_slot * XYZZ30_BYTES);
    acc30_add(a, o);
  };
  if (hk != KEY_INV) copy_rec(lds + hs * XYZZ30_BYTES, pts_in + (size_t)s0 * XYZZ30_BYTES);
  if (tk != KEY_INV) copy_rec(lds + ts * XYZZ30_BYTES, pts_in + ((size_t)s0 + 1) * XYZZ30_BYTES);
  // a producer emits (head, INV) or (head, tail) or (INV, INV); normalise (INV, tail) defensively
  if (hk == KEY_INV && tk != KEY_INV) {
    hk = tk;
    hs = ts;
    tk = KEY_INV;
  }
  // equal head/tail keys cannot be produced, but merging them keeps the invariant "hk != tk"
  __syncthreads();
  if (hk != KEY_INV && hk == tk) {
    Acc30 a;
    add_slots(hs, ts, a);
    acc30_store(lds + hs * XYZZ30_BYTES, a);
    tk = KEY_INV;
  }
  __syncthreads();

  // step 0: every lane boundary at once.  In the common case each producer's last run continues
  // into the next producer's first run and ends there; when both neighbours hold two runs the
  // merged run is bounded on both sides and therefore a complete bucket.  This removes the
  // boundary additions from all six tree steps below (they then only carry keys).
  {
    const uint32_t nhk = __shfl_down(hk, 1), nhs = __shfl_down(hs, 1), ntk = __shfl_down(tk, 1);
    const uint32_t ptk = __shfl_up(tk, 1);
    const bool give = lane < 63 && tk != KEY_INV && ntk != KEY_INV && nhk == tk;   // my tail + next head
    const bool taken = lane > 0 && tk != KEY_INV && ptk != KEY_INV && ptk == hk;   // my head consumed by prev
    if (__any(give)) {
      if (give) {
        Acc30 a;
        add_slots(ts, nhs, a);
        acc30_store(buckets + (size_t)tk * XYZZ30_BYTES, a);
      }
    }
    __syncthreads();
    if (taken) {  // drop my head: (head, tail) -> (tail, INV)
      hk = tk; hs = ts; tk = KEY_INV;
    }
    if (give) {   // drop my tail (after `taken` so a lane that does both ends up empty)
      if (taken) { hk = KEY_INV; } else { tk = KEY_INV; }
    }
  }

  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t bhk = __shfl_down(hk, d), bhs = __shfl_down(hs, d);
    const uint32_t btk = __shfl_down(tk, d), bts = __shfl_down(ts, d);
    const bool act = (lane & (2 * d - 1)) == 0;
    bool do_add = false;
    uint32_t add_dst = 0, add_src = 0;
    uint32_t f1k = KEY_INV, f1s = 0, f2k = KEY_INV, f2s = 0;  // runs that become interior
    if (act && bhk != KEY_INV) {
      if (hk == KEY_INV) {
        hk = bhk; hs = bhs; tk = btk; ts = bts;
      } else {
        const bool a_two = tk != KEY_INV, b_two = btk != KEY_INV;
        const uint32_t alk = a_two ? tk : hk, als = a_two ? ts : hs;
        if (alk == bhk) {
          do_add = true; add_dst = als; add_src = bhs;
          if (b_two) {
            if (a_two) { f1k = alk; f1s = als; }
            tk = btk; ts = bts;
          }
        } else {
          if (a_two) { f1k = tk; f1s = ts; }
          if (b_two) { f2k = bhk; f2s = bhs; tk = btk; ts = bts; }
          else { tk = bhk; ts = bhs; }
        }
      }
    }
    if (__any(do_add)) {
      if (do_add) {
        Acc30 a;
        add_slots(add_dst, add_src, a);
        acc30_store(lds + add_dst * XYZZ30_BYTES, a);
      }
    }
    if (f1k != KEY_INV) copy_rec(buckets + (size_t)f1k * XYZZ30_BYTES, lds + f1s * XYZZ30_BYTES);
    if (f2k != KEY_INV) copy_rec(buckets + (size_t)f2k * XYZZ30_BYTES, lds + f2s * XYZZ30_BYTES);
    __syncthreads();
  }
  if (lane == 0) {
    if (final_level) {
      if (hk != KEY_INV) copy_rec(buckets + (size_t)hk * XYZZ30_BYTES, lds + hs * XYZZ30_BYTES);
      if (tk != KEY_INV) copy_rec(buckets + (size_t)tk * XYZZ30_BYTES, lds + ts * XYZZ30_BYTES);
    } else {
      keys_out[2 * (size_t)wave] = hk;
      keys_out[2 * (size_t)wave + 1] = tk;
      if (hk != KEY_INV) copy_rec(pts_out + (size_t)(2 * (size_t)wave) * XYZZ30_BYTES, lds + hs * XYZZ30_BYTES);
      if (tk != KEY_INV) copy_rec(pts_out + (size_t)(2 * (size_t)wave + 1) * XYZZ30_BYTES, lds + ts * XYZZ30_BYTES);
    }
  }
}

// ------------------------------------------------------------------------------------------
// bucket reduction by plain sums, several lanes per output element
//
// sum_b b*B_b over one bucket set = sum_j 2^j Z_j + Tot, Z_j = sum of the buckets whose index
// (|digit| - 1) has bit j set.  The index is split into up to three digit fields of <= 7 bits; the
// per-field marginals Y_k[v] = sum of buckets whose k-th field equals v are produced by strided
// plain sums (one or two levels), and the Z_j are bit-plane sums over the small Y_k arrays.  Every
// sum is over <= 128 elements, done by 2^lpo_shift lanes per output (sequential part + shuffle
// tree), so the dependent-addition depth is ~20 instead of 2*2^(c-1) for the CPU's running sum
// (variable_base.rs:140-166), at 2-3x its work.
// ------------------------------------------------------------------------------------------
enum { GS_STRIDED = 0, GS_PLANE = 1 };
struct GroupSumArgs {
  const uint8_t* in;
  uint8_t* out;
  int mode;
  uint32_t n_out;       // number of outputs (all bucket sets)
  uint32_t per_win;     // outputs per bucket set
  uint32_t win_stride;  // input elements per bucket set
  // GS_STRIDED: out[w][r] = sum_{e < len} in[w*win_stride + (r / n_lo)*s_hi + (r % n_lo)*s_lo + e*s_e]
  uint32_t n_lo, s_hi, s_lo, s_e, len;
  // GS_PLANE: input length 2^nb per set; r < nb: elements with bit r set; r == nb: elements with bit 0 clear (total = that + plane 0)
  uint32_t nb;
  uint32_t lpo_shift;   // lanes per output = 2^lpo_shift
  int out_canonical;    // write canonical 192-byte records (the plane sums the host reads) instead of 208-byte loose ones
  const uint32_t* in_cnt;  // input = the bucket array: entries per bucket; a bucket without entries was never written and counts as the identity
};
struct GroupSumJobs {
  GroupSumArgs j[3];
  uint32_t blk_end[3];  // cumulative block counts
  const uint32_t* err_src;  // last launch of a call: forward the scalar-range flag behind the plane sums
  uint32_t* err_dst;         // ... and the clock readings of the call's k_acc0 (two 64-bit words at err_dst + 2)
  const unsigned long long* clk_src;
};

// up to three independent jobs per launch so that passes of the same level overlap instead of
// serialising their (latency-bound) depth.  Inputs and intermediate arrays are 208-byte loose records, sums run on the asm
// addition (acc30_add); only the last launch of a call (out_canonical) writes the canonical 192-byte records the host reads.
__global__ __launch_bounds__(256) void k_group_sum(GroupSumJobs J) {
  if (J.err_dst && blockIdx.x == 0 && threadIdx.x == 0) {
    *J.err_dst = *J.err_src;
    if (J.clk_src) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(J.err_dst + 2);
      d[0] = J.clk_src[0];
      d[1] = J.clk_src[1];
    }
  }
  const int job = blockIdx.x < J.blk_end[0] ? 0 : (blockIdx.x < J.blk_end[1] ? 1 : 2);
  const GroupSumArgs& a = J.j[job];
  const uint32_t blk0 = job == 0 ? 0u : J.blk_end[job - 1];
  const uint8_t* __restrict__ in = a.in;
  uint8_t* __restrict__ out = a.out;
  const uint32_t gt = (blockIdx.x - blk0) * blockDim.x + threadIdx.x;
  const uint32_t lpo = 1u << a.lpo_shift;
  const uint32_t o = gt >> a.lpo_shift, q = gt & (lpo - 1u);
  Acc30 acc;
  acc30_zero(acc);
  const uint32_t* __restrict__ in_cnt = a.in_cnt;
  auto filled = [&](size_t slot) { return in_cnt == nullptr || in_cnt[slot] != 0u; };
  auto add_rec = [&](size_t slot) {
    if (!filled(slot)) return;  // an empty bucket: nothing was stored there by this call (the array is not cleared between calls)
    Acc30 v;
    acc30_load(v, in + slot * XYZZ30_BYTES);
    acc30_add(acc, v);
  };
  auto load_first = [&](size_t slot) {
    if (filled(slot)) acc30_load(acc, in + slot * XYZZ30_BYTES);
  };
  // the first element of a lane is LOADED, not added to the identity: one addition less on the dependent chain of the
  // latency-bound levels (6 -> 5 and 5 -> 4 additions of a lone wave, ~20 us each)
  if (o < a.n_out) {
    const uint32_t w = o / a.per_win, r = o % a.per_win;
    const size_t base = (size_t)w * a.win_stride;
    if (a.mode == GS_STRIDED) {
      const size_t b0 = base + (size_t)(r / a.n_lo) * a.s_hi + (size_t)(r % a.n_lo) * a.s_lo;
      if (q < a.len) load_first(b0 + (size_t)q * a.s_e);
      for (uint32_t e = q + lpo; e < a.len; e += lpo) add_rec(b0 + (size_t)e * a.s_e);
    } else {
      const uint32_t nb = a.nb;
      if (r == nb) {  // the elements whose bit 0 is CLEAR: total = this + plane 0 on the host (one addition per bucket set there);
                      // summing all 2^nb here made this output twice as deep as every other one of the launch
        const uint32_t cnt0 = nb ? (1u << (nb - 1)) : 1u;
        if (q < cnt0) load_first(base + 2u * q);
        for (uint32_t e = q + lpo; e < cnt0; e += lpo) add_rec(base + 2u * e);
      } else {        // elements whose bit r is set
        const uint32_t half = nb ? (1u << (nb - 1)) : 0u;
        auto slot_of = [&](uint32_t e) { return ((e >> r) << (r + 1)) | (1u << r) | (e & ((1u << r) - 1u)); };
        if (q < half) load_first(base + slot_of(q));
        for (uint32_t e = q + lpo; e < half; e += lpo) add_rec(base + slot_of(e));
      }
    }
  }
#pragma unroll 1
  for (int m = (int)(lpo >> 1); m >= 1; m >>= 1) {
    Acc30 other;
    acc30_shfl_xor(other, acc, m);
    acc30_add(acc, other);
  }
  if (o < a.n_out && q == 0) {
    if (a.out_canonical) g1_store_xyzz(out + (size_t)o * XYZZ_BYTES, acc30_to_canonical(acc));
    else acc30_store(out + (size_t)o * XYZZ30_BYTES, acc);
  }
}

// ------------------------------------------------------------------------------------------
// bases import / generation
// ------------------------------------------------------------------------------------------
// staging (stride >= 96, optional infinity flag at byte 96, ark-ff Montgomery form) -> packed 96-byte
// records in the device form (see g1.cuh: a * 2^390 with GM_FQ30)
__global__ void k_pack_bases(const uint8_t* __restrict__ src, size_t stride, size_t n, uint8_t* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(src + i * stride);
  bool inf = stride >= 97 && src[i * stride + 96] != 0;
  Fq x, y;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    x.l[k] = inf ? 0u : s[k];
    y.l[k] = inf ? 0u : s[12 + k];
  }
  G1Affine a;
  a.x = fqe_import(x);
  a.y = fqe_import(y);
  g1_store_affine(dst + i * AFF_BYTES, a);
}
// device form -> ark-ff Montgomery form, 96-byte records (gm_g1_bases_download)
__global__ void k_export_bases(const uint8_t* __restrict__ src, size_t n, uint8_t* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine a = g1_load_affine(src + i * AFF_BYTES);
  fp_store<FqParams>(dst + i * AFF_BYTES, fqe_export(a.x));
  fp_store<FqParams>(dst + i * AFF_BYTES + 48, fqe_export(a.y));
}

// phi(P) = (beta x, y): the image of every registered base under the GLV endomorphism (see glv_split)
__global__ void k_phi_bases(const uint8_t* __restrict__ src, size_t n, uint8_t* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine a = g1_load_affine(src + i * AFF_BYTES);
  // beta in the device's Montgomery form (a * 2^390 with the radix-2^30 product core, a * 2^384 otherwise)
#if GM_FQ30
  constexpr uint32_t BETA[12] = {0x9c907181u, 0xef2f7921u, 0xb26574c3u, 0x1bcc91d7u, 0x191c3ebcu, 0x856e7b9au,
                                 0x67fd6ffau, 0xbd16b0d2u, 0xeb0c0550u, 0x18c86532u, 0x6567dd7du, 0x09c6d485u};
#else
  constexpr uint32_t BETA[12] = {0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu,
                                 0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u};
#endif
  Fq b;
#pragma unroll
  for (int k = 0; k < 12; k++) b.l[k] = BETA[k];
#if GM_FQ30 == 1
  a.x = fq_mul(a.x, fq30_unpack(b));
#else
  a.x = fq_mul(a.x, b);
#endif
  g1_store_affine(dst + i * AFF_BYTES, a);  // the identity (0, 0) maps to itself
}

// out[i] = k_i * base via 32 windows of 8 bits against a (32 x 256)-entry affine table
__global__ __launch_bounds__(256) void k_fixed_base_table(const uint8_t* __restrict__ base, uint8_t* __restrict__ table) {
  // table[w][d] = d * 2^(8w) * base ; one thread per (w, d): double-and-add of the 13-bit... d*2^(8w)
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 32 * 256) return;
  const uint32_t w = t >> 8, dgt = t & 255u;
  G1Affine b = g1_load_affine(base);
  G1Xyzz acc = G1Xyzz::identity();
  // scalar = dgt << (8w): process bits of dgt from the top, then 8w doublings
  for (int bit = 7; bit >= 0; bit--) {
    acc = xyzz_dbl(acc);
    if ((dgt >> bit) & 1u) xyzz_madd(acc, b);
  }
  for (uint32_t k = 0; k < 8 * w; k++) acc = xyzz_dbl(acc);
  g1_store_xyzz(table + (size_t)t * XYZZ_BYTES, acc);
}

// Fq inversion by Fermat (a^(q-2)); used once per generated point
GM_DEV FqE fq_inv(const FqE& a) {
  // q - 2, little-endian 32-bit limbs
  uint32_t e[12];
#pragma unroll
  for (int i = 0; i < 12; i++) e[i] = FqParams::MOD[i];
  e[0] -= 2u;
  FqE acc = fqe_one();
  for (int i = 380; i >= 0; i--) {
    acc = fq_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) acc = fq_mul(acc, a);
  }
  return acc;
}

__global__ __launch_bounds__(256) void k_xyzz_to_affine(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Xyzz p = g1_load_xyzz(in + i * XYZZ_BYTES);
  G1Affine a;
  if (p.is_identity()) {
    a.x = fqe_zero();
    a.y = fqe_zero();
  } else {
    // x = X/ZZ, y = Y/ZZZ ; 1/ZZZ = inv, 1/ZZ = inv * ZZZ / ZZ ... use one inversion of ZZ*ZZZ
    FqE t = fq_mul(p.zz, p.zzz);
    FqE ti = fq_inv(t);
    a.x = fq_mul(p.x, fq_mul(ti, p.zzz));
    a.y = fq_mul(p.y, fq_mul(ti, p.zz));
  }
  g1_store_affine(out + i * AFF_BYTES, a);
}

// Batched normalisation: one Fermat inversion (~570 products) per NORM_K points instead of per point
// (Montgomery's trick inside a thread: prefix products of t_k = zz_k * zzz_k, one inversion, back-substitution).
// Identity records (zz = 0) are skipped in the products and written as the all-zero affine record.
constexpr int NORM_K = 8;
__global__ __launch_bounds__(256) void k_xyzz_to_affine_batch(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i0 = t * NORM_K;
  if (i0 >= n) return;
  const int m = (int)min((size_t)NORM_K, n - i0);
  FqE pre[NORM_K];
  FqE run = fqe_one();
  for (int k = 0; k < m; k++) {
    const uint8_t* rec = in + (i0 + k) * XYZZ_BYTES;
    const FqE zz = fqe_load(rec + 96);
    if (!fq_is_exact_zero(zz)) run = fq_mul(run, fq_mul(zz, fqe_load(rec + 144)));
    pre[k] = run;
  }
  FqE inv = fq_inv(run);
  for (int k = m - 1; k >= 0; k--) {
    const uint8_t* rec = in + (i0 + k) * XYZZ_BYTES;
    const FqE zz = fqe_load(rec + 96);
    G1Affine a;
    if (fq_is_exact_zero(zz)) {
      a.x = fqe_zero();
      a.y = fqe_zero();
    } else {
      const FqE zzz = fqe_load(rec + 144);
      const FqE ti = k ? fq_mul(inv, pre[k - 1]) : inv;  // 1 / (zz * zzz)
      inv = fq_mul(inv, fq_mul(zz, zzz));
      a.x = fq_mul(fqe_load(rec), fq_mul(ti, zzz));
      a.y = fq_mul(fqe_load(rec + 48), fq_mul(ti, zz));
    }
    g1_store_affine(out + (i0 + k) * AFF_BYTES, a);
  }
}

// scalars canonical (mont = 0) or Montgomery; table in XYZZ; output affine
__global__ __launch_bounds__(256) void k_fixed_base_mul(const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                        const uint8_t* __restrict__ table_aff, uint8_t* __restrict__ out_xyzz) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = fp_load<FrParams>(scalars + 8 * i);
  if (mont) s = fp_from_mont<FrParams>(s);
  G1Xyzz acc = G1Xyzz::identity();
  for (int w = 0; w < 32; w++) {
    uint32_t dgt = (s.l[w >> 2] >> (8 * (w & 3))) & 255u;
    if (dgt) {
      G1Affine p = g1_load_affine(table_aff + ((size_t)w * 256 + dgt) * AFF_BYTES);
      xyzz_madd(acc, p);
    }
  }
  g1_store_xyzz(out_xyzz + i * XYZZ_BYTES, acc);  // normalised in batches by k_xyzz_to_affine_batch
}

// herring split_fold over G1 (src/herring/time_prover.rs:72-76): out[i] = P[2i] + s * P[2i+1], affine out.
// s: canonical scalar, 8 x u32.  One thread per output: double-and-add, then one inversion.
__global__ __launch_bounds__(256) void k_g1_split_fold(const uint8_t* __restrict__ in, size_t n, const uint32_t* __restrict__ s8,
                                                       uint8_t* __restrict__ out) {
  const size_t m = (n + 1) / 2;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  G1Affine lo = g1_load_affine(in + (2 * i) * AFF_BYTES);
  G1Xyzz acc = G1Xyzz::identity();
  if (2 * i + 1 < n) {
    G1Affine hi = g1_load_affine(in + (2 * i + 1) * AFF_BYTES);
    for (int bit = 254; bit >= 0; bit--) {
      acc = xyzz_dbl(acc);
      if ((s8[bit >> 5] >> (bit & 31)) & 1u) xyzz_madd(acc, hi);
    }
  }
  xyzz_madd(acc, lo);
  G1Affine a;
  if (acc.is_identity()) {
    a.x = fqe_zero();
    a.y = fqe_zero();
  } else {
    FqE t = fq_mul(acc.zz, acc.zzz);
    FqE ti = fq_inv(t);
    a.x = fq_mul(acc.x, fq_mul(ti, acc.zzz));
    a.y = fq_mul(acc.y, fq_mul(ti, acc.zz));
  }
  g1_store_affine(out + i * AFF_BYTES, a);
}

// table[(w + 1) * n + i] = 2^c * table[w * n + i]: c doublings and one normalisation per point
__global__ __launch_bounds__(256) void k_table_next(const uint8_t* __restrict__ prev, uint8_t* __restrict__ next_xyzz, size_t n, int c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = g1_load_affine(prev + i * AFF_BYTES);
  G1Xyzz acc = G1Xyzz::from_affine(p);
  for (int k = 0; k < c; k++) acc = xyzz_dbl(acc);
  g1_store_xyzz(next_xyzz + i * XYZZ_BYTES, acc);  // normalised in batches by k_xyzz_to_affine_batch
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int ceil_log2_sz(size_t n) {
  int b = 0;
  size_t v = n > 1 ? n - 1 : 0;
  while (v) {
    b++;
    v >>= 1;
  }
  return b;
}

// window width: balances n*W mixed additions against W*2^(c-1) bucket work; tuned on MI355X
// (sweep of 2^10 .. 2^19 pairs, c = 8 .. 16: with c = 16 there are exactly 16 windows and sparse buckets,
// so chunk boundaries rarely split a bucket and k_merge has little to do -- from 2^15 pairs on that beats
// the smaller bucket count of a narrower window, e.g. 3.52 vs 3.96 ms at 2^19, 1.61 vs 1.84 ms at 2^16)
static int choose_window(size_t n) {
  int lg = ceil_log2_sz(n);
  // 13 windows: another -4 % at 2^25 and 2^26, -2 % at 2^24 (43.2 vs 44.1 ms); at 2^23 the 6.8 M buckets still cost more
  // than the saved window (25.2 vs 23.7 ms), so the switch sits at 7 * 2^21 pairs
  static const size_t c20_min = getenv("GM_MSM_C20_MIN") ? (size_t)strtoull(getenv("GM_MSM_C20_MIN"), nullptr, 10) : ((size_t)7 << 21);  // tuning override
  if (n >= c20_min) return 20;
  if (lg >= 23) return 19;  // 14 windows: -2.5 % at 2^23, -7 % at 2^24, -11 % at 2^26 against c = 16 (the 3.7 M buckets cost 2.7 ms to reduce)
  // (GM_MSM_C_MEDIUM: A/B knob for 2^14 .. 2^17 pairs, the calls that share the GPU with the big ones inside a batch)
  static const int c_medium = getenv("GM_MSM_C_MEDIUM") ? atoi(getenv("GM_MSM_C_MEDIUM")) : 0;
  if (c_medium && lg >= 14 && lg <= 17) return c_medium;
  if (lg >= 14) return 16;
  // small calls, re-tuned with the flat digit kernels (tools/tune_small.py, round 2): the launch chain and the
  // merge depth dominate, so sparse buckets (c = 8: <= 2 lanes per bucket at L = 4) win from 2^11 pairs on --
  // 0.72 vs 0.97 ms (c = 13) at 2^13, 0.89 vs 1.07 ms at 2^14 with c = 16
  if (lg >= 11) return 8;
  if (lg >= 8) return lg - 4;
  if (lg >= 5) return lg + 1;
  return 4;
}

// ---- the footprint contract (capi.hip: gm_snark_footprint / gm_psnark_footprint) ------------------------------------------------
// What a workspace holds: every grow-only buffer of it.
size_t msm_workspace_held(const MsmWorkspace& ws) {
  size_t b = 0;
  for (const DevBuf* d : {&ws.scalars, &ws.counts, &ws.offsets, &ws.cursor, &ws.entries, &ws.tmp_entries, &ws.sortmeta, &ws.buckets, &ws.pk[0], &ws.pk[1],
                          &ws.pp[0], &ws.pp[1], &ws.rows, &ws.cols, &ws.planes, &ws.misc, &ws.clk})
    b += d->cap;
  return b;
}
// Upper bound of what a workspace holds after ONE call of n pairs walking a prefix of `bases` (the calls of a prover): the same
// window / table choice as msm_enqueue, its ensure() sizes, DevBuf's 12.5 % growth.  Entries and their sort double dominate:
// 2 x 8 bytes x windows x pairs (14 GB at 2^26 pairs).
size_t msm_workspace_bound(Context* C, const Bases* bases, size_t n) {
  if (n == 0) return 0;
  n = std::min<size_t>(n, (size_t)1 << 26);  // msm_run cuts longer calls
  int c = C->msm_c_override ? C->msm_c_override : choose_window(n);
  bool table = false;
  if (bases && !C->msm_c_override && n < ((size_t)1 << ENTRY_W_SHIFT)) {
    if (bases->table && n >= std::max(C->msm_table_min, bases->tab_min)) {
      c = bases->tab_c;
      table = true;
    } else {
      for (const Bases::TableSet& ts : bases->extra)
        if (n >= ts.min_n && n < ts.max_n && n <= ts.n) {
          c = ts.c;
          table = true;
          break;
        }
    }
  }
  const size_t W = (size_t)(256 + c - 1) / (size_t)c;
  const size_t N = n * W, nbuckets = (table ? 1 : W) << (c - 1);
  auto grow = [](size_t b) { return b + b / 8 + 256; };
  const size_t lanes0 = N >= ((size_t)1 << 24) ? 262144 : 131072;
  const size_t L = std::min<size_t>(256, std::max<size_t>(4, (N + lanes0 - 1) / lanes0)) + 1;
  const size_t E1 = 2 * ((N / L + 256 + 255) / 256 * 256), E2 = 2 * ((E1 + 127) / 128);
  size_t b = 2 * grow(N * 8);                                  // entries, tmp_entries
  b += grow(nbuckets * XYZZ30_BYTES) + 4 * grow((nbuckets + 8192) * 4);  // buckets; counts, offsets, cursor, misc
  b += grow(E1 * 4) + grow(E1 * XYZZ30_BYTES) + grow(E2 * 4) + grow(E2 * XYZZ30_BYTES);  // keyed partials of the two merge levels
  b += (size_t)W * ((size_t)3 << 14) * XYZZ30_BYTES + ((size_t)16 << 20);                // row / column sums of the bucket reduction, planes, small buffers
  return b;
}

constexpr size_t MSM_SMALL_N = (size_t)1 << 17;
constexpr size_t MSM_SPLIT_MIN_N = (size_t)1 << 17;  // smaller calls are launch-latency bound: one chain of launches beats two
static int msm_run_one(Context* C, const Bases* bases, int64_t first, int64_t step, const void* d_scalars, int mont, size_t n,
                       bool normalize, uint64_t out_jac[18]);
static int msm_run_batch_at(Context* C, const Bases* bases, int64_t first, int64_t step, const size_t* pair_offsets, const void* const* d_scalars,
                            int mont, const size_t* ns, size_t k, bool normalize, uint64_t* out_jac, const int64_t* firsts = nullptr);

// One MSM is an enqueue (every kernel + the async copy of the window bit-planes to pinned memory, no host
// wait) and a finish (wait for the copy, Horner over the bit positions on the host).  Splitting them lets
// a batch of MSMs overlap the host tail of call i with the kernels of call i+1 (msm_run_batch).
// (struct MsmPending: ctx.hpp)
// A call may be enqueued in `nparts` window groups (part p owns windows [p W / nparts, (p + 1) W / nparts)), each with
// its own workspace; `sts` names the streams of its three phases -- sort, accumulate, merge + reduce + copy-out.
struct MsmStreams {
  hipStream_t sort, acc, tail;
};
// several small calls as the levels of ONE pass (k_digits_multi): level l pairs scalars[l][i], i < n[l], with base start[l] + step i
struct MsmMulti {
  // block = false: tiny calls (<= 2^13 pairs), c = 8, the flat three-launch sort; block = true: mid-size calls (up to 2^20 pairs each)
  // through the block sort, c = 16 or the key's c <= 20 tables -- one sort / accumulation / merge / reduction chain for all of them
  bool block = false;
  int levels = 0;
  const void* scalars[MULTI_MAX_LEVELS];
  size_t n[MULTI_MAX_LEVELS];
  int64_t start[MULTI_MAX_LEVELS];
};
constexpr size_t MSM_MULTI_MAX_N = (size_t)1 << 13;  // calls this small (c = 8 on their own as well) are fused
constexpr size_t MSM_MULTI_BLOCK_MAX_N = (size_t)1 << 20;  // and the mid-size ones above them, up to this, as the levels of a block-sorted pass
static int msm_enqueue(Context* C, MsmWorkspace& ws, MsmStreams sts, const Bases* bases, int64_t first, int64_t step, const void* d_scalars,
                       int mont, size_t n, int slot, MsmPending* P, int part = 0, int nparts = 1, const MsmMulti* multi = nullptr);
static int msm_finish_parts(Context* C, const MsmPending* parts, int nparts, bool normalize, uint64_t out_jac[18], bool use_pool = true,
                            uint64_t* const* multi_out = nullptr);
// the helper threads of the host tail (class HornerPool, below)
static void horner_pool_prewake();
static void horner_pool_run(int n, const std::function<void(int)>& fn);
static int msm_finish(Context* C, const MsmPending& P, bool normalize, uint64_t out_jac[18], bool use_pool = true) {
  return msm_finish_parts(C, &P, 1, normalize, out_jac, use_pool);
}

// Calls larger than 2^26 pairs are split into 2^26-pair MSMs whose results are added on the host --
// the same composition ChunkedPippenger / msm_chunks use (src/kzg/space.rs:41-53), with the chunk
// sized for the device (entry indices are 26 + 5 bits; n * windows must stay below 2^32).
int msm_run(Context* C, const Bases* bases, int64_t first, int64_t step, const void* d_scalars, int mont, size_t n,
            bool normalize, uint64_t out_jac[18]) {
  const size_t CH = (size_t)1 << 26;
  // (running one large call as two half-size calls on the two lanes was measured and is slower: 4.95 vs
  // 4.59 ms at 2^20 -- the doubled reduce and host tails are not hidden)
  if (n <= CH) return msm_run_one(C, bases, first, step, d_scalars, mont, n, normalize, out_jac);
  gmh::G1 acc = gmh::G1::identity();
  for (size_t off = 0; off < n; off += CH) {
    const size_t m = n - off < CH ? n - off : CH;
    uint64_t part[18];
    int rc = msm_run_one(C, bases, first + step * (int64_t)off, step, reinterpret_cast<const uint8_t*>(d_scalars) + off * 32, mont, m, false, part);
    if (rc) return rc;
    acc = acc.add(gmh::G1::from_limbs(part));
  }
  if (normalize) acc = acc.normalized();
  acc.to_limbs(out_jac);
  return GM_OK;
}

static int msm_run_one(Context* C, const Bases* bases, int64_t first, int64_t step, const void* d_scalars, int mont, size_t n,
                       bool normalize, uint64_t out_jac[18]) {
  GM_MSM_LOCK(C);
  // Optional (gm_set_msm_split, off by default): one large call as two window groups, high windows first, a
  // software pipeline over three streams --
  //   sort stream   sort(H)  sort(L)
  //   main stream            acc0(H)            acc0(L)
  //   tail streams                     merge + reduce + copy-out (H) | (L)
  //   host                                                  Horner(H)      Horner(L)
  // -- so that the sort and the merge / reduce of one group could run in the shadow of the accumulation of the
  // other (the reduce work is per window: nothing is duplicated, unlike a split by pairs).  Measured on MI355X at
  // 2^20 pairs: 4.18 ms against 4.05 ms unsplit.  k_acc0 fills every CU with two 196-register waves per SIMD for
  // its whole duration, and k_merge / k_group_sum (216 registers) cannot become resident next to them, so the
  // tail of the first group still runs after the second accumulation, twice as many latency-bound launches as
  // before.  Results are identical (tests/test_gpu_msm.py::test_msm_window_group_split).
  const bool tables = bases->table != nullptr && !C->msm_c_override && n >= std::max(C->msm_table_min, bases->tab_min) &&
                      n < ((size_t)1 << ENTRY_W_SHIFT);
  const bool split = GM_GLV(1) && C->msm_split && n >= MSM_SPLIT_MIN_N && !tables && C->msm_affine_levels == 0 && C->stream_b && C->small_stream[0];
  if (!split) {
    MsmPending P;
    int rc = msm_enqueue(C, C->msm, MsmStreams{C->stream, C->stream, C->stream}, bases, first, step, d_scalars, mont, n, 0, &P);
    if (rc) return rc;
    return msm_finish(C, P, normalize, out_jac);
  }
  if (!C->have_start_ev) {
    GM_HIP(hipEventCreateWithFlags(&C->start_ev, hipEventDisableTiming));
    C->have_start_ev = true;
  }
  // the scalars may have been produced by earlier work on the main stream
  GM_HIP(hipEventRecord(C->start_ev, C->stream));
  GM_HIP(hipStreamWaitEvent(C->small_stream[0], C->start_ev, 0));
  MsmPending P[2];
  const MsmStreams sts{C->small_stream[0], C->stream, C->stream_b};
  const MsmStreams sts_lo{C->small_stream[0], C->stream, C->small_stream[1]};  // the two tails end up side by side
  int rc = msm_enqueue(C, C->msm, sts, bases, first, step, d_scalars, mont, n, 0, &P[1], 1, 2);  // high windows first
  if (!rc) rc = msm_enqueue(C, C->msm_b, sts_lo, bases, first, step, d_scalars, mont, n, 0, &P[0], 0, 2);
  if (rc) {
    (void)hipStreamSynchronize(C->small_stream[0]);
    (void)hipStreamSynchronize(C->stream);
    (void)hipStreamSynchronize(C->stream_b);
    (void)hipStreamSynchronize(C->small_stream[1]);
    return rc;
  }
  return msm_finish_parts(C, P, 2, normalize, out_jac);
}

// k MSMs against the same registered bases.  Results are identical to k msm_run calls; what changes is
// the schedule: (i) the kernels of a call are enqueued before the host finishes the previous one (two
// pinned result buffers on the main workspace), so the host Horner runs under the next call's kernels;
// (ii) small calls (<= MSM_SMALL_N pairs) are latency-bound chains of a dozen tiny launches, so they go
// round-robin to MSM_SMALL_LANES extra workspaces with their own streams and run side by side -- the
// folding commitments of the tensor check are ~20 MSMs of sizes n/2, n/4, ..., 1.
int msm_run_batch(Context* C, const Bases* bases, int64_t first, int64_t step, const void* const* d_scalars, int mont, const size_t* ns,
                  size_t k, bool normalize, uint64_t* out_jac) {
  return msm_run_batch_at(C, bases, first, step, nullptr, d_scalars, mont, ns, k, normalize, out_jac);
}
int msm_run_batch_offsets(Context* C, const Bases* bases, const size_t* pair_offsets, int64_t step, const void* const* d_scalars, int mont,
                          const size_t* ns, size_t k, bool normalize, uint64_t* out_jac) {
  return msm_run_batch_at(C, bases, 0, step, pair_offsets, d_scalars, mont, ns, k, normalize, out_jac);
}
// pair_offsets[j] (optional): call j starts at base pair_offsets[j] (absolute, whatever the step) and walks step from there
// firsts[j] (optional): call j starts at base firsts[j] instead of `first` (herring: even / odd halves of one array)
static int msm_run_batch_at(Context* C, const Bases* bases, int64_t first, int64_t step, const size_t* pair_offsets, const void* const* d_scalars,
                            int mont, const size_t* ns, size_t k, bool normalize, uint64_t* out_jac, const int64_t* firsts) {
  const size_t CH = (size_t)1 << 26;
  bool pipelined = !C->prof.on;
  for (size_t j = 0; j < k; j++) pipelined = pipelined && ns[j] <= CH;
  if (!pipelined) {
    for (size_t j = 0; j < k; j++) {
      int rc = msm_run(C, bases, (pair_offsets ? (int64_t)pair_offsets[j] : (firsts ? firsts[j] : first)), step, d_scalars[j], mont, ns[j], normalize,
                       out_jac + 18 * j);
      if (rc) return rc;
    }
    return GM_OK;
  }
  GM_MSM_LOCK(C);
  if (!C->have_start_ev) {
    GM_HIP(hipEventCreateWithFlags(&C->start_ev, hipEventDisableTiming));
    C->have_start_ev = true;
  }
  GM_HIP(hipEventRecord(C->start_ev, C->stream));
  struct Inflight {
    size_t j;
    int lane;  // 0 = main workspace, 1.. = small workspaces
    MsmPending P;
    std::vector<size_t> fused;  // the calls of a fused pass (MsmMulti), in level order; empty: the one call j
  };
  auto finish_entry = [&](Inflight& e, bool use_pool) -> int {
    if (e.fused.empty()) return msm_finish(C, e.P, normalize, out_jac + 18 * e.j, use_pool);
    std::vector<uint64_t*> outs(e.fused.size());
    for (size_t l = 0; l < e.fused.size(); l++) outs[l] = out_jac + 18 * e.fused[l];
    return msm_finish_parts(C, &e.P, 1, normalize, nullptr, use_pool, outs.data());
  };
  std::vector<Inflight> q;  // calls enqueued so far, in order; at most 4 big + MSM_SMALL_LANES small ones are unfinished at a time
  std::vector<char> finished;
  auto fail = [&](int rc) {
    (void)hipStreamSynchronize(C->stream);
    for (int s = 0; s < MSM_SMALL_LANES; s++)
      if (C->small_stream[s]) (void)hipStreamSynchronize(C->small_stream[s]);
    if (C->stream_b) (void)hipStreamSynchronize(C->stream_b);
    for (int s = 0; s < 2 + MSM_SMALL_LANES && C->cu_split; s++) {
      (void)hipStreamSynchronize(C->part_acc[s]);
      (void)hipStreamSynchronize(C->part_tail[s]);
    }
    return rc;
  };
  auto busy = [&](int lane, int hslot) {
    for (size_t t = 0; t < q.size(); t++)
      if (!finished[t] && q[t].lane == lane && q[t].P.slot == hslot) return true;
    return false;
  };
  // The host tail of a small call is ~0.2 ms, most of it the 256 dependent doublings of the window Horner.  Every small call in flight
  // is finished at once, one per helper thread (each tail serial on its thread), instead of one after the other on the calling
  // thread.  Measured NEUTRAL (GM_MSM_PAR_FINISH=0 / 1: snark -i 20 20.1 / 19.8 ms, -i 22 48.2 / 48.1): a batch is bound by how its
  // kernels share the GPU, not by the host (enqueueing twenty calls takes 0.7 ms of 9.3, GM_MSM_TRACE=1).
  auto finish_smalls = [&]() -> int {
    std::vector<size_t> idx;
    for (size_t t = 0; t < q.size(); t++)
      if (!finished[t] && q[t].lane > 0) idx.push_back(t);
    if (idx.empty()) return GM_OK;
    std::vector<int> rcs(idx.size(), GM_OK);
    // a fused pass has hundreds of bucket sets: its tail is pool work of its own (window sums, then one Horner per call)
    {
      std::vector<size_t> rest;
      int rc_f = GM_OK;
      for (size_t t : idx) {
        if (q[t].fused.empty()) {
          rest.push_back(t);
          continue;
        }
        const int r1 = finish_entry(q[t], true);
        finished[t] = 1;
        if (r1 && !rc_f) rc_f = r1;
      }
      if (rc_f) return rc_f;
      idx.swap(rest);
      if (idx.empty()) return GM_OK;
      rcs.assign(idx.size(), GM_OK);
    }
    static const bool par_finish = !(getenv("GM_MSM_PAR_FINISH") && !strcmp(getenv("GM_MSM_PAR_FINISH"), "0"));  // A/B knob
    if (!par_finish) {
      for (size_t i = 0; i < idx.size(); i++) rcs[i] = finish_entry(q[idx[i]], true);
    } else if (idx.size() == 1) {
      rcs[0] = finish_entry(q[idx[0]], true);
    } else {
      horner_pool_prewake();
      horner_pool_run((int)idx.size(), [&](int i) { rcs[(size_t)i] = finish_entry(q[idx[(size_t)i]], false); });
    }
    int rc = GM_OK;
    for (size_t i = 0; i < idx.size(); i++) {
      finished[idx[i]] = 1;
      if (rcs[i] && !rc) rc = rcs[i];
    }
    return rc;
  };
  auto drain_oldest = [&]() -> int {
    for (size_t t = 0; t < q.size(); t++) {
      if (finished[t]) continue;
      if (q[t].lane > 0) return finish_smalls();
      finished[t] = 1;
      return finish_entry(q[t], true);
    }
    return GM_OK;
  };
  // free a (workspace, result buffer) pair: finish the call that holds IT -- not the oldest call of the batch, which may be a big
  // one that still has milliseconds of accumulation ahead while the small lanes have long been idle
  auto drain_pair = [&](int lane, int hslot) -> int {
    for (size_t t = 0; t < q.size(); t++) {
      if (finished[t] || q[t].lane != lane || q[t].P.slot != hslot) continue;
      if (lane > 0) return finish_smalls();
      finished[t] = 1;
      return finish_entry(q[t], true);
    }
    return GM_OK;
  };
  static const bool batch_trace = getenv("GM_MSM_TRACE") != nullptr;  // host time of the batch: enqueueing vs finishing
  double enqueue_s = 0.0;
  const auto t_batch0 = std::chrono::steady_clock::now();
  static const int small_slots = std::max(1, std::min(MSM_SLOTS, getenv("GM_MSM_SMALL_SLOTS") ? atoi(getenv("GM_MSM_SMALL_SLOTS")) : MSM_SLOTS));
  static const int small_lanes = std::max(1, std::min(MSM_SMALL_LANES, getenv("GM_MSM_SMALL_LANES") ? atoi(getenv("GM_MSM_SMALL_LANES")) : MSM_SMALL_LANES));
  int big_rr = 0, small_rr = 0;
  auto is_small = [&](size_t j) { return ns[j] <= MSM_SMALL_N && C->small_stream[0] != nullptr && C->msm_affine_levels <= 0; };
  // ORDER.  Small calls do not progress beside the accumulation of a big one: the SIMD arbiter serves the oldest waves first
  // (section 4.1) and a k_acc0 grid keeps every SIMD supplied with older waves, so a small call's kernels crawl (a memset of
  // a few KB was seen to take 0.7 ms there) and its XYZZ + XYZZ kernels (214 VGPRs) do not even fit beside two accumulation waves.
  // So the small calls of a batch are ENQUEUED FIRST: they get their sorts and accumulations in while the first big call is still
  // sorting, instead of queueing up behind three accumulations (snark -i 20 15.8 -> 15.2 ms, -i 24 119.3 -> 118.6, the sharded share
  // 29.8 -> 29.1, psnark -i 22 369 -> 368: two A/B runs each).  Holding the big calls back until the small ones are DONE
  // (GM_MSM_BATCH_ORDER=smalls) is worse than either (10.3 against 9.8 ms for the batch 2^20 .. 2); GM_MSM_BATCH_ORDER=given keeps the
  // caller's order.
  static const bool order_given = getenv("GM_MSM_BATCH_ORDER") && !strcmp(getenv("GM_MSM_BATCH_ORDER"), "given");
  static const bool order_gated = getenv("GM_MSM_BATCH_ORDER") && !strcmp(getenv("GM_MSM_BATCH_ORDER"), "smalls");
  const bool smalls_first = !order_given, smalls_nogate = !order_gated;
  std::vector<size_t> order;
  if (smalls_first) {
    for (size_t j = 0; j < k; j++)
      if (is_small(j)) order.push_back(j);
    for (size_t j = 0; j < k; j++)
      if (!is_small(j)) order.push_back(j);
  } else {
    for (size_t j = 0; j < k; j++) order.push_back(j);
  }
  // the tiny calls of the batch (<= 2^13 pairs: the tail of every folding tree) become the levels of ONE fused pass (MsmMulti): a dozen
  // latency-bound chains of ~15 launches collapse into one.  It is enqueued where the first of them stood.  GM_MSM_FUSE=0 keeps them apart.
  static const bool fuse_env = !(getenv("GM_MSM_FUSE") && !strcmp(getenv("GM_MSM_FUSE"), "0"));
  std::vector<std::vector<size_t>> fused_groups;  // each: <= MULTI_MAX_LEVELS tiny calls, in batch order (a preprocessing proof has ~50)
  std::vector<int> group_of(k, -1);
  if (fuse_env && C->small_stream[0] != nullptr && C->msm_affine_levels <= 0 && !C->msm_c_override) {
    std::vector<size_t> tiny;
    for (size_t j = 0; j < k; j++)
      if (ns[j] >= 1 && ns[j] <= MSM_MULTI_MAX_N) tiny.push_back(j);
    if (tiny.size() >= 2)
      for (size_t at = 0; at < tiny.size(); at += MULTI_MAX_LEVELS) {
        std::vector<size_t> grp(tiny.begin() + at, tiny.begin() + std::min(tiny.size(), at + MULTI_MAX_LEVELS));
        if (grp.size() < 2) break;  // a lone leftover runs on its own
        for (size_t j : grp) group_of[j] = (int)fused_groups.size();
        fused_groups.push_back(std::move(grp));
      }
  }
  // ... and the MID-SIZE calls (2^13 < n <= 2^20: the middle of a folding tree, seven latency-bound chains whose sorts, merges and bucket
  // reductions do not hide under each other's accumulations -- profiles/r5_batch_timeline.txt) CAN run as the levels of ONE
  // block-sorted pass: one sort, one accumulation that fills the GPU, one merge, one reduction over (level, window) bucket sets.
  // MEASURED NEUTRAL, hence off by default (GM_MSM_FUSE_MID=1 turns it on; read per call so that a test can): the batch 2^20 .. 2
  // takes 8.25 ms fused against 8.09-8.23 apart -- the fused pass is one serial chain (sort 1.0, accumulation 4.6, reduction of
  // 7 x 2^19 buckets 1.7, host tail 0.6 ms) where the separate calls at least overlap their tails a little; without tables
  // (c = 13 .. 16) 9.2-9.4 ms.  What an MSM costs beyond its accumulation is work, not launch count (profiles/r5_fused_mid_probe.txt).
  const bool fuse_mid_env = getenv("GM_MSM_FUSE_MID") && !strcmp(getenv("GM_MSM_FUSE_MID"), "1");
  std::vector<char> group_block;
  group_block.assign(fused_groups.size(), 0);
  if (fuse_env && fuse_mid_env && C->small_stream[0] != nullptr && C->msm_affine_levels <= 0 && !C->msm_c_override && bases->n <= ((size_t)1 << 30)) {
    std::vector<size_t> mid;
    for (size_t j = 0; j < k; j++)
      if (ns[j] > MSM_MULTI_MAX_N && ns[j] <= MSM_MULTI_BLOCK_MAX_N) mid.push_back(j);
    size_t at = 0;
    while (mid.size() - at >= 2) {
      std::vector<size_t> grp;
      size_t pairs = 0;
      while (at < mid.size() && grp.size() < (size_t)MULTI_MAX_LEVELS && pairs + ns[mid[at]] <= ((size_t)1 << 22)) {
        pairs += ns[mid[at]];
        grp.push_back(mid[at++]);
      }
      if (grp.size() < 2) break;
      for (size_t j : grp) group_of[j] = (int)fused_groups.size();
      fused_groups.push_back(std::move(grp));
      group_block.push_back(1);
    }
  }
  std::vector<char> group_done(fused_groups.size(), 0);
  size_t last_big_pos = (size_t)-1;  // position in `order` of the last call that takes a big lane
  for (size_t jo = 0; jo < k; jo++) {
    const size_t j = order[jo];
    const bool blk = group_of[j] >= 0 && group_block[(size_t)group_of[j]];
    if (blk || (group_of[j] < 0 && !is_small(j))) last_big_pos = jo;
  }
  bool gated = false;
  std::vector<hipEvent_t> gate_evs;
  int gate_left = 0;
  for (size_t jo = 0; jo < k; jo++) {
    const size_t j = order[jo];
    const bool fused_here = group_of[j] >= 0;
    if (fused_here && group_done[(size_t)group_of[j]]) continue;
    const bool block_here = fused_here && group_block[(size_t)group_of[j]];
    const bool small = block_here ? false : (fused_here || is_small(j));
    // big calls alternate between the two full-size lanes (0 and -1), each with two result buffers
    static const bool two_big = !(getenv("GM_MSM_BIG_LANES") && !strcmp(getenv("GM_MSM_BIG_LANES"), "1"));
    int lane, hslot;
    if (small) {
      // several result buffers per small lane as well: the next call of a lane is enqueued behind the first (same stream, same
      // workspace: stream order keeps them apart) without a host round trip in between
      lane = 1 + (small_rr % small_lanes);
      hslot = (small_rr / small_lanes) % small_slots;
      small_rr++;
    } else {
      lane = (two_big && C->stream_b && (big_rr & 1)) ? -1 : 0;
      hslot = (big_rr >> 1) & 1;
      big_rr++;
      if (smalls_first && !smalls_nogate && !gated) {  // (GM_MSM_BATCH_ORDER=smalls) the big lanes start behind the small calls that are still in flight
        gated = true;
        for (size_t t = 0; t < q.size(); t++)
          if (!finished[t] && q[t].lane > 0 && !q[t].P.empty) gate_evs.push_back(q[t].P.ws->done_ev[q[t].P.slot]);
        gate_left = C->stream_b ? 2 : 1;  // the first call of each big lane waits (its stream order holds the later ones back)
      }
    }
    // a (workspace, result buffer) pair is free again once its previous call has been finished
    while (busy(lane, hslot)) {
      int rc = drain_pair(lane, hslot);
      if (rc) return fail(rc);
    }
    Inflight e;
    e.j = j;
    e.lane = lane;
    MsmWorkspace& ws = lane > 0 ? C->msm_small[lane - 1] : (lane < 0 ? C->msm_b : C->msm);
    hipStream_t st = lane > 0 ? C->small_stream[lane - 1] : (lane < 0 ? C->stream_b : C->stream);
    MsmStreams sts{st, st, st};
    if (C->cu_split) {  // the CU partition of a batch (ctx.hpp): sorts where GM_CU_SPLIT_SORT says, accumulation and tail on their own CUs
      static const bool sort_on_tail = getenv("GM_CU_SPLIT_SORT") && !strcmp(getenv("GM_CU_SPLIT_SORT"), "tail");
      static const bool small_whole = getenv("GM_CU_SPLIT_SMALL") && !strcmp(getenv("GM_CU_SPLIT_SMALL"), "tail");
      const int li = lane > 0 ? 1 + lane : (lane < 0 ? 1 : 0);
      if (lane > 0 && small_whole) sts = MsmStreams{C->part_tail[li], C->part_tail[li], C->part_tail[li]};  // a small call entirely on the tail CUs
      else sts = MsmStreams{sort_on_tail ? C->part_tail[li] : C->part_acc[li], C->part_acc[li], C->part_tail[li]};
      // the LAST big call has no accumulation after it to hide its tail under: its merge and reduction take the seven XCDs of the
      // accumulations instead of the one of the tails (GM_CU_SPLIT_LAST=0: the tail XCD like every other call)
      static const bool last_wide = !(getenv("GM_CU_SPLIT_LAST") && atoi(getenv("GM_CU_SPLIT_LAST")) == 0);
      if (last_wide && !small && jo == last_big_pos) sts.tail = sts.acc;
      st = sts.sort;
      // the calls of a lane share its workspace: on ONE stream the order kept them apart, over three streams the next call's sort
      // must wait for the previous call's tail (it clears the counters the bucket reduction reads)
      if (ws.have_done_ev)
        for (int e2 = 0; e2 < MSM_SLOTS; e2++) GM_HIP(hipStreamWaitEvent(sts.sort, ws.done_ev[e2], 0));
      GM_HIP(hipStreamWaitEvent(sts.sort, C->start_ev, 0));
      if (sts.tail != sts.sort) GM_HIP(hipStreamWaitEvent(sts.tail, C->start_ev, 0));
      if (sts.acc != sts.sort) GM_HIP(hipStreamWaitEvent(sts.acc, C->start_ev, 0));
    } else
    if (st != C->stream) GM_HIP(hipStreamWaitEvent(st, C->start_ev, 0));  // scalars produced on the main stream
    if (!small && gate_left > 0) {
      // on the stream this call's SORT is enqueued on: with the CU partition that is a partition stream, not C->stream / stream_b
      // (the gate used to wait on those two and was a silent no-op under cu_split)
      gate_left--;
      for (hipEvent_t ev : gate_evs) GM_HIP(hipStreamWaitEvent(st, ev, 0));
    }
    const auto tq0 = std::chrono::steady_clock::now();
    auto start_of = [&](size_t jj) { return pair_offsets ? (int64_t)pair_offsets[jj] : (firsts ? firsts[jj] : first); };
    int rc;
    if (fused_here) {
      const std::vector<size_t>& grp = fused_groups[(size_t)group_of[j]];
      MsmMulti M;
      M.block = block_here;
      M.levels = (int)grp.size();
      for (size_t l = 0; l < grp.size(); l++) {
        M.scalars[l] = d_scalars[grp[l]];
        M.n[l] = ns[grp[l]];
        M.start[l] = start_of(grp[l]);
      }
      e.fused = grp;
      group_done[(size_t)group_of[j]] = 1;
      rc = msm_enqueue(C, ws, sts, bases, 0, step, nullptr, mont, 0, hslot, &e.P, 0, 1, &M);
    } else
      rc = msm_enqueue(C, ws, sts, bases, start_of(j), step, d_scalars[j], mont,
                         ns[j], hslot, &e.P);
    enqueue_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count();
    if (batch_trace)
      fprintf(stderr, "[gm msm batch]   enqueue call %zu (n = %zu, lane %d): host %.3f .. %.3f ms\n", j, fused_here ? (size_t)0 : ns[j], lane,
              std::chrono::duration<double, std::milli>(tq0 - t_batch0).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_batch0).count());
    if (rc) return fail(rc);
    q.push_back(e);
    finished.push_back(0);
  }
  const auto t_enqueued = std::chrono::steady_clock::now();
  for (;;) {
    bool left = false;
    for (char f : finished) left = left || !f;
    if (!left) break;
    const auto td0 = std::chrono::steady_clock::now();
    size_t before = 0;
    for (char f : finished) before += f ? 1 : 0;
    int rc = drain_oldest();
    if (rc) return fail(rc);
    if (batch_trace) {
      size_t after = 0;
      for (char f : finished) after += f ? 1 : 0;
      fprintf(stderr, "[gm msm batch]   drain: %zu call(s) finished in %.3f ms\n", after - before,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count());
    }
  }
  if (batch_trace)
    fprintf(stderr, "[gm msm batch] %zu calls: %.3f ms, of which enqueueing %.3f ms, final drains %.3f ms\n", k,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_batch0).count(), enqueue_s * 1e3,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enqueued).count());
  return GM_OK;
}

static int msm_enqueue(Context* C, MsmWorkspace& ws, MsmStreams sts, const Bases* bases, int64_t first, int64_t step, const void* d_scalars,
                       int mont, size_t n, int slot, MsmPending* P, int part, int nparts, const MsmMulti* multi) {
  hipStream_t st = sts.sort;  // memsets + sort; the accumulation runs on sts.acc, everything after it on sts.tail
  const size_t nbases = bases->n;
  P->ws = &ws;
  P->slot = slot;
  P->multi_levels = 0;
  if (multi) {
    GM_CHECK(multi->levels >= 1 && multi->levels <= MULTI_MAX_LEVELS && nparts == 1, GM_EINVAL, "msm: %d fused calls (1 .. %d)", multi->levels, MULTI_MAX_LEVELS);
    // the entries of a fused pass carry ABSOLUTE base indices in 30 bits (bit 30: the GLV half, bit 31: the sign)
    GM_CHECK(nbases <= ((size_t)1 << 30), GM_EINVAL, "msm: a key of %zu points is too long for a fused pass (2^30)", nbases);
    n = 0;
    for (int l = 0; l < multi->levels; l++) {
      GM_CHECK(multi->n[l] >= 1 && multi->n[l] <= (multi->block ? MSM_MULTI_BLOCK_MAX_N : MSM_MULTI_MAX_N), GM_EINVAL, "msm: a fused call of %zu pairs", multi->n[l]);
      const int64_t lo = multi->start[l], hi = multi->start[l] + step * (int64_t)(multi->n[l] - 1);
      GM_CHECK(lo >= 0 && hi >= 0 && (size_t)lo < nbases && (size_t)hi < nbases, GM_EINVAL, "msm: base range [%lld .. %lld] outside registered bases (len %zu)",
               (long long)lo, (long long)hi, nbases);
      n += multi->n[l];
    }
    first = 0;  // the entries carry absolute base indices
  }
  P->empty = n == 0;
  if (n == 0) return GM_OK;
  GM_CHECK(n < (1ull << 31), GM_EINVAL, "msm: n = %zu exceeds 2^31 - 1 pairs per call; chunk the stream", n);
  if (!multi) {
    int64_t last = first + step * (int64_t)(n - 1);
    GM_CHECK(first >= 0 && last >= 0 && (size_t)first < nbases && (size_t)last < nbases, GM_EINVAL,
             "msm: base range [%lld .. %lld] outside registered bases (len %zu)", (long long)first, (long long)last,
             nbases);
  }
  // fixed-base tables (gm_g1_bases_precompute) serve large MSMs; small ones are latency-bound and
  // cheaper with few buckets
  const size_t tab_min = std::max(C->msm_table_min, bases->tab_min);
  const uint8_t* tab_ptr = nullptr;  // the table set this call takes: the key's main one, or a prefix set that covers its range
  int tab_c_sel = 0;
  size_t tab_n = 0;
  static const bool fuse_tables_env = !(getenv("GM_MSM_FUSE_TABLES") && atoi(getenv("GM_MSM_FUSE_TABLES")) == 0);  // A/B knob
  if (multi && multi->block && !C->msm_c_override && fuse_tables_env) {
    // a levelled pass takes a table set that covers EVERY level's range and whose window leaves the bucket count sane (one set of
    // 2^(c-1) buckets per level): the key's own c = 20 tables, or a c <= 20 prefix set
    auto covers = [&](size_t tn) {
      for (int l = 0; l < multi->levels; l++) {
        const int64_t lo = multi->start[l], hi = multi->start[l] + step * (int64_t)(multi->n[l] - 1);
        if (lo < 0 || hi < 0 || (size_t)lo >= tn || (size_t)hi >= tn) return false;
      }
      return tn <= ((size_t)1 << ENTRY_W_SHIFT);
    };
    if (bases->table != nullptr && bases->tab_c <= 20 && covers(bases->n)) {
      tab_ptr = bases->table;
      tab_c_sel = bases->tab_c;
      tab_n = bases->n;
    } else {
      for (const Bases::TableSet& ts : bases->extra)
        if (ts.c <= 20 && ts.c >= 18 && covers(ts.n)) {
          tab_ptr = ts.t;
          tab_c_sel = ts.c;
          tab_n = ts.n;
          break;
        }
    }
  }
  if (!multi && !C->msm_c_override && n < ((size_t)1 << ENTRY_W_SHIFT)) {
    if (bases->table != nullptr && n >= tab_min) {
      tab_ptr = bases->table;
      tab_c_sel = bases->tab_c;
      tab_n = bases->n;
    } else if (C->msm_table_min <= ((size_t)1 << 17) || n >= C->msm_table_min) {
      const int64_t last = first + step * (int64_t)(n - 1);
      for (const Bases::TableSet& ts : bases->extra)
        if (n >= ts.min_n && n < ts.max_n && first >= 0 && last >= 0 && (size_t)first < ts.n && (size_t)last < ts.n) {
          tab_ptr = ts.t;
          tab_c_sel = ts.c;
          tab_n = ts.n;
          break;
        }
    }
  }
  const bool use_table = tab_ptr != nullptr;
  const bool multi_block = multi && multi->block;
  static const int fuse_c_env = getenv("GM_MSM_FUSE_C") ? atoi(getenv("GM_MSM_FUSE_C")) : 16;  // A/B knob: window of a levelled pass without tables
  const int c = multi ? (multi_block ? (use_table ? tab_c_sel : fuse_c_env) : 8) : (use_table ? tab_c_sel : (C->msm_c_override ? C->msm_c_override : choose_window(n)));
  GM_CHECK(c >= 2 && c <= 22, GM_EINVAL, "msm: window width %d out of range [2, 22]", c);
  // GLV (glv_split): two 128-bit digit strings per scalar over HALF the windows, the second one on phi(P)
  static const bool sort_atomic_env0 = getenv("GM_MSM_SORT") && !strcmp(getenv("GM_MSM_SORT"), "atomic");
  // waves per SIMD of k_acc0: 2 (198 VGPRs) and 3 (168) measure the same (2.13 / 2.15 ms at 2^20: the kernel is issue-bound,
  // not latency-bound); 2 leaves 112 registers per lane of a SIMD for other kernels
  static const int acc0_waves = getenv("GM_ACC0_WAVES") ? atoi(getenv("GM_ACC0_WAVES")) : 2;
  const size_t bucket_bytes = XYZZ30_BYTES;  // buckets, partials, row / column sums: 208-byte loose records (g1.cuh: Acc30)
  const bool use_glv = GM_GLV(1) && !multi && bases->phi != nullptr && !use_table && C->msm_affine_levels == 0 && !sort_atomic_env0 && n <= ((size_t)1 << 26);
  const int W = ((use_glv ? 128 : 256) + c - 1) / c;
  GM_CHECK(nparts == 1 || !use_table, GM_EINVAL, "msm: the fixed-base table path is not split into window groups");
  const int w_lo = part * W / nparts, Wg = (part + 1) * W / nparts - w_lo;  // this call's window group
  const uint32_t B = 1u << (c - 1);
  // bucket sets (fused calls: one set per (call, window); with tables one per call)
  const int Wb = multi ? multi->levels * (use_table ? 1 : W) : (use_table ? 1 : Wg);
  const size_t nbuckets = (size_t)Wb * B;
  const uint8_t* d_bases = use_table ? tab_ptr : bases->d;
  const long long tab_stride = use_table ? (long long)tab_n : 0;
  const uint64_t Nall = (uint64_t)n * (uint64_t)W * (use_glv ? 2u : 1u);
  const uint64_t N = (uint64_t)n * (uint64_t)(use_table ? W : Wg) * (use_glv ? 2u : 1u);  // entries of this window group
  GM_CHECK(Nall < (1ull << 32), GM_EINVAL, "msm: n*W = %llu entries exceed 2^32; chunk the stream", (unsigned long long)Nall);

  // affine tree levels in front of the XYZZ accumulation (see k_lvl_*): automatic = as many as leave ~4
  // entries per bucket, none for small calls where the per-level round trip to the host costs more
  int levels = C->msm_affine_levels;
  if (levels < 0) {
    levels = 0;
    if (N >= ((uint64_t)1 << 22))
      for (uint64_t avg = N / nbuckets; avg >= 8 && levels < 6; avg >>= 1) levels++;
  }
  GM_CHECK(levels <= 8, GM_EINVAL, "msm: %d affine levels (max 8)", levels);
  uint64_t Nacc = N;  // upper bound on the entries k_acc0 sees: every level maps m -> ceil(m / 2) per bucket
  for (int l = 0; l < levels; l++) Nacc = Nacc / 2 + 2 * nbuckets + 2;

  // level-0 chunk length: keep >= 2 waves per SIMD when the problem is large enough
  static const int L_env = getenv("GM_MSM_L") ? atoi(getenv("GM_MSM_L")) : 0;  // tuning override
  // (longer chunks for big calls: fewer keyed partials for k_merge -- 4.1 -> 2.4 ms at 2^24 pairs)
  // (from 2^24 entries on, two rounds of blocks overlap gather and arithmetic better than one: 4.18 -> 4.01 ms at 2^20 pairs)
  // (a window group of a split call uses the chunk length of the whole call: its two accumulations run back to back)
  const uint64_t Nacc_all = nparts > 1 ? Nall : Nacc;
  const uint64_t lanes0 = Nacc_all >= ((uint64_t)1 << 24) ? 262144 : 131072;
  static const uint64_t L_cap = getenv("GM_MSM_LCAP") ? (uint64_t)atoi(getenv("GM_MSM_LCAP")) : 256;  // tuning override
  uint32_t L = (uint32_t)std::min<uint64_t>(L_cap, std::max<uint64_t>(4, (Nacc_all + lanes0 - 1) / lanes0));
  if (L_env > 0) L = (uint32_t)L_env;
  L = (L + 1u) & ~1u;  // even: every lane's chunk starts 16-byte aligned (k_acc0 reads its entries in 16-byte words)
  const uint64_t T0 = (Nacc + L - 1) / L;
  const uint64_t T0pad = (T0 + 255) / 256 * 256;
  const uint64_t E1 = 2 * T0pad;

  int rc;
  // [nbuckets + 1] = scalar-range flag; behind it the coarse-bin counters of the block sort, so that ONE memset clears both
  const size_t counts_words = (nbuckets + 2 + SORT_GMAX + 1 + 63) / 64 * 64;
  if ((rc = ws.counts.ensure(counts_words * 4))) return rc;
  if ((rc = ws.offsets.ensure((nbuckets + 1) * 4))) return rc;
  if ((rc = ws.cursor.ensure((nbuckets + 1) * 4))) return rc;
  if ((rc = ws.misc.ensure((nbuckets / SCAN_PER_BLOCK + 2) * 4))) return rc;
  if ((rc = ws.entries.ensure(N * 8))) return rc;
  if ((rc = ws.buckets.ensure(nbuckets * bucket_bytes))) return rc;
  if ((rc = ws.pk[0].ensure(E1 * 4))) return rc;
  if ((rc = ws.pp[0].ensure(E1 * bucket_bytes))) return rc;
  const uint64_t E2 = 2 * ((E1 + 127) / 128);
  if ((rc = ws.pk[1].ensure(E2 * 4))) return rc;
  if ((rc = ws.pp[1].ensure(E2 * bucket_bytes))) return rc;

  const uint32_t* sc = reinterpret_cast<const uint32_t*>(d_scalars);
  GM_HIP(hipMemsetAsync(ws.counts.p, 0, counts_words * 4, st));
  uint32_t* d_err = ws.counts.as<uint32_t>() + nbuckets + 1;
  // levels (an experiment build) rewrite the entry lists, so the counts no longer say which buckets get written: clear there
  const bool bucket_counts_valid = levels == 0;
  if (!bucket_counts_valid) GM_HIP(hipMemsetAsync(ws.buckets.p, 0, nbuckets * bucket_bytes, st));
  const uint32_t dblocks = (uint32_t)((n + 255) / 256);
  Profiler& pf = C->prof;
  auto run_scan = [&]() {
    const uint32_t nb = (uint32_t)((nbuckets + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, st, ws.counts.as<uint32_t>(), (uint32_t)nbuckets,
                       ws.misc.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, ws.misc.as<uint32_t>(), nb,
                       ws.offsets.as<uint32_t>() + nbuckets);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, ws.counts.as<uint32_t>(), (uint32_t)nbuckets,
                       ws.misc.as<uint32_t>(), ws.offsets.as<uint32_t>(), ws.cursor.as<uint32_t>());
  };
  static const bool sort_atomic_env = getenv("GM_MSM_SORT") && !strcmp(getenv("GM_MSM_SORT"), "atomic");
  static const bool sort_flat_env = !(getenv("GM_MSM_SORT") && !strcmp(getenv("GM_MSM_SORT"), "blocks"));
  const bool sort_atomic = sort_atomic_env && !use_table && nparts == 1 && !multi_block;
  // (a table call keeps the block sort at every size: one shared bucket set means 16 x the contention on the flat path's global
  // counters -- measured with a flat variant for tables: 0.63 against 0.61 ms at 2^14 pairs, 0.90 against 0.81 at 2^16)
  const bool sort_flat = sort_flat_env && !sort_atomic && !use_table && nparts == 1 && !multi_block && N <= ((uint64_t)1 << 21) && nbuckets <= ((size_t)1 << 18);
  LevelGeom lg{};
  if (multi_block) {
    lg.levels = multi->levels;
    lg.step = (int)step;
    uint32_t at = 0;
    for (int l = 0; l < multi->levels; l++) {
      lg.start[l] = at;
      lg.scal[l] = reinterpret_cast<const uint32_t*>(multi->scalars[l]);
      lg.base0[l] = (long long)multi->start[l];
      at += (uint32_t)multi->n[l];
    }
    lg.start[multi->levels] = at;
  }
  if (multi && !multi_block) {
    MultiGeom mg{};
    mg.c = c;
    mg.W = W;
    mg.levels = multi->levels;
    mg.step = (int)step;
    mg.B = B;
    for (int j = 0; j + 1 < W; j++) {
      const int b = c * j + c - 1;
      if (b < 256) mg.K[b >> 5] |= 1u << (b & 31);
    }
    uint32_t at = 0;
    for (int l = 0; l < multi->levels; l++) {
      mg.start[l] = at;
      mg.scal[l] = reinterpret_cast<const uint32_t*>(multi->scalars[l]);
      mg.base0[l] = (long long)multi->start[l];
      at += (uint32_t)multi->n[l];
    }
    mg.start[multi->levels] = at;
    GM_CHECK(N <= ((uint64_t)1 << 21) && nbuckets <= ((size_t)1 << 18), GM_EINVAL, "msm: fused calls too large for the flat sort (%llu entries, %zu buckets)",
             (unsigned long long)N, nbuckets);
    const dim3 grid((uint32_t)((n + 255) / 256), (uint32_t)W);
    pf.begin(part, PROF_DIGITS, st);
    hipLaunchKernelGGL(k_digits_multi<false>, grid, dim3(256), 0, st, mg, mont, ws.counts.as<uint32_t>(), (uint64_t*)nullptr, d_err);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, ws.counts.as<uint32_t>(), (uint32_t)nbuckets, ws.offsets.as<uint32_t>(),
                       ws.cursor.as<uint32_t>());
    pf.end(part, PROF_DIGITS, st);
    pf.begin(part, PROF_SCATTER, st);
    hipLaunchKernelGGL(k_digits_multi<true>, grid, dim3(256), 0, st, mg, mont, ws.cursor.as<uint32_t>(), ws.entries.as<uint64_t>(), d_err);
    pf.end(part, PROF_SCATTER, st);
  } else if (sort_flat) {
    FlatGeom fg{};
    fg.c = c;
    fg.W = W;
    fg.glv = use_glv ? 1 : 0;
    fg.B = B;
    for (int j = 0; j + 1 < W; j++) {
      const int b = c * j + c - 1;
      if (b < 256) fg.K[b >> 5] |= 1u << (b & 31);
    }
    const dim3 grid((uint32_t)((n + 255) / 256), (uint32_t)(W * (use_glv ? 2 : 1)));
    pf.begin(part, PROF_DIGITS, st);
    hipLaunchKernelGGL(k_digits_flat<false>, grid, dim3(256), 0, st, sc, (uint32_t)n, mont, fg, ws.counts.as<uint32_t>(), (uint64_t*)nullptr, d_err);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, ws.counts.as<uint32_t>(), (uint32_t)nbuckets, ws.offsets.as<uint32_t>(),
                       ws.cursor.as<uint32_t>());
    pf.end(part, PROF_DIGITS, st);
    pf.begin(part, PROF_SCATTER, st);
    hipLaunchKernelGGL(k_digits_flat<true>, grid, dim3(256), 0, st, sc, (uint32_t)n, mont, fg, ws.cursor.as<uint32_t>(), ws.entries.as<uint64_t>(), d_err);
    pf.end(part, PROF_SCATTER, st);
  } else if (sort_atomic) {
    pf.begin(part, PROF_DIGITS, st);
    hipLaunchKernelGGL(k_msm_digits<false>, dim3(dblocks), dim3(256), 0, st, sc, (uint32_t)n, mont, c, W, B,
                       ws.counts.as<uint32_t>(), (uint64_t*)nullptr, d_err);
    pf.end(part, PROF_DIGITS, st);
    pf.begin(part, PROF_SCAN, st);
    run_scan();
    pf.end(part, PROF_SCAN, st);
    pf.begin(part, PROF_SCATTER, st);
    hipLaunchKernelGGL(k_msm_digits<true>, dim3(dblocks), dim3(256), 0, st, sc, (uint32_t)n, mont, c, W, B,
                       ws.cursor.as<uint32_t>(), ws.entries.as<uint64_t>(), d_err);
    pf.end(part, PROF_SCATTER, st);
  } else {
    SortGeom sg;
    sg.c = c;
    sg.W = W;
    sg.w_lo = use_table ? 0 : w_lo;
    sg.Wg = use_table ? W : Wg;
    sg.glv = use_glv ? 1 : 0;
    sg.B = B;
    sg.shared = use_table ? 1 : 0;
    sg.FB = std::min<uint32_t>((uint32_t)(c - 1), 10u);
    // (a levelled pass has levels x as many buckets: 2048 fine bins keep the coarse counters of the LDS-staged scatter inside 160 KB)
    if (multi_block) sg.FB = std::min<uint32_t>((uint32_t)(c - 1), std::max<uint32_t>(sg.FB, 11u));
    while ((nbuckets >> sg.FB) > SORT_GMAX && (1u << sg.FB) < SORT_FMAX) sg.FB++;
    sg.G = (uint32_t)(nbuckets >> sg.FB);
    GM_CHECK(sg.G <= SORT_GMAX, GM_EINVAL, "msm: %u coarse sort bins exceed %u (window %d too wide for this sort)", sg.G, SORT_GMAX, c);
    if ((rc = ws.tmp_entries.ensure(N * 8))) return rc;
    if ((rc = ws.sortmeta.ensure((size_t)(3 * (sg.G + 1)) * 4))) return rc;
    uint32_t* gcount = ws.counts.as<uint32_t>() + nbuckets + 2;  // cleared with the bucket counters above
    uint32_t* goff = ws.sortmeta.as<uint32_t>();
    uint32_t* gcursor = goff + (sg.G + 1);
    uint32_t* blkoff = gcursor + (sg.G + 1);
    const uint32_t b1 = (uint32_t)((n + SORT_TS - 1) / SORT_TS);
    const uint32_t b2 = (uint32_t)(N / SORT_CH + sg.G + 1);
    pf.begin(part, PROF_DIGITS, st);
    hipLaunchKernelGGL(k_sort1<false>, dim3(b1), dim3(256), 0, st, sc, (uint32_t)n, mont, sg, lg, gcount, (uint64_t*)nullptr, d_err);
    hipLaunchKernelGGL(k_sort1_scan, dim3(1), dim3(1024), 0, st, gcount, sg.G, goff, gcursor, blkoff);
    const size_t stage1_lds = (size_t)SORT_TS * sg.Wg * (use_glv ? 2 : 1) * 8 + (size_t)3 * sg.G * 4 + 1024 * 4;
    static const bool sort1_staged_env = !(getenv("GM_MSM_SORT1") && !strcmp(getenv("GM_MSM_SORT1"), "direct"));
    if (sort1_staged_env && sg.Wg * (use_glv ? 2 : 1) <= SORT1_STAGE_WMAX && stage1_lds <= 160 * 1024) {
      static bool attr_set = false;
      if (!attr_set) {
        GM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort1_staged), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
      }
      hipLaunchKernelGGL(k_sort1_staged, dim3(b1), dim3(1024), stage1_lds, st, sc, (uint32_t)n, mont, sg, lg, gcursor, ws.tmp_entries.as<uint64_t>());
    } else {
      hipLaunchKernelGGL(k_sort1<true>, dim3(b1), dim3(256), 0, st, sc, (uint32_t)n, mont, sg, lg, gcursor, ws.tmp_entries.as<uint64_t>(), d_err);
    }
    pf.end(part, PROF_DIGITS, st);
    pf.begin(part, PROF_SCATTER, st);
    hipLaunchKernelGGL(k_sort2<false>, dim3(b2), dim3(256), 0, st, ws.tmp_entries.as<uint64_t>(), goff, blkoff, sg,
                       ws.counts.as<uint32_t>(), (uint64_t*)nullptr);
    run_scan();
    static const bool sort_staged = !(getenv("GM_MSM_SORT2") && !strcmp(getenv("GM_MSM_SORT2"), "direct"));
    if (sort_staged && (1u << sg.FB) <= SORT_STAGE_FMAX)
      hipLaunchKernelGGL(k_sort2_staged, dim3(b2), dim3(1024), 0, st, ws.tmp_entries.as<uint64_t>(), goff, blkoff, sg,
                         ws.cursor.as<uint32_t>(), ws.entries.as<uint64_t>());
    else
      hipLaunchKernelGGL(k_sort2<true>, dim3(b2), dim3(256), 0, st, ws.tmp_entries.as<uint64_t>(), goff, blkoff, sg,
                         ws.cursor.as<uint32_t>(), ws.entries.as<uint64_t>());
    pf.end(part, PROF_SCATTER, st);
  }
  const uint64_t* acc_entries = ws.entries.as<uint64_t>();
  const uint32_t* acc_total = ws.offsets.as<uint32_t>() + nbuckets;
  const uint8_t* acc_bases = d_bases;
  long long acc_first = (long long)first, acc_step = multi ? 1ll : (long long)step, acc_tab = tab_stride;
#ifdef GM_EXPERIMENTS
#define GM_MSM_LEVELS_HOST 1
#include "msm_levels.inc"
#undef GM_MSM_LEVELS_HOST
#else
  GM_CHECK(levels == 0, GM_EINVAL, "msm: affine tree levels are an experiment of round 2; rebuild with -DGM_EXPERIMENTS (DESIGN.md section 8)");
#endif
  if (!ws.have_done_ev) {
    for (int e = 0; e < MSM_SLOTS; e++) GM_HIP(hipEventCreateWithFlags(&ws.done_ev[e], hipEventDisableTiming));
    GM_HIP(hipEventCreateWithFlags(&ws.sort_ev, hipEventDisableTiming));
    GM_HIP(hipEventCreateWithFlags(&ws.acc_ev, hipEventDisableTiming));
    ws.have_done_ev = true;
  }
  if (sts.acc != st) {
    GM_HIP(hipEventRecord(ws.sort_ev, st));
    GM_HIP(hipStreamWaitEvent(sts.acc, ws.sort_ev, 0));
  }
  st = sts.acc;
  pf.begin(part, PROF_ACC0, st);
  // dynamic LDS padding caps the blocks per CU: the kernel needs 162 VGPRs, so three waves per SIMD WOULD fit
  static const size_t acc0_lds_pad = getenv("GM_ACC0_LDS_PAD") ? (size_t)strtoull(getenv("GM_ACC0_LDS_PAD"), nullptr, 10) : 0;
  if ((rc = ws.clk.ensure(16))) return rc;
  unsigned long long* acc_clk = ws.clk.as<unsigned long long>();
  // gather of the next entry under the addition of this one: GM_ACC0_PREFETCH = 0 / 1 forces, default = on the table path
  static const int acc0_pf_env = getenv("GM_ACC0_PREFETCH") ? atoi(getenv("GM_ACC0_PREFETCH")) : -1;
  const bool acc0_pf = acc0_pf_env >= 0 ? acc0_pf_env != 0 : use_table;
  if (acc0_pf)
    hipLaunchKernelGGL(k_acc0_pf<2>, dim3((uint32_t)(T0pad / 256)), dim3(256), 0, st, acc_entries, acc_total, acc_bases, acc_first, acc_step,
                       acc_tab, L, ws.pk[0].as<uint32_t>(), ws.pp[0].as<uint8_t>(), ws.buckets.as<uint8_t>(), use_glv ? bases->phi : (const uint8_t*)nullptr, acc_clk);
  else if (acc0_waves == 2)
    hipLaunchKernelGGL(k_acc0<2>, dim3((uint32_t)(T0pad / 256)), dim3(256), acc0_lds_pad, st, acc_entries, acc_total, acc_bases, acc_first, acc_step,
                       acc_tab, L, ws.pk[0].as<uint32_t>(), ws.pp[0].as<uint8_t>(), ws.buckets.as<uint8_t>(), use_glv ? bases->phi : (const uint8_t*)nullptr, acc_clk);
  else
    hipLaunchKernelGGL(k_acc0<3>, dim3((uint32_t)(T0pad / 256)), dim3(256), 0, st, acc_entries, acc_total, acc_bases, acc_first, acc_step,
                       acc_tab, L, ws.pk[0].as<uint32_t>(), ws.pp[0].as<uint8_t>(), ws.buckets.as<uint8_t>(), use_glv ? bases->phi : (const uint8_t*)nullptr, acc_clk);
  pf.end(part, PROF_ACC0, st);
  if (sts.tail != st) {
    GM_HIP(hipEventRecord(ws.acc_ev, st));
    GM_HIP(hipStreamWaitEvent(sts.tail, ws.acc_ev, 0));
  }
  st = sts.tail;
  pf.begin(part, PROF_MERGE, st);
  {
    uint64_t E = E1;
    int src = 0;
    for (;;) {
      uint32_t waves = (uint32_t)((E + 127) / 128);
      int final_level = waves == 1;
      hipLaunchKernelGGL(k_merge, dim3(waves), dim3(64), 0, st, ws.pk[src].as<uint32_t>(), ws.pp[src].as<uint8_t>(),
                         (uint32_t)E, ws.pk[src ^ 1].as<uint32_t>(), ws.pp[src ^ 1].as<uint8_t>(),
                         ws.buckets.as<uint8_t>(), final_level);
      if (final_level) break;
      E = 2ull * waves;
      src ^= 1;
    }
  }

  pf.end(part, PROF_MERGE, st);
  pf.begin(part, PROF_REDUCE, st);
  // bucket reduction (see k_group_sum): index bits split into m <= 3 fields w0 (low), w1, w2
  const uint32_t nbits = (uint32_t)(c - 1);
  uint32_t wf[3] = {0, 0, 0};
  int m;
  static const int fields_env = getenv("GM_MSM_FIELDS") ? atoi(getenv("GM_MSM_FIELDS")) : 0;  // tuning override
  if (nbits <= 7) {
    m = 1;
    wf[0] = nbits;
  } else if (nbits <= 14 || (fields_env == 2 && nbits <= 16)) {
    m = 2;
    wf[0] = nbits / 2;
    wf[1] = nbits - wf[0];
  } else {
    m = 3;
    // an odd bit goes to the LOW field (19 bits of the c = 20 tables: 7 + 6 + 6): level 1 sums over it anyway (throughput-bound),
    // and the latency-bound level 2 then sums 64 instead of 128 elements per output -- two dependent additions less
    static const bool odd_low = !(getenv("GM_MSM_ODD_BIT") && !strcmp(getenv("GM_MSM_ODD_BIT"), "high"));  // A/B knob
    wf[0] = odd_low ? (nbits + 2) / 3 : nbits / 3;
    wf[1] = (nbits - wf[0]) / 2;
    wf[2] = nbits - wf[0] - wf[1];
  }
  // lanes per output: enough lanes to put ~2^17 threads in flight (2 waves per SIMD), never more than
  // the element count; few lanes keep the work near the minimum (the shuffle tree runs on every lane)
  static const int lpo_env = getenv("GM_MSM_LPO_LOG") ? atoi(getenv("GM_MSM_LPO_LOG")) : 0;  // tuning override
  // (measured: 2^17 threads per launch for the 0.5 M buckets of c = 16 -- 0.57 vs 0.67 ms --, 2^18 for the
  // millions of buckets of the wide windows)
  const int lpo_target_log = lpo_env ? lpo_env : (nbuckets >= ((size_t)1 << 21) ? 18 : 17);
  // the jobs of one launch share the GPU: aim the launch as a whole at ~2^17 threads
  uint32_t lpo_jobs = 1;
  auto lpo_for = [&](uint32_t len, uint32_t n_out) {
    uint32_t s = 0;
    while (s < 5 && (1u << s) < len && ((uint64_t)n_out << s) * lpo_jobs < (1ull << lpo_target_log)) s++;
    return s;
  };
  // the bucket array is NOT cleared per call (109 MB at 2^20 pairs): the first level skips buckets whose entry count is zero
  const uint8_t* bucket_in = ws.buckets.as<uint8_t>();
  const uint32_t* bucket_cnt = bucket_counts_valid ? ws.counts.as<uint32_t>() : nullptr;
  auto strided = [&](const uint8_t* in, uint8_t* out, uint32_t win_in, uint32_t n_hi, uint32_t n_lo, uint32_t s_hi, uint32_t s_lo,
                     uint32_t s_e, uint32_t len) {
    GroupSumArgs g{};
    g.in = in; g.out = out; g.mode = GS_STRIDED; g.in_cnt = in == bucket_in ? bucket_cnt : nullptr;
    g.per_win = n_hi * n_lo; g.n_out = (uint32_t)Wb * g.per_win; g.win_stride = win_in;
    g.n_lo = n_lo; g.s_hi = s_hi; g.s_lo = s_lo; g.s_e = s_e; g.len = len; g.lpo_shift = lpo_for(len, g.n_out);
    return g;
  };
  auto plane = [&](const uint8_t* in, uint8_t* out, uint32_t nb) {
    GroupSumArgs g{};
    g.in = in; g.out = out; g.mode = GS_PLANE; g.out_canonical = 1;  // what the host reads
    g.in_cnt = in == bucket_in ? bucket_cnt : nullptr;
    g.per_win = nb + 1; g.n_out = (uint32_t)Wb * g.per_win; g.win_stride = 1u << nb; g.nb = nb;
    g.lpo_shift = lpo_for(nb ? (1u << (nb - 1)) : 1u, g.n_out);
    return g;
  };
  uint8_t* planes = nullptr;
  size_t plane_count = 0;
  auto launch = [&](std::initializer_list<GroupSumArgs> jobs, bool last = false) {
    GroupSumJobs J{};
    if (last) {
      J.err_src = d_err;
      J.err_dst = reinterpret_cast<uint32_t*>(planes + plane_count * XYZZ_BYTES);
      J.clk_src = ws.clk.as<unsigned long long>();
    }
    uint32_t tot = 0;
    int k = 0;
    for (const GroupSumArgs& g : jobs) {
      J.j[k] = g;
      tot += (uint32_t)((((uint64_t)g.n_out << g.lpo_shift) + 255) / 256);
      J.blk_end[k] = tot;
      k++;
    }
    for (; k < 3; k++) J.blk_end[k] = tot;
    hipLaunchKernelGGL(k_group_sum, dim3(tot), dim3(256), 0, st, J);
  };
  // plane output layout: field k occupies Wb * (wf[k] + 1) records starting at plane_off[k]
  size_t plane_off[3];
  for (int k = 0; k < m; k++) {
    plane_off[k] = plane_count;
    plane_count += (size_t)Wb * (wf[k] + 1);
  }
  const size_t plane_bytes = plane_count * XYZZ_BYTES + 24;  // + the scalar-range flag + k_acc0's clock readings
  if (ws.host_planes_cap[slot] < plane_bytes) {
    if (ws.host_planes[slot]) (void)hipHostFree(ws.host_planes[slot]);
    ws.host_planes[slot] = nullptr;
    ws.host_planes_cap[slot] = 0;
    GM_HIP(hipHostMalloc((void**)&ws.host_planes[slot], plane_bytes, hipHostMallocDefault));
    ws.host_planes_cap[slot] = plane_bytes;
  }
  const bool zc = (C->zero_copy & 2) != 0;  // the last reduction launch writes what the host reads into the pinned buffer itself
  if (zc) {
    planes = reinterpret_cast<uint8_t*>(ws.host_planes[slot]);
  } else {
    if ((rc = ws.planes.ensure(plane_count * XYZZ_BYTES + 64))) return rc;
    planes = ws.planes.as<uint8_t>();
  }
  const uint8_t* X = ws.buckets.as<uint8_t>();
  const uint32_t n0 = 1u << wf[0], n1 = 1u << wf[1], n2 = 1u << wf[2];
  if (m == 1) {
    lpo_jobs = 1;
    launch({plane(X, planes, wf[0])}, true);
  } else if (m == 2) {
    // Y1[d1] = sum_{d0} X (rows), Y0[d0] = sum_{d1} X (columns)
    if ((rc = ws.rows.ensure((size_t)Wb * (n0 + n1) * XYZZ30_BYTES))) return rc;
    uint8_t* Y0 = ws.rows.as<uint8_t>();
    uint8_t* Y1 = Y0 + (size_t)Wb * n0 * XYZZ30_BYTES;
    lpo_jobs = 2;
    launch({strided(X, Y1, B, 1, n1, 0, n0, 1, n0), strided(X, Y0, B, 1, n0, 0, 1, n0, n1)});
    launch({plane(Y0, planes + plane_off[0] * XYZZ_BYTES, wf[0]), plane(Y1, planes + plane_off[1] * XYZZ_BYTES, wf[1])}, true);
  } else {
    // level 1: A[d2][d1] = sum_{d0} X, Bm[d2][d0] = sum_{d1} X
    if ((rc = ws.rows.ensure((size_t)Wb * ((size_t)n2 * n1 + (size_t)n2 * n0) * XYZZ30_BYTES))) return rc;
    if ((rc = ws.cols.ensure((size_t)Wb * (n0 + n1 + n2) * XYZZ30_BYTES))) return rc;
    uint8_t* A = ws.rows.as<uint8_t>();
    uint8_t* Bm = A + (size_t)Wb * n2 * n1 * XYZZ30_BYTES;
    uint8_t* Y0 = ws.cols.as<uint8_t>();
    uint8_t* Y1 = Y0 + (size_t)Wb * n0 * XYZZ30_BYTES;
    uint8_t* Y2 = Y1 + (size_t)Wb * n1 * XYZZ30_BYTES;
    lpo_jobs = 2;
    launch({strided(X, A, B, n2, n1, n1 * n0, n0, 1, n0), strided(X, Bm, B, n2, n0, n1 * n0, 1, n0, n1)});
    // level 2: Y2[d2] = sum_{d1} A, Y1[d1] = sum_{d2} A, Y0[d0] = sum_{d2} Bm
    lpo_jobs = 3;
    launch({strided(A, Y2, n2 * n1, 1, n2, 0, n1, 1, n1), strided(A, Y1, n2 * n1, 1, n1, 0, 1, n1, n2),
            strided(Bm, Y0, n2 * n0, 1, n0, 0, 1, n0, n2)});
    launch({plane(Y0, planes + plane_off[0] * XYZZ_BYTES, wf[0]), plane(Y1, planes + plane_off[1] * XYZZ_BYTES, wf[1]),
            plane(Y2, planes + plane_off[2] * XYZZ_BYTES, wf[2])}, true);
  }
  pf.end(part, PROF_REDUCE, st);
  GM_HIP(hipGetLastError());
  if (!zc) GM_HIP(hipMemcpyAsync(ws.host_planes[slot], planes, plane_bytes, hipMemcpyDeviceToHost, st));
  GM_HIP(hipEventRecord(ws.done_ev[slot], st));
  P->Wb = Wb;
  P->multi_levels = multi ? multi->levels : 0;
  P->multi_W = multi ? (use_table ? 1 : W) : 0;
  P->plane_count = plane_count;
  P->c = c;
  P->m = m;
  P->nbits = nbits;
  for (int k = 0; k < 3; k++) {
    P->wf[k] = wf[k];
    P->plane_off[k] = k < m ? plane_off[k] : 0;
  }
  return GM_OK;
}

// ------------------------------------------------------------------------------------------
// Host tail.  A lone GPU lane needs ~25 us per dependent group addition, a CPU core ~1 us, so the last
// O(windows x bits) sequential additions run on the host -- and there they are the largest serial piece of a
// call (~0.3 ms for the 256 doublings + 272 additions of 16 windows x 16 bit-planes).  The per-window sums
// S_w = sum_j 2^j Z_{w,j} + Tot_w are independent, so a few persistent helper threads compute them side by
// side; only the window Horner (c doublings + 1 addition per window) stays serial.
// ------------------------------------------------------------------------------------------
class HornerPool {
 public:
  static HornerPool& get() {
    static HornerPool p;
    return p;
  }
  int threads() const { return (int)workers_.size() + 1; }
  // called before the host blocks on the device: the helpers leave their condition variable now (a futex wake
  // costs tens of microseconds) and poll for the job that follows
  void prewake() {
    if (workers_.empty()) return;
    wake_.fetch_add(1, std::memory_order_release);
    cv_.notify_all();
  }
  // fn(i) for i in [0, n), on the helpers and the calling thread; returns when all are done
  void run(int n, const std::function<void(int)>& fn) {
    if (workers_.empty() || n <= 1) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn;
    n_ = n;
    next_.store(0, std::memory_order_relaxed);
    left_.store(n, std::memory_order_relaxed);
    gen_.fetch_add(1, std::memory_order_release);
    lk.unlock();
    cv_.notify_all();
    work();
    while (left_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  }
  ~HornerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }

 private:
  // CPUs this process may actually use: the logical count, the affinity mask and the cgroup CPU quota (a container on a
  // 256-thread host with `cpu.max = 1600000 100000` has 16: polling helpers beyond the quota get the whole process THROTTLED
  // for the rest of a 100 ms period -- seen as 3.7 instead of 3.1 ms per MSM on such boxes with 12 pollers)
  static int effective_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
      const int a = CPU_COUNT(&set);
      if (a > 0 && (n <= 0 || a < n)) n = a;
    }
    auto quota = [](const char* path, const char* path_period) -> double {
      FILE* f = fopen(path, "r");
      if (!f) return 0;
      char a[64] = {0}, b[64] = {0};
      double q = 0;
      if (path_period == nullptr) {  // cgroup v2: "max 100000" or "1600000 100000"
        if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) q = atof(a) / atof(b);
        fclose(f);
        return q;
      }
      long long qu = -1;
      if (fscanf(f, "%lld", &qu) != 1) qu = -1;
      fclose(f);
      FILE* g = fopen(path_period, "r");
      long long per = 0;
      if (g) {
        if (fscanf(g, "%lld", &per) != 1) per = 0;
        fclose(g);
      }
      return (qu > 0 && per > 0) ? (double)qu / (double)per : 0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (q > 0 && (n <= 0 || q < n)) n = (int)(q + 0.5);
    return n > 0 ? n : 1;
  }
  HornerPool() {
    const char* e = getenv("GM_HOST_THREADS");
    const int cpus = effective_cpus();
    // default: 8 threads (12 measure the same, 4 cost 2 %, 1 costs 8 % of a one-call MSM at 2^20), never more than half the CPUs
    // the process may use: the helpers poll for the whole device phase of a call once pre-woken, and one process per GPU shares
    // the host
    int want = e ? atoi(e) : std::min(8, std::max(2, cpus / 2));
    if (want > cpus) want = cpus;
    for (int i = 1; i < want; i++) workers_.emplace_back([this] { loop(); });
  }
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_) break;
      (*fn_)(i);
      left_.fetch_sub(1, std::memory_order_release);
    }
  }
  void loop() {
    uint64_t seen = 0, seen_wake = 0;
    auto budget = std::chrono::microseconds(200);
    for (;;) {
      // spin for the next job -- briefly after a job (MSM calls come back to back inside a prover), for as long
      // as a device call can take after prewake() -- then sleep
      const auto t0 = std::chrono::steady_clock::now();
      bool job = false;
      while (!stop_) {
        if (gen_.load(std::memory_order_acquire) != seen) {
          job = true;
          break;
        }
        if (std::chrono::steady_clock::now() - t0 > budget) {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || wake_.load(std::memory_order_acquire) != seen_wake || stop_; });
          break;
        }
      }
      if (stop_) return;
      if (!job && gen_.load(std::memory_order_acquire) == seen) {  // woken ahead of a job: poll for it
        seen_wake = wake_.load(std::memory_order_acquire);
        budget = std::chrono::microseconds(20000);
        continue;
      }
      seen = gen_.load(std::memory_order_acquire);
      seen_wake = wake_.load(std::memory_order_acquire);
      budget = std::chrono::microseconds(200);
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0}, wake_{0};
  std::atomic<int> next_{0}, left_{0};
  int n_ = 0;
  const std::function<void(int)>* fn_ = nullptr;
  bool stop_ = false;
};

static void horner_pool_prewake() { HornerPool::get().prewake(); }
static void horner_pool_run(int n, const std::function<void(int)>& fn) { HornerPool::get().run(n, fn); }

// use_pool = false: the whole tail on the calling thread (the batch finishes several small calls side by side, ONE per helper
// thread -- a small call's tail is 256 dependent doublings that no pool can split, and a batch of twenty of them was HOST-bound)
static int msm_finish_parts(Context* C, const MsmPending* parts, int nparts, bool normalize, uint64_t out_jac[18], bool use_pool, uint64_t* const* multi_out) {
  gmh::G1 result = gmh::G1::identity();
  // Horner over bit positions, window groups and bucket sets high -> low (variable_base.rs:168-175 with the
  // weighted bucket sum unrolled into its bit-planes): total = sum_w 2^(cw) * (sum_j 2^j Z_{w,j} + Tot_w).
  // The host starts on the high group as soon as its planes have landed; the device is still busy with the
  // low one.
  int rc = GM_OK;
  for (int p = nparts - 1; p >= 0; p--) {
    const MsmPending& P = parts[p];
    if (P.empty) continue;
    MsmWorkspace& ws = *P.ws;
    if (P.Wb > 1 && use_pool) HornerPool::get().prewake();
    GM_HIP(hipEventSynchronize(ws.done_ev[P.slot]));
    const uint64_t* hp = ws.host_planes[P.slot];
    if ((uint32_t)hp[P.plane_count * 24] != 0) {
      set_error("msm: a scalar passed as a canonical integer is >= 2^255 (not the BigInt image of an Fr element)");
      rc = GM_EINVAL;  // keep draining: every part's copy-out must have completed before the workspaces are reused
      continue;
    }
    if (rc) continue;
    auto plane_at = [&](int w, int field, uint32_t j) {
      return gmh::xyzz_to_jac_dev(hp + (P.plane_off[field] + (size_t)w * (P.wf[field] + 1) + j) * 24);
    };
    // S_w = sum_j 2^j Z_{w,j} + Tot_w, one task per bucket set
    std::vector<gmh::G1> S((size_t)P.Wb);
    auto window_sum = [&](int w) {
      gmh::G1 s = gmh::G1::identity();
      int field = P.m - 1;
      uint32_t jj = P.wf[field];
      for (int j = (int)P.nbits - 1; j >= 0; j--) {
        while (jj == 0) {
          field--;
          jj = P.wf[field];
        }
        jj--;
        s = s.dbl();
        s = s.add(plane_at(w, field, jj));
      }
      gmh::G1 tot = plane_at(w, 0, P.wf[0]);  // the low field's elements with bit 0 clear ...
      if (P.wf[0] >= 1) tot = tot.add(plane_at(w, 0, 0));  // ... + those with bit 0 set = Tot_w
      S[(size_t)w] = s.add(tot);
    };
    if (use_pool) HornerPool::get().run(P.Wb, window_sum);
    else
      for (int w = 0; w < P.Wb; w++) window_sum(w);
    if (P.multi_levels > 0) {  // fused small calls: bucket sets [l W, (l + 1) W) are the windows of call l
      if (!multi_out || nparts != 1) return GM_ESTATE;
      auto level = [&](int l) {
        gmh::G1 r = gmh::G1::identity();
        for (int w = P.multi_W - 1; w >= 0; w--) {
          for (int j = 0; j < P.c; j++) r = r.dbl();
          r = r.add(S[(size_t)(l * P.multi_W + w)]);
        }
        if (normalize) r = r.normalized();
        r.to_limbs(multi_out[l]);
      };
      if (use_pool) HornerPool::get().run(P.multi_levels, level);
      else
        for (int l = 0; l < P.multi_levels; l++) level(l);
      return GM_OK;
    }
    for (int w = P.Wb - 1; w >= 0; w--) {
      for (int j = 0; j < P.c; j++) result = result.dbl();
      result = result.add(S[(size_t)w]);
    }
  }
  if (use_pool) C->prof.collect();
  if (C->prof.on && use_pool)  // the clock readings of the call's k_acc0 travelled with the plane sums
    for (int p = 0; p < nparts; p++)
      if (!parts[p].empty) {
        const uint64_t* hp = parts[p].ws->host_planes[parts[p].slot] + parts[p].plane_count * 24;
        C->prof.acc0_cycles += (double)hp[1];
        C->prof.acc0_ticks += (double)hp[2];
      }
  if (rc) return rc;
  if (normalize) result = result.normalized();
  result.to_limbs(out_jac);
  return GM_OK;
}

// ---- bases management -----------------------------------------------------------------------
int bases_from_host(Context* C, const void* bases, size_t stride, size_t n, std::unique_ptr<Bases>& out) {
  GM_CHECK(stride >= 96 && (stride % 8) == 0, GM_EINVAL, "bases: stride %zu must be >= 96 and a multiple of 8", stride);
  auto b = std::make_unique<Bases>();
  b->n = n;
  if (n) {
    GM_HIP(dev_malloc((void**)&b->d, n * AFF_BYTES));
    {
      uint8_t* stage = nullptr;
      GM_HIP(dev_malloc((void**)&stage, n * stride));
      GM_HIP(hipMemcpyAsync(stage, bases, n * stride, hipMemcpyHostToDevice, C->stream));
      hipLaunchKernelGGL(k_pack_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, C->stream, stage, stride, n, b->d);
      GM_HIP(hipStreamSynchronize(C->stream));
      GM_HIP(gm::raw_free(stage));
    }
    GM_HIP(hipStreamSynchronize(C->stream));
  }
  out = std::move(b);
  return GM_OK;
}

// gm_g1_bases_register / fixed_base_register / srs_register, when gm_set_msm_glv(1) is in force (off by default:
// measured below): the GLV images next to the bases (one product per point, +96 bytes per point).  Calls on bases
// without them take the one-string path.
//   MI355X, one-call MSMs, GLV vs plain (ms): n = 2: 0.31 / 0.36, 2^12: 0.59 / 0.63, 2^16: 1.57 / 1.04, 2^18: 1.44 / 1.52,
//   2^20: 3.91 / 3.67 (3.70 with L = 128), 2^22: 13.2 / 12.7, 2^24: 42.6 / 43.3.  Half the bucket sets halve k_group_sum's
//   level 1 (0.45 -> 0.35 ms at 2^20) and the host's doublings, but the buckets are twice as dense, k_merge pays for
//   it (0.16 -> 0.52 ms at the tuned chunk length), the gathers touch two arrays (k_acc0 +1.8 %) and the key doubles in
//   HBM: no net gain where it matters.  Results are identical (tests/test_gpu_msm.py::test_msm_glv_same_results).
int bases_build_phi(Context* C, Bases* b) {
#ifndef GM_EXPERIMENTS
  (void)C;
  (void)b;
  return GM_OK;  // the GLV addressing mode is not in this build (msm_glv.inc, -DGM_EXPERIMENTS)
#else
  static const bool glv_env = getenv("GM_GLV") && !strcmp(getenv("GM_GLV"), "1");
  if (!(glv_env || C->msm_glv) || b->n == 0 || b->phi) return GM_OK;
  GM_HIP(dev_malloc((void**)&b->phi, b->n * AFF_BYTES));
  hipLaunchKernelGGL(k_phi_bases, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, C->stream, b->d, b->n, b->phi);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
#endif
}

int bases_export(Context* C, const Bases* b, size_t offset, size_t n, void* out96) {
  if (n == 0) return GM_OK;
  uint8_t* tmp = nullptr;
  GM_HIP(dev_malloc((void**)&tmp, n * AFF_BYTES));
  hipLaunchKernelGGL(k_export_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, C->stream, b->d + offset * AFF_BYTES, n, tmp);
  GM_HIP(hipGetLastError());
  GM_HIP(hipMemcpyAsync(out96, tmp, n * AFF_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  GM_HIP(gm::raw_free(tmp));
  return GM_OK;
}

// fixed-base table (affine, 32 x 256 entries) for `base`
static int build_fixed_table(Context* C, const uint64_t base_affine[12], uint8_t** table_aff) {
  uint8_t *d_base = nullptr, *t_xyzz = nullptr;
  GM_HIP(dev_malloc((void**)&d_base, AFF_BYTES));
  GM_HIP(dev_malloc((void**)&t_xyzz, (size_t)32 * 256 * XYZZ_BYTES));
  GM_HIP(dev_malloc((void**)table_aff, (size_t)32 * 256 * AFF_BYTES));
  {
    uint8_t* stage = nullptr;
    GM_HIP(dev_malloc((void**)&stage, AFF_BYTES));
    GM_HIP(hipMemcpyAsync(stage, base_affine, AFF_BYTES, hipMemcpyHostToDevice, C->stream));
    hipLaunchKernelGGL(k_pack_bases, dim3(1), dim3(64), 0, C->stream, stage, (size_t)AFF_BYTES, (size_t)1, d_base);
    GM_HIP(hipStreamSynchronize(C->stream));
    GM_HIP(gm::raw_free(stage));
  }
  hipLaunchKernelGGL(k_fixed_base_table, dim3(32), dim3(256), 0, C->stream, d_base, t_xyzz);
  hipLaunchKernelGGL(k_xyzz_to_affine, dim3(32), dim3(256), 0, C->stream, t_xyzz, (size_t)32 * 256, *table_aff);
  GM_HIP(hipStreamSynchronize(C->stream));
  GM_HIP(gm::raw_free(d_base));
  GM_HIP(gm::raw_free(t_xyzz));
  return GM_OK;
}

// ---- herring TimeProver over G1Module (src/herring/time_prover.rs:42-137, module.rs:81-102) ------
int fr_stride_raw(Context* C, const uint8_t* in, size_t start, size_t stride, size_t count, uint8_t* out);
int fr_fold_raw(Context* C, const uint8_t* f, size_t n, const uint64_t r[4], uint8_t* out);

// ------------------------------------------------------------------------------------------
// ChunkedPippenger / msm_chunks over HOST-resident pairs: bounded device memory
//
// The reference's streaming MSMs (ChunkedPippenger, src/kzg/msm/stream_pippenger.rs:209-272; msm_chunks,
// src/kzg/space.rs:22-55) collect `max_msm_buffer` pairs from the streams, run one MSM, add it to the running
// sum and start over: memory is O(buffer), not O(stream).  Here the buffer is a pair of device slots of
// `chunk` pairs each.  Pairs arrive from host memory in blocks of any size (msm_stream_add); they are copied
// straight into the current slot on a copy stream, and when the slot is full its MSM is enqueued on one of the
// two full-size lanes while the host goes on copying the next block into the other slot -- H2D copy of chunk
// i + 1 under the kernels of chunk i, host Horner of chunk i - 1 under both.  The sum does not depend on where
// the stream is cut (tests/test_gpu_msm_stream.py), so results equal the one-call MSM of the whole stream.
// Device memory: 2 slots x chunk x (stride + 96 + 32) bytes + the MSM workspace of one chunk per lane; the
// stream itself (SRS and polynomial larger than HBM) stays on the host.
//
// No MSM stays in flight on the shared lane workspaces when an API call returns (another thread may use them
// next): add() drains what it enqueued before returning, so the overlap is within one add() call -- push
// blocks of several chunks.  Pinned host buffers (gm_host_alloc) are copied by DMA; pageable ones go through
// the runtime's staging copy at a fraction of the PCIe rate.
// ------------------------------------------------------------------------------------------
static void msm_stream_reset(MsmStream* S) {
  S->acc_set = false;
  S->fill = 0;
  S->total = 0;
  S->cur = 0;
  S->next_base = S->base0;
}
static int msm_stream_drain(Context* C, MsmStream* S, MsmStreamSlot& sl) {
  if (!sl.inflight) return GM_OK;
  sl.inflight = false;
  uint64_t part[18];
  int rc = msm_finish(C, sl.P, false, part);
  if (rc) return rc;
  if (!S->acc_set) {
    memcpy(S->acc, part, sizeof(part));
    S->acc_set = true;
  } else {
    gmh::G1::from_limbs(S->acc).add(gmh::G1::from_limbs(part)).to_limbs(S->acc);
  }
  return GM_OK;
}

static int msm_stream_flush(Context* C, MsmStream* S) {
  MsmStreamSlot& sl = S->s[S->cur];
  const size_t m = S->fill;
  if (m == 0) return GM_OK;
  MsmWorkspace& ws = S->cur == 0 ? C->msm : C->msm_b;
  hipStream_t st = S->cur == 0 ? C->stream : C->stream_b;
  Bases own;
  const Bases* b = &own;
  int64_t first = 0, step = 1;
  if (S->bases_handle) {
    b = find_bases(S->bases_handle);
    GM_CHECK(b != nullptr, GM_EHANDLE, "msm_stream: the bases (handle %llu) were freed under the stream", (unsigned long long)S->bases_handle);
    first = S->next_base;
    step = S->step;
    S->next_base += step * (int64_t)m;
  } else {
    hipLaunchKernelGGL(k_pack_bases, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, S->copy, sl.raw, S->stride, m, sl.packed);
    own.d = sl.packed;
    own.n = m;
  }
  GM_HIP(hipEventRecord(sl.copied, S->copy));
  GM_HIP(hipStreamWaitEvent(st, sl.copied, 0));
  int rc = msm_enqueue(C, ws, MsmStreams{st, st, st}, b, first, step, sl.scalars, S->mont, m, 0, &sl.P);
  if (rc) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  sl.inflight = true;
  S->total += m;
  S->fill = 0;
  S->cur ^= 1;
  // the stage timers hold one call at a time
  if (C->prof.on) return msm_stream_drain(C, S, sl);
  return GM_OK;
}

int msm_stream_create(Context* C, uint64_t bases_handle, size_t offset, int reversed, size_t chunk, size_t stride, int mont, uint64_t* handle) {
  GM_CHECK(chunk >= 1 && chunk <= ((size_t)1 << 26), GM_EINVAL, "msm_stream: chunk of %zu pairs outside [1, 2^26]", chunk);
  GM_CHECK(stride >= 96 && (stride % 8) == 0, GM_EINVAL, "msm_stream: stride %zu must be >= 96 and a multiple of 8", stride);
  auto S = std::make_unique<MsmStream>();
  S->chunk = chunk;
  S->stride = stride;
  S->mont = mont ? 1 : 0;
  S->bases_handle = bases_handle;
  S->base0 = S->next_base = (int64_t)offset;
  S->step = reversed ? -1 : 1;
  GM_HIP(hipStreamCreateWithFlags(&S->copy, hipStreamNonBlocking));
  std::lock_guard<std::mutex> lk(C->mu);
  *handle = C->next_handle++;
  C->msm_streams[*handle] = std::move(S);
  return GM_OK;
}

void msm_stream_destroy(Context* C, MsmStream* S) {
  if (S->copy) (void)hipStreamSynchronize(S->copy);
  for (auto& sl : S->s) {
    if (sl.inflight) {
      uint64_t part[18];
      (void)msm_finish(C, sl.P, false, part);
    }
    if (C) {
      C->pool.free(sl.raw, sl.raw_cap);
      C->pool.free(sl.packed, sl.packed_cap);
      C->pool.free(sl.scalars, sl.scalars_cap);
    }
    if (sl.have_ev) (void)hipEventDestroy(sl.copied);
  }
  if (S->copy) (void)hipStreamDestroy(S->copy);
}

int msm_stream_add(Context* C, MsmStream* S, const void* bases_host, const void* scalars_host, size_t n) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_MSM_LOCK(C);
  const uint8_t* bp = reinterpret_cast<const uint8_t*>(bases_host);
  const uint8_t* sp = reinterpret_cast<const uint8_t*>(scalars_host);
  const bool own_bases = S->bases_handle == 0;
  int rc = GM_OK;
  while (n && !rc) {
    MsmStreamSlot& sl = S->s[S->cur];
    if ((rc = msm_stream_drain(C, S, sl))) break;  // the MSM that reads this slot's buffers
    // a failing runtime call must not return from inside the loop: the drain + reset below are what keeps the invariant
    // "nothing of this stream stays on the shared lanes" when an MSM of an earlier slot is still in flight
    auto hip_rc = [&](hipError_t e, const char* what) { return e == hipSuccess ? GM_OK : hip_fail(e, what, __FILE__, __LINE__); };
    if (!sl.have_ev) {
      if ((rc = hip_rc(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming), "hipEventCreateWithFlags(stream slot)"))) break;
      sl.have_ev = true;
    }
    if (!sl.scalars) {
      if ((rc = C->pool.alloc(S->chunk * 32 + 32, (void**)&sl.scalars, &sl.scalars_cap))) break;
      if (own_bases) {
        if ((rc = C->pool.alloc(S->chunk * S->stride, (void**)&sl.raw, &sl.raw_cap))) break;
        if ((rc = C->pool.alloc(S->chunk * AFF_BYTES, (void**)&sl.packed, &sl.packed_cap))) break;
      }
    }
    const size_t take = std::min(n, S->chunk - S->fill);
    if (own_bases &&
        (rc = hip_rc(hipMemcpyAsync(sl.raw + S->fill * S->stride, bp, take * S->stride, hipMemcpyHostToDevice, S->copy), "hipMemcpyAsync(stream bases)")))
      break;
    if ((rc = hip_rc(hipMemcpyAsync(sl.scalars + S->fill * 32, sp, take * 32, hipMemcpyHostToDevice, S->copy), "hipMemcpyAsync(stream scalars)"))) break;
    S->fill += take;
    n -= take;
    if (own_bases) bp += take * S->stride;
    sp += take * 32;
    if (S->fill == S->chunk) rc = msm_stream_flush(C, S);
  }
  // the caller's buffers are free again on return, and nothing of this stream stays on the shared lanes
  (void)hipStreamSynchronize(S->copy);
  for (auto& sl : S->s) {
    int r2 = msm_stream_drain(C, S, sl);
    if (!rc) rc = r2;
  }
  if (rc) msm_stream_reset(S);  // a failed add leaves an empty stream, not a partial sum
  return rc;
}

int msm_stream_finalize(Context* C, MsmStream* S, uint64_t out_jac[18], size_t* pairs) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_MSM_LOCK(C);
  int rc = msm_stream_flush(C, S);
  for (auto& sl : S->s) {
    int r2 = msm_stream_drain(C, S, sl);
    if (!rc) rc = r2;
  }
  if (!rc) {
    gmh::G1 r = S->acc_set ? gmh::G1::from_limbs(S->acc).normalized() : gmh::G1::identity();
    r.to_limbs(out_jac);
    if (pairs) *pairs = S->total;
  }
  msm_stream_reset(S);  // ready for the next stream (the reference's finalize consumes the object)
  return rc;
}

void hg1_destroy(Context* C, HerringG1* H);
int hg1_create(Context* C, const void* f_bases, size_t stride, size_t nf, const uint64_t* g_mont, size_t ng, const uint64_t twist[4],
               uint64_t* handle) {
  GM_CHECK(nf >= 1 && ng >= 1, GM_EINVAL, "herring G1 prover: empty vectors");
  std::unique_ptr<Bases> b;
  int rc = bases_from_host(C, f_bases, stride, nf, b);
  if (rc) return rc;
  auto H = std::make_unique<HerringG1>();
  H->nf = nf;
  H->ng = ng;
  H->f[0] = b->d;  // take ownership of the packed copy
  b->d = nullptr;
  auto fail = [&](int code) {  // what has been allocated so far goes back
    hg1_destroy(C, H.get());
    return code;
  };
  {
    hipError_t e = dev_malloc((void**)&H->f[1], ((nf + 1) / 2) * AFF_BYTES);
    if (e != hipSuccess) return fail(hip_fail(e, "dev_malloc(herring G1 fold buffer)", __FILE__, __LINE__));
  }
  if ((rc = C->pool.alloc(ng * 32, (void**)&H->g[0], &H->gcap[0]))) return fail(rc);
  if ((rc = C->pool.alloc(((ng + 1) / 2) * 32, (void**)&H->g[1], &H->gcap[1]))) return fail(rc);
  if ((rc = C->pool.alloc(3 * ((((ng + 1) / 2) + 1) * 32), (void**)&H->tmp, &H->tmpcap))) return fail(rc);  // three compacted scalar vectors
  {
    hipError_t e = hipMemcpyAsync(H->g[0], g_mont, ng * 32, hipMemcpyHostToDevice, C->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpyAsync(herring G1 scalars)", __FILE__, __LINE__));
  }
  memcpy(H->twist, twist, 32);
  size_t mn = nf < ng ? nf : ng;
  H->tot_rounds = (size_t)ceil_log2_sz(mn);  // Witness::required_rounds: log2(min(len)) (time_prover.rs:36-39)
  std::lock_guard<std::mutex> lk(C->mu);
  *handle = C->next_handle++;
  C->herring_g1[*handle] = std::move(H);
  return GM_OK;
}

void hg1_destroy(Context* C, HerringG1* H) {
  for (int i = 0; i < 2; i++) {
    if (H->f[i]) (void)gm::raw_free(H->f[i]);
    if (C) C->pool.free(H->g[i], H->gcap[i]);
  }
  if (C) C->pool.free(H->tmp, H->tmpcap);
}

static int hg1_fold_locked(Context* C, HerringG1* H, const uint64_t r[4]) {
  GM_MSM_LOCK(C);  // the folding scalar is staged in the MSM workspace (C->msm.misc)
  gmh::Fr rr = gmh::Fr::from_limbs(r), tw = gmh::Fr::from_limbs(H->twist);
  gmh::Fr rt = rr * tw;
  uint64_t canon[4];
  rt.to_canonical(canon);  // scalar multiplication wants the integer
  int rc = C->msm.misc.ensure(64);
  if (rc) return rc;
  GM_HIP(hipMemcpyAsync(C->msm.misc.p, canon, 32, hipMemcpyHostToDevice, C->stream));
  const size_t m = (H->nf + 1) / 2;
  hipLaunchKernelGGL(k_g1_split_fold, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, C->stream, H->f[H->cur], H->nf,
                     C->msm.misc.as<uint32_t>(), H->f[H->cur ^ 1]);
  GM_HIP(hipGetLastError());
  if ((rc = fr_fold_raw(C, H->g[H->cur], H->ng, r, H->g[H->cur ^ 1]))) return rc;
  H->cur ^= 1;
  H->nf = m;
  H->ng = (H->ng + 1) / 2;
  tw.sqr().to_limbs(H->twist);
  return GM_OK;
}

int hg1_fold(Context* C, HerringG1* H, const uint64_t r[4]) {
  std::lock_guard<std::mutex> lk(H->mu);
  return hg1_fold_locked(C, H, r);
}

// next_message: a = <f_even, g_even>, b = <f_even, g_odd> + <f_odd, g_even>, each an MSM (module.rs:91-101)
int hg1_round(Context* C, HerringG1* H, const uint64_t* challenge, uint64_t a_jac[18], uint64_t b_jac[18], int* has_msg) {
  std::lock_guard<std::mutex> lk(H->mu);
  GM_CHECK(H->round <= H->tot_rounds, GM_ESTATE, "More rounds than needed.");
  int rc;
  if (challenge && (rc = hg1_fold_locked(C, H, challenge))) return rc;
  if (H->round == H->tot_rounds) {
    *has_msg = 0;
    return GM_OK;
  }
  Bases fb;
  fb.d = H->f[H->cur];
  fb.n = H->nf;
  const uint8_t* g = H->g[H->cur];
  const size_t fe = (H->nf + 1) / 2, fo = H->nf / 2, ge = (H->ng + 1) / 2, go = H->ng / 2;
  // a = <f_even, g_even>, b = <f_even, g_odd> + <f_odd, g_even>: three strided MSMs over the same point array, issued
  // as ONE batch (these are small calls -- 2^10 points in the reference's tests -- and run side by side on the small lanes)
  const size_t slot = (ge + 1) * 32;  // bytes per compacted scalar vector inside H->tmp
  const int64_t firsts[3] = {0, 0, 1};
  const size_t g_first[3] = {0, 1, 0};
  const size_t fcount[3] = {fe, fe, fo}, gcount[3] = {ge, go, ge};
  size_t cnts[3];
  const void* sc[3];
  for (int j = 0; j < 3; j++) {
    cnts[j] = fcount[j] < gcount[j] ? fcount[j] : gcount[j];  // zip
    uint8_t* dst = H->tmp + (size_t)j * slot;
    if ((rc = fr_stride_raw(C, g, g_first[j], 2, cnts[j], dst))) return rc;
    sc[j] = dst;
  }
  uint64_t res[3 * 18];
  if ((rc = msm_run_batch_at(C, &fb, 0, 2, nullptr, sc, 1, cnts, 3, true, res, firsts))) return rc;
  memcpy(a_jac, res, 18 * sizeof(uint64_t));
  const uint64_t *b1 = res + 18, *b2 = res + 36;
  gmh::G1 bsum = gmh::G1::from_limbs(b1).add(gmh::G1::from_limbs(b2)).normalized();
  bsum.to_limbs(b_jac);
  H->round += 1;
  *has_msg = 1;
  return GM_OK;
}

int hg1_final(Context* C, HerringG1* H, uint64_t f0_jac[18], uint64_t g0[4], int* has) {
  std::lock_guard<std::mutex> lk(H->mu);
  if (H->round != H->tot_rounds) {
    *has = 0;
    return GM_OK;
  }
  uint64_t aff[12];
  GM_HIP(hipMemcpyAsync(aff, H->f[H->cur], AFF_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipMemcpyAsync(g0, H->g[H->cur], 32, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  gmh::G1 p = gmh::G1::identity();
  bool zero = true;
  for (int i = 0; i < 12; i++) zero &= aff[i] == 0;
  if (!zero) {
    p.x = gmh::fq_from_device(aff);
    p.y = gmh::fq_from_device(aff + 6);
    p.z = gmh::Fq::one();
  }
  p.to_limbs(f0_jac);
  *has = 1;
  return GM_OK;
}

// table[w * n + i] = 2^(c w) * d[i], w < ceil(256 / c); row 0 is a copy of d
static int build_window_table(Context* C, const uint8_t* d, size_t n, int c, uint8_t** out) {
  const int W = (256 + c - 1) / c;
  uint8_t* t = nullptr;
  GM_HIP(dev_malloc((void**)&t, (size_t)W * n * AFF_BYTES));
  GM_HIP(hipMemcpyAsync(t, d, n * AFF_BYTES, hipMemcpyDeviceToDevice, C->stream));
  // one slab of XYZZ results at a time (192 B per point), normalised NORM_K points per inversion
  const size_t slab = std::min<size_t>(n, (size_t)1 << 22);
  uint8_t* xy = nullptr;
  {
    const hipError_t e = dev_malloc((void**)&xy, slab * XYZZ_BYTES);
    if (e != hipSuccess) (void)gm::raw_free(t);
    GM_HIP(e);
  }
  for (int w = 0; w + 1 < W; w++)
    for (size_t off = 0; off < n; off += slab) {
      const size_t m = std::min(slab, n - off);
      hipLaunchKernelGGL(k_table_next, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, C->stream,
                         t + ((size_t)w * n + off) * AFF_BYTES, xy, m, c);
      hipLaunchKernelGGL(k_xyzz_to_affine_batch, dim3((unsigned)(((m + NORM_K - 1) / NORM_K + 255) / 256)), dim3(256), 0, C->stream, xy, m,
                         t + ((size_t)(w + 1) * n + off) * AFF_BYTES);
    }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
  (void)gm::raw_free(xy);
  if (e != hipSuccess) (void)gm::raw_free(t);
  GM_HIP(e);
  *out = t;
  return GM_OK;
}

void bases_free_tables(Bases* b) {
  if (b->table) (void)gm::raw_free(b->table);
  b->table = nullptr;
  for (auto& ts : b->extra)
    if (ts.t) (void)gm::raw_free(ts.t);
  b->extra.clear();
}

// gm_g1_bases_precompute
int bases_precompute(Context* C, Bases* b, int c) {
  // automatic width (measured): c = 20 (13 windows, 2^19 shared buckets) wins from 2^17 pairs on -- 4.14 vs 4.76 ms at
  // 2^20 --; a key of >= 2^23 points mostly serves big calls, where c = 22 (12 windows, 2^21 buckets) gives
  // 43.8 vs 47.8 (c = 20) vs 51.5 ms (no tables) at 2^24 but loses below 2^22 pairs to its bucket reduction
  const bool auto_c = c == 0;
  // a key too long for tables of its own (the pair index of a table entry has 26 bits) still gets PREFIX tables (automatic mode
  // only): half of the pairs of a proof are folding levels, and they walk the first powers
  const bool prefix_only = auto_c && b->n >= ((size_t)1 << ENTRY_W_SHIFT);
  if (auto_c) c = b->n >= ((size_t)1 << 23) ? 22 : 20;
  GM_CHECK(c >= 8 && c <= 22, GM_EINVAL, "bases_precompute: window %d outside [8, 22]", c);
  GM_CHECK(b->n >= 1 && (prefix_only || b->n < ((size_t)1 << ENTRY_W_SHIFT)), GM_EINVAL, "bases_precompute: %zu bases (need 1 .. 2^26 - 1)", b->n);
  const int W = (256 + c - 1) / c;
  GM_MSM_LOCK(C);
  bases_free_tables(b);
  int rc = GM_OK;
  if (!prefix_only) {
    uint8_t* t = nullptr;
    if ((rc = build_window_table(C, b->d, b->n, c, &t))) return rc;
    b->table = t;
    b->tab_c = c;
    b->tab_W = W;
    b->tab_min = c >= 22 ? ((size_t)1 << 22) : (c >= 21 ? ((size_t)1 << 21) : 0);
  }
  if (!auto_c) return GM_OK;
  return bases_build_prefix_sets(C, b);
}

// the PREFIX table sets of a key in automatic mode (also on their own: gm_g1_bases_precompute(handle, -1) after the sets were
// released under memory pressure, capi.hip: release_spare_tables)
int bases_build_prefix_sets(Context* C, Bases* b) {
  GM_MSM_LOCK(C);
  const bool prefix_only = b->table == nullptr && b->n >= ((size_t)1 << ENTRY_W_SHIFT);
  int rc = GM_OK;
  for (auto& ts : b->extra)
    if (ts.t) (void)gm::raw_free(ts.t);
  b->extra.clear();
  b->extras_released = false;
  static const bool prefix_env = !(getenv("GM_PREFIX_TABLES") && atoi(getenv("GM_PREFIX_TABLES")) == 0);
  auto add_set = [&](int sc, size_t points, size_t min_n, size_t max_n) -> int {
    Bases::TableSet ts;
    ts.c = sc;
    ts.W = (256 + sc - 1) / sc;
    ts.n = std::min<size_t>(b->n, points);
    ts.min_n = min_n;
    ts.max_n = max_n;
    const int r = build_window_table(C, b->d, ts.n, ts.c, &ts.t);
    if (r == GM_OK) b->extra.push_back(ts);
    return r == GM_ENOMEM ? GM_OK : r;  // no room: those calls keep the plain path
  };
  // (GM_PREFIX_TABLES=0: none of the two)
  // a long key: c = 22 over the first 2^25 points (38.7 GB) for the calls of 2^22 .. 2^25 pairs inside them
  if (prefix_only && prefix_env && (rc = add_set(22, (size_t)1 << 25, (size_t)1 << 22, ((size_t)1 << 25) + 1))) return rc;
  // the calls below the range of a c = 22 table: c = 20 over the first 2^22 points (5.2 GB)
  if ((prefix_only || b->tab_min > 0) && prefix_env && (rc = add_set(20, (size_t)1 << 22, (size_t)1 << 17, (size_t)1 << 22))) return rc;
  // small calls (GM_SMALL_TABLE_C=0: none): a c = 16 table over the first 2^17 points (201 MB) -- the same 16 additions per pair as
  // the plain path, but ONE bucket set (2^15 instead of 16 x 2^15 buckets to reduce) and 16 instead of 256 final doublings on the
  // host, for the latency-bound calls of 2^11 .. 2^17 - 1 pairs at the end of a folding tree.  One call at 2^14 pairs 0.72 -> 0.61 ms,
  // snark -i 20 14.2 -> 13.6 ms, psnark -i 18 54.5 -> 53.0 ms; c = 12 / 14 / 17 / 18 lose (profiles/r4_small_tables_probe.txt)
  static const int small_c = getenv("GM_SMALL_TABLE_C") ? atoi(getenv("GM_SMALL_TABLE_C")) : 16;
  static const int small_min_log = getenv("GM_SMALL_TABLE_MIN") ? atoi(getenv("GM_SMALL_TABLE_MIN")) : 11;
  if (small_c >= 8 && small_c <= 20 && (rc = add_set(small_c, (size_t)1 << 17, (size_t)1 << small_min_log, (size_t)1 << 17))) return rc;
  return GM_OK;
}

int fixed_base_generate(Context* C, const uint64_t base_affine[12], const void* d_scalars, int mont, size_t n,
                        std::unique_ptr<Bases>& out) {
  auto b = std::make_unique<Bases>();
  b->n = n;
  if (n) {
    uint8_t* table = nullptr;
    int rc = build_fixed_table(C, base_affine, &table);
    if (rc) return rc;
    GM_HIP(dev_malloc((void**)&b->d, n * AFF_BYTES));
    const size_t slab = std::min<size_t>(n, (size_t)1 << 22);
    uint8_t* xy = nullptr;
    GM_HIP(dev_malloc((void**)&xy, slab * XYZZ_BYTES));
    for (size_t off = 0; off < n; off += slab) {
      const size_t m = std::min(slab, n - off);
      hipLaunchKernelGGL(k_fixed_base_mul, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, C->stream,
                         reinterpret_cast<const uint32_t*>(d_scalars) + 8 * off, mont, m, table, xy);
      hipLaunchKernelGGL(k_xyzz_to_affine_batch, dim3((unsigned)(((m + NORM_K - 1) / NORM_K + 255) / 256)), dim3(256), 0, C->stream, xy, m,
                         b->d + off * AFF_BYTES);
    }
    GM_HIP(hipGetLastError());
    GM_HIP(hipStreamSynchronize(C->stream));
    GM_HIP(gm::raw_free(xy));
    GM_HIP(gm::raw_free(table));
  }
  out = std::move(b);
  return GM_OK;
}

}  // namespace gm

#ifdef GM_ACC0_CYCLES
// development build only (make EXTRA=-DGM_ACC0_CYCLES): shader cycles / 100 MHz ticks / waves summed over the k_acc0 launches
// since the last reset -- tools/acc0_cycles.py turns them into the clock and the cycles per entry of the kernel itself
extern "C" int gm_debug_acc0_cycles(uint64_t out[3], int reset) {
  unsigned long long h[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gm::gm_acc0_dbg), sizeof h) != hipSuccess) return 1;
  out[0] = h[0];
  out[1] = h[1];
  out[2] = h[2];
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gm::gm_acc0_dbg), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif

// Scalar-field (Fr) kernels: the sumcheck time prover and the dense vector passes of src/misc.rs.
//
// All vectors are arrays of 32-byte Montgomery elements (the ark-ff memory image), read and
// written with 16-byte vector accesses; lane i touches element i of a wave-contiguous tile so
// a wave moves 2 KiB per load instruction.  These kernels are HBM-streaming work with ~10
// modular multiplications per 384 bytes moved (sumcheck round); the arithmetic is the
// 8-limb product-scanning multiplier of field.cuh.
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "field.cuh"
#include "host_field.hpp"

namespace gm {

constexpr int FR_BYTES = 32;

__device__ __noinline__ Fr fr_mul_fn(const Fr a, const Fr b) { return fp_mul<FrParams>(a, b); }
GM_DEV Fr fr_mul(const Fr& a, const Fr& b) { return fp_mul<FrParams>(a, b); }
GM_DEV Fr fr_add(const Fr& a, const Fr& b) { return fp_add<FrParams>(a, b); }
GM_DEV Fr fr_sub(const Fr& a, const Fr& b) { return fp_sub<FrParams>(a, b); }

GM_DEV Fr fr_load_or_zero(const uint8_t* base, size_t i, size_t n) {
  if (i < n) return fp_load<FrParams>(base + i * FR_BYTES);
  return Fr::zero();
}

// table of x^(2^b), b < 40, passed by value to kernels that need x^k for per-thread k
struct PowTable {
  uint32_t p[40][8];
};
GM_DEV Fr pow_from_table(const PowTable& t, uint64_t e) {
  Fr acc = Fr::one();
  for (int b = 0; b < 40; b++) {
    if ((e >> b) & 1ull) {
      Fr m;
#pragma unroll
      for (int i = 0; i < 8; i++) m.l[i] = t.p[b][i];
      acc = fr_mul(acc, m);
    }
  }
  return acc;
}

GM_DEV Fr fr_shfl_down(const Fr& v, int d) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], d);
  return r;
}
GM_DEV Fr wave_sum(Fr v) {
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) v = fr_add(v, fr_shfl_down(v, d));
  return v;
}
// block-wide sum of K values per thread; result valid in thread 0.  blockDim = 256.
template <int K>
GM_DEV void block_sum(Fr (&v)[K], uint8_t* lds /* 4 * K * 32 bytes */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) fp_store<FrParams>(lds + (wave * K + k) * FR_BYTES, v[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      Fr s = fp_load<FrParams>(lds + k * FR_BYTES);
      for (int w = 1; w < 4; w++) s = fr_add(s, fp_load<FrParams>(lds + (w * K + k) * FR_BYTES));
      v[k] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// sumcheck round: optional fold (time_prover.rs:75-80) fused with the next message (:83-123)
// ------------------------------------------------------------------------------------------
struct ScArgs {
  const uint8_t* f_in;
  const uint8_t* g_in;
  uint8_t* f_out;
  uint8_t* g_out;
  size_t nf_in, ng_in;  // lengths of the input vectors
  uint32_t rho_tau[8];  // rho * twist   (fold multiplier of f)
  uint32_t rho[8];      // rho           (fold multiplier of g)
  uint32_t tau[8];      // twist the MESSAGE is computed with (already squared when folding)
  uint32_t origin[8];   // tau^(2 * first pair index of this shard)
  PowTable tau2;        // (tau^2)^(2^b)
  uint32_t log_threads; // total threads = 2^log_threads
  size_t npairs;        // loop bound: pairs of the message vectors (covers both vectors entirely)
};

// LAZY = the three inner products of the message accumulate UNREDUCED 512-bit products (fp_mac_wide: the product half of a
// multiplication, 80 multiply-adds instead of 137) in 17-limb accumulators and are Montgomery-reduced ONCE per thread (r^2 =
// 0.205 x 2^512: the seventeenth limb holds what 512 bits cannot).  The reference does the same on the CPU (`ip_unsafe`,
// src/misc.rs:235-266).
// TW1 = the message's twist is ONE (Sumcheck::new_time(.., &F::one()): the second sumcheck of both SNARKs, three of the thirteen provers of the
// third; herring's module provers): the running twist power is 1 for every pair, so the two products with it and its own update -- three of the
// ~ nine multiplications a fold + message pair costs in this ALU-bound kernel -- are not computed at all
template <bool FOLD, bool MSG, bool LAZY, bool TW1>
GM_DEV void sc_round_body(const ScArgs& A, uint8_t* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 3 * FR_BYTES];
  const size_t T = (size_t)1 << A.log_threads;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr rho_tau, rho, tau, tw, step;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    rho_tau.l[i] = A.rho_tau[i];
    rho.l[i] = A.rho[i];
    tau.l[i] = A.tau[i];
    tw.l[i] = A.origin[i];
    step.l[i] = A.tau2.p[A.log_threads][i];
  }
  Fr acc[3] = {Fr::zero(), Fr::zero(), Fr::zero()};  // a, b1 = sum fe*go*tw, b2 = sum ge*fo*tw
  FpWide_FrParams wide[3];  // 17 limbs each: room for 2^34 products of a thread (it sees 2^11 at most)
  if (LAZY) {
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int i = 0; i < 17; i++) wide[k].l[i] = 0;
  }
  if (MSG && !TW1 && t < A.npairs) tw = fr_mul(tw, pow_from_table(A.tau2, t));
  // message vectors: FOLD ? folded (length ceil(n_in/2)) : the inputs
  const size_t nf = FOLD ? (A.nf_in + 1) / 2 : A.nf_in;
  const size_t ng = FOLD ? (A.ng_in + 1) / 2 : A.ng_in;
  for (size_t j = t; j < A.npairs; j += T) {
    Fr fe, fo, ge, go;
    if (FOLD) {
      Fr f0 = fr_load_or_zero(A.f_in, 4 * j, A.nf_in), f1 = fr_load_or_zero(A.f_in, 4 * j + 1, A.nf_in);
      Fr f2 = fr_load_or_zero(A.f_in, 4 * j + 2, A.nf_in), f3 = fr_load_or_zero(A.f_in, 4 * j + 3, A.nf_in);
      Fr g0 = fr_load_or_zero(A.g_in, 4 * j, A.ng_in), g1 = fr_load_or_zero(A.g_in, 4 * j + 1, A.ng_in);
      Fr g2 = fr_load_or_zero(A.g_in, 4 * j + 2, A.ng_in), g3 = fr_load_or_zero(A.g_in, 4 * j + 3, A.ng_in);
      fe = fr_add(f0, fr_mul(rho_tau, f1));
      fo = fr_add(f2, fr_mul(rho_tau, f3));
      ge = fr_add(g0, fr_mul(rho, g1));
      go = fr_add(g2, fr_mul(rho, g3));
      if (2 * j < nf) fp_store<FrParams>(A.f_out + (2 * j) * FR_BYTES, fe);
      if (2 * j + 1 < nf) fp_store<FrParams>(A.f_out + (2 * j + 1) * FR_BYTES, fo);
      if (2 * j < ng) fp_store<FrParams>(A.g_out + (2 * j) * FR_BYTES, ge);
      if (2 * j + 1 < ng) fp_store<FrParams>(A.g_out + (2 * j + 1) * FR_BYTES, go);
    } else {
      fe = fr_load_or_zero(A.f_in, 2 * j, nf);
      fo = fr_load_or_zero(A.f_in, 2 * j + 1, nf);
      ge = fr_load_or_zero(A.g_in, 2 * j, ng);
      go = fr_load_or_zero(A.g_in, 2 * j + 1, ng);
    }
    if (MSG) {
      Fr u = TW1 ? fe : fr_mul(fe, tw), w = TW1 ? fo : fr_mul(fo, tw);
      if (LAZY) {
        fp_mac_wide(wide[0], u, ge);
        fp_mac_wide(wide[1], u, go);
        fp_mac_wide(wide[2], w, ge);
      } else {
        acc[0] = fr_add(acc[0], fr_mul(u, ge));
        acc[1] = fr_add(acc[1], fr_mul(u, go));
        acc[2] = fr_add(acc[2], fr_mul(w, ge));
      }
      if (!TW1) tw = fr_mul(tw, step);
    }
  }
  if (MSG) {
    if (LAZY) {
#pragma unroll
      for (int k = 0; k < 3; k++) acc[k] = fr_add(acc[k], fp_redc_wide(wide[k]));
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
      // b = b1 + tau * b2
      Fr b = fr_add(acc[1], TW1 ? acc[2] : fr_mul(tau, acc[2]));
      fp_store<FrParams>(partials + ((size_t)blockIdx.x * 2) * FR_BYTES, acc[0]);
      fp_store<FrParams>(partials + ((size_t)blockIdx.x * 2 + 1) * FR_BYTES, b);
    }
  }
}
template <bool FOLD, bool MSG, bool LAZY = false, bool TW1 = false>
__global__ __launch_bounds__(256) void k_sc_round(ScArgs A, uint8_t* __restrict__ partials) {
  sc_round_body<FOLD, MSG, LAZY, TW1>(A, partials);
}
// The round of SEVERAL provers in one launch (Sumcheck::prove_batch maps its provers over rayon, proof.rs:85; the third sumcheck of psnark
// has 13 of them, each a launch of its own until round 5 -- ~240 launches per proof whose tails are launch latency): blockIdx.y = prover,
// its descriptor (the arguments of k_sc_round, where its partial sums go, how many blocks it wants) in device memory
struct ScMultiDesc {
  ScArgs A;
  uint8_t* partials;
  uint32_t blocks, tw1;  // tw1: this prover's message twist is one (block-uniform branch)
};
template <bool FOLD, bool MSG, bool LAZY = false>
__global__ __launch_bounds__(256) void k_sc_round_multi(const ScMultiDesc* __restrict__ descs) {
  const ScMultiDesc& D = descs[blockIdx.y];
  if (blockIdx.x >= D.blocks) return;
  if (MSG && D.tw1) sc_round_body<FOLD, MSG, LAZY, true>(D.A, D.partials);
  else sc_round_body<FOLD, MSG, LAZY, false>(D.A, D.partials);
}

// ------------------------------------------------------------------------------------------
// vector helpers
// ------------------------------------------------------------------------------------------
// out[i] = f[2i] + r * f[2i+1]                                                   misc.rs:52-56
__global__ __launch_bounds__(256) void k_fold(const uint8_t* __restrict__ f, size_t n, const uint32_t* __restrict__ r8,
                                              uint8_t* __restrict__ out) {
  const size_t m = (n + 1) / 2;
  Fr r = fp_load<FrParams>(r8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
    Fr e = fp_load<FrParams>(f + (2 * i) * FR_BYTES);
    Fr o = fr_load_or_zero(f, 2 * i + 1, n);
    fp_store<FrParams>(out + i * FR_BYTES, fr_add(e, fr_mul(r, o)));
  }
}

// out[i] = x^i: wave tiles of 64 * K elements, lane-strided inside the tile      misc.rs:59-65
__global__ __launch_bounds__(256) void k_powers(PowTable xt, size_t start, size_t n, uint8_t* __restrict__ out) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fr cur = pow_from_table(xt, start + t);
  Fr step = pow_from_table(xt, T);
  for (size_t i = t; i < n; i += T) {
    fp_store<FrParams>(out + i * FR_BYTES, cur);
    cur = fr_mul(cur, step);
  }
}

// half tables for tensor: tab[idx] = prod_{j<k} rho_j^{bit_j(idx)}
__global__ __launch_bounds__(256) void k_tensor_table(const uint32_t* __restrict__ rhos, uint32_t k, uint8_t* __restrict__ tab) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (1u << k)) return;
  Fr acc = Fr::one();
  for (uint32_t j = 0; j < k; j++)
    if ((idx >> j) & 1u) acc = fr_mul(acc, fp_load<FrParams>(rhos + 8 * j));
  fp_store<FrParams>(tab + (size_t)idx * FR_BYTES, acc);
}
// out[idx] = lo[idx & mask] * hi[idx >> klo]                                   misc.rs:133-149
// (start: the first index of a RANGE of the tensor -- a block of a block-sharded prover, gm_fr_tensor_range)
__global__ __launch_bounds__(256) void k_tensor(const uint8_t* __restrict__ lo, const uint8_t* __restrict__ hi,
                                                uint32_t klo, size_t start, size_t n, uint8_t* __restrict__ out) {
  const size_t mask = ((size_t)1 << klo) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t idx = start + i;
    Fr a = fp_load<FrParams>(lo + (idx & mask) * FR_BYTES);
    Fr b = fp_load<FrParams>(hi + (idx >> klo) * FR_BYTES);
    fp_store<FrParams>(out + i * FR_BYTES, fr_mul(a, b));
  }
}

// out[i] = lo[idx & mask] * hi[idx >> klo], idx = index[i]: `lookup(v, index)` (plookup/time_prover.rs:5-8) of a vector that is a FUNCTION of the
// index -- tensor(rho) (lo / hi: the half tables of k_tensor_table) or powers(x) (lo[j] = x^j, hi[j] = x^(j 2^klo)) -- without the vector: one
// multiplication per looked-up element from two L2-resident tables instead of an O(n) pass to build the vector and a random gather from it
__global__ __launch_bounds__(256) void k_gather_prod2(const uint8_t* __restrict__ lo, const uint8_t* __restrict__ hi, uint32_t klo,
                                                      const uint32_t* __restrict__ index, size_t n, uint8_t* __restrict__ out) {
  const size_t mask = ((size_t)1 << klo) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t idx = index[i];
    Fr a = fp_load<FrParams>(lo + (idx & mask) * FR_BYTES);
    Fr b = fp_load<FrParams>(hi + (idx >> klo) * FR_BYTES);
    fp_store<FrParams>(out + i * FR_BYTES, fr_mul(a, b));
  }
}

// out = a . b                                                                   misc.rs:205-208
__global__ __launch_bounds__(256) void k_hadamard(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
                                                  uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, fr_mul(fp_load<FrParams>(a + i * FR_BYTES), fp_load<FrParams>(b + i * FR_BYTES)));
}

// partial inner products, one per block                                         misc.rs:215-218
__global__ __launch_bounds__(256) void k_ip(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
                                            uint8_t* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * FR_BYTES];
  Fr acc[1] = {Fr::zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fr_add(acc[0], fr_mul(fp_load<FrParams>(a + i * FR_BYTES), fp_load<FrParams>(b + i * FR_BYTES)));
  block_sum<1>(acc, lds);
  if (threadIdx.x == 0) fp_store<FrParams>(partials + (size_t)blockIdx.x * FR_BYTES, acc[0]);
}

// evaluate_le at up to 3 points in one pass over the coefficients               misc.rs:194-199
// thread-strided: thread t owns coefficients t, t+T, ...; sum_i p_i x^i = sum_t x^t * Horner_{x^T}(p_t, p_{t+T}, ...)
struct EvalArgs {
  PowTable xt[3];
  uint32_t npoints;
  uint32_t log_threads;
  // neg_of[k] = j + 1: point k is MINUS point j.  A thread owns coefficients t, t + T, ... with T even, all of the parity of t,
  // and (-x)^T = x^T, so its share of p(-x) is (-1)^t times its share of p(x): no second Horner chain.  The tensor check
  // evaluates every polynomial at beta^2, beta, -beta and every folding at beta, -beta (tensorcheck/mod.rs:228-247).
  uint32_t neg_of[3];
};
// one block's share of p(x_k): thread t of T owns coefficients t, t + T, ...; partial sums of the block -> out[k]
__device__ __forceinline__ void eval_le_block(const uint8_t* __restrict__ p, size_t n, const EvalArgs& A, uint32_t log_threads, size_t block,
                                              uint8_t* __restrict__ out, uint8_t* lds) {
  const size_t T = (size_t)1 << log_threads;
  const size_t t = block * blockDim.x + threadIdx.x;
  Fr acc[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
  if (t < n) {
    Fr step[3];
    for (uint32_t k = 0; k < 3; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) step[k].l[i] = A.xt[k < A.npoints ? k : 0].p[log_threads][i];
    // highest index owned by this thread
    size_t cnt = (n - t + T - 1) / T;
    for (size_t c = cnt; c-- > 0;) {
      Fr coef = fp_load<FrParams>(p + (t + c * T) * FR_BYTES);
      for (uint32_t k = 0; k < A.npoints; k++)
        if (!A.neg_of[k]) acc[k] = fr_add(fr_mul(acc[k], step[k]), coef);
    }
    for (uint32_t k = 0; k < A.npoints; k++)
      if (!A.neg_of[k]) acc[k] = fr_mul(acc[k], pow_from_table(A.xt[k], t));
    for (uint32_t k = 0; k < A.npoints; k++)
      if (A.neg_of[k]) acc[k] = (t & 1) ? fp_neg<FrParams>(acc[A.neg_of[k] - 1]) : acc[A.neg_of[k] - 1];
  }
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; k++) fp_store<FrParams>(out + k * FR_BYTES, acc[k]);
}
__global__ __launch_bounds__(256) void k_eval_le(const uint8_t* __restrict__ p, size_t n, EvalArgs A,
                                                 uint8_t* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 3 * FR_BYTES];
  eval_le_block(p, n, A, A.log_threads, blockIdx.x, partials + (size_t)blockIdx.x * 3 * FR_BYTES, lds);
}
// several polynomials at the same points in ONE launch (the 23 folding levels of a tensor check: 2^23 ... 2 coefficients, most of
// them a lone block whose launch costs more than its work): job j owns blocks [blk_start, next job's blk_start)
struct EvalJob {
  const uint8_t* p;
  uint64_t n;
  uint64_t out_off;  // byte offset of the job's partial records
  uint32_t log_threads, blk_start;
};
__global__ __launch_bounds__(256) void k_eval_le_multi(const EvalJob* __restrict__ jobs, uint32_t njobs, EvalArgs A,
                                                       uint8_t* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 3 * FR_BYTES];
  uint32_t j = 0;
  while (j + 1 < njobs && jobs[j + 1].blk_start <= blockIdx.x) j++;
  const EvalJob J = jobs[j];
  const size_t lb = blockIdx.x - J.blk_start;
  eval_le_block(J.p, (size_t)J.n, A, J.log_threads, lb, partials + J.out_off + lb * 3 * FR_BYTES, lds);
}

// out[i] = sum_j c_j p_j[i], polynomials of different lengths (missing = 0)      misc.rs:37-48
// (32 terms: the merged opening polynomial of a tensor check is w + one term per folding level -- 25 at 2^24 constraints, 29 at 2^28 --
// and a second pass would read and write the 2^n-element sum once more)
constexpr size_t LINCOMB_MAX = 32;
struct LincombArgs {
  const uint8_t* p[LINCOMB_MAX];
  size_t len[LINCOMB_MAX];
  uint32_t c[LINCOMB_MAX][8];
  uint32_t k;
};
__global__ __launch_bounds__(256) void k_lincomb(LincombArgs A, size_t n, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = Fr::zero();
    for (uint32_t j = 0; j < A.k; j++) {
      if (i < A.len[j]) {
        Fr c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.l[q] = A.c[j][q];
        acc = fr_add(acc, fr_mul(c, fp_load<FrParams>(A.p[j] + i * FR_BYTES)));
      }
    }
    fp_store<FrParams>(out + i * FR_BYTES, acc);
  }
}
// accumulate variant for more than 24 polynomials: out[i] += ...
__global__ __launch_bounds__(256) void k_lincomb_acc(LincombArgs A, size_t n, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = fp_load<FrParams>(out + i * FR_BYTES);
    for (uint32_t j = 0; j < A.k; j++) {
      if (i < A.len[j]) {
        Fr c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.l[q] = A.c[j][q];
        acc = fr_add(acc, fr_mul(c, fp_load<FrParams>(A.p[j] + i * FR_BYTES)));
      }
    }
    fp_store<FrParams>(out + i * FR_BYTES, acc);
  }
}

// number of trailing (high-index) zero elements: each block reports the highest non-zero index + 1
// highest index + 1 of a non-zero element of v[lo, n) (0: none)
__global__ __launch_bounds__(256) void k_high_nonzero(const uint8_t* __restrict__ v, size_t lo, size_t n, unsigned long long* __restrict__ result) {
  unsigned long long best = 0;
  for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr x = fp_load<FrParams>(v + i * FR_BYTES);
    if (!x.is_zero()) best = i + 1;
  }
  if (best) atomicMax(result, best);
}

// out[i] = in[start + i * stride]
__global__ void k_fr_stride(const uint8_t* __restrict__ in, size_t start, size_t stride, size_t count, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, fp_load<FrParams>(in + (start + i * stride) * FR_BYTES));
}

// v[idx[j]] += val[j] for a handful of distinct positions (the carries / remainders at the seams of the laid-out opening)
__global__ void k_add_at(uint8_t* __restrict__ v, const unsigned long long* __restrict__ idx, const uint8_t* __restrict__ vals, unsigned k) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < k) fp_store<FrParams>(v + idx[j] * FR_BYTES, fr_add(fp_load<FrParams>(v + idx[j] * FR_BYTES), fp_load<FrParams>(vals + (size_t)j * FR_BYTES)));
}

// out[i] = in[n - 1 - i]: big-endian stream <-> little-endian coefficient vector (Reverse, src/iterable/slice.rs:17-39)
__global__ void k_reverse(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, fp_load<FrParams>(in + (n - 1 - i) * FR_BYTES));
}

__global__ void k_fill(uint8_t* __restrict__ v, size_t n, const uint32_t* __restrict__ val) {
  Fr x = fp_load<FrParams>(val);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(v + i * FR_BYTES, x);
}

// Division by a linear factor (x - alpha), high to low: q_i = f_{i+1} + alpha * q_{i+1}.
// Blocked three-phase evaluation of the first-order recurrence (an affine-map scan):
//   phase 1: per chunk c of K coefficients, t_c = sum_{j in chunk} f_j alpha^(j - lo_c)
//   phase 2: suffix combine S_c = t_c + alpha^K S_{c+1}   (sequential over the few chunk sums per
//            block, then across blocks on one wave -- the chunk count is n / K)
//   phase 3: rerun the recurrence inside each chunk seeded with S_{c+1}
// The same quotient as DensePolynomial::div by (x - alpha)                   src/kzg/time.rs:134-145
constexpr int DIV_K = 64;
__global__ __launch_bounds__(256) void k_div_phase1(const uint8_t* __restrict__ f, size_t n, const uint32_t* __restrict__ alpha8,
                                                    uint8_t* __restrict__ chunk_sums) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nch = (n + DIV_K - 1) / DIV_K;
  if (c >= nch) return;
  Fr alpha = fp_load<FrParams>(alpha8);
  const size_t lo = c * DIV_K, hi = min(lo + (size_t)DIV_K, n);
  Fr acc = Fr::zero();
  for (size_t j = hi; j-- > lo;) acc = fr_add(fr_mul(acc, alpha), fp_load<FrParams>(f + j * FR_BYTES));
  fp_store<FrParams>(chunk_sums + c * FR_BYTES, acc);
}
// in-place suffix combine over chunk sums with multiplier m = alpha^K, done by ONE thread per
// segment of `seg` chunks plus a sequential pass over segments (nch / seg is small)
__global__ __launch_bounds__(256) void k_div_phase2a(uint8_t* __restrict__ sums, size_t nch, size_t seg, const uint32_t* __restrict__ m8,
                                                     uint8_t* __restrict__ seg_sums) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nseg = (nch + seg - 1) / seg;
  if (s >= nseg) return;
  Fr m = fp_load<FrParams>(m8);
  const size_t lo = s * seg, hi = min(lo + seg, nch);
  Fr acc = Fr::zero();
  for (size_t c = hi; c-- > lo;) {
    acc = fr_add(fr_mul(acc, m), fp_load<FrParams>(sums + c * FR_BYTES));
    fp_store<FrParams>(sums + c * FR_BYTES, acc);  // local suffix (within the segment)
  }
  fp_store<FrParams>(seg_sums + s * FR_BYTES, acc);
}
// (the sequential combine over the few segment sums runs on the host, see fr_div_linear_factors)
// phase 3: q_i for i in chunk c.  S_{c+1} = local suffix of chunk c+1 within its segment
// + m^(chunks remaining in that segment) * carry[segment]
__global__ __launch_bounds__(256) void k_div_phase3(const uint8_t* __restrict__ f, size_t n, const uint32_t* __restrict__ alpha8,
                                                    const uint8_t* __restrict__ sums, const uint8_t* __restrict__ carry,
                                                    size_t seg, PowTable mt, uint8_t* __restrict__ q, uint8_t* __restrict__ rem) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nch = (n + DIV_K - 1) / DIV_K;
  if (c >= nch) return;
  Fr alpha = fp_load<FrParams>(alpha8);
  // value of the recurrence state entering chunk c from above: state = sum_{j >= hi_c} f_j alpha^(j - hi_c)
  Fr state = Fr::zero();
  if (c + 1 < nch) {
    const size_t c1 = c + 1, s1 = c1 / seg;
    const size_t seg_hi = min((s1 + 1) * seg, nch);
    state = fr_add(fp_load<FrParams>(sums + c1 * FR_BYTES),
                   fr_mul(pow_from_table(mt, seg_hi - c1), fp_load<FrParams>(carry + s1 * FR_BYTES)));
  }
  const size_t lo = c * DIV_K, hi = min(lo + (size_t)DIV_K, n);
  // q_{j-1} = f_j + alpha * q_j  with q_{n-1} := 0 ; state before processing f_j equals q_j
  for (size_t j = hi; j-- > lo;) {
    state = fr_add(fr_mul(state, alpha), fp_load<FrParams>(f + j * FR_BYTES));
    if (j >= 1) fp_store<FrParams>(q + (j - 1) * FR_BYTES, state);
    else fp_store<FrParams>(rem, state);  // f(alpha)
  }
}


// ---- entry-product / plookup vector builders (psnark) ------------------------------------------
// out[j] = src[index[j]]: `lookup` (plookup/time_prover.rs:5-8) and `sorted` (:67-74, with the
// extended-frequency index)
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ src, const uint32_t* __restrict__ index, size_t n,
                                                uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, fp_load<FrParams>(src + (size_t)index[i] * FR_BYTES));
}
// out[i] = v[i] + F::from(index[i]) * zeta (index == nullptr: the range 0..n)   plookup/time_prover.rs:11-21
// zeta2 = zeta * R^2 so that the Montgomery product with the plain integer is (index * zeta) * R
__global__ __launch_bounds__(256) void k_alg_hash(const uint8_t* __restrict__ v, const uint32_t* __restrict__ index, size_t n, uint64_t base,
                                                  const uint32_t* __restrict__ zeta2_8, uint8_t* __restrict__ out) {
  Fr z2 = fp_load<FrParams>(zeta2_8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t k = index ? (uint64_t)index[i] : base + (uint64_t)i;  // (base: the first index of a RANGE of the hashed vector)
    Fr kk = Fr::zero();
    kk.l[0] = (uint32_t)k;
    kk.l[1] = (uint32_t)(k >> 32);
    fp_store<FrParams>(out + i * FR_BYTES, fr_add(fp_load<FrParams>(v + i * FR_BYTES), fr_mul(kk, z2)));
  }
}
// plookup_set (plookup/time_prover.rs:23-35): out has n + 1 entries,
// out[i] = (1 + z) y + [i >= 1] v[i-1] + [i < n] z v[i]
__global__ __launch_bounds__(256) void k_plookup_set(const uint8_t* __restrict__ v, size_t n, const uint32_t* __restrict__ yz8,
                                                     uint8_t* __restrict__ out) {
  Fr y1z = fp_load<FrParams>(yz8), z = fp_load<FrParams>(yz8 + 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = y1z;
    if (i >= 1) acc = fr_add(acc, fp_load<FrParams>(v + (i - 1) * FR_BYTES));
    if (i < n) acc = fr_add(acc, fr_mul(z, fp_load<FrParams>(v + i * FR_BYTES)));
    fp_store<FrParams>(out + i * FR_BYTES, acc);
  }
}
// out[i] = v[i] + y: plookup_subset (:62-64)
__global__ __launch_bounds__(256) void k_add_scalar(const uint8_t* __restrict__ v, size_t n, const uint32_t* __restrict__ y8,
                                                    uint8_t* __restrict__ out) {
  Fr y = fp_load<FrParams>(y8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, fr_add(fp_load<FrParams>(v + i * FR_BYTES), y));
}
// right_rotation(monic(v)) = [1, v_0, ..., v_{n-1}]            entryproduct/time_prover.rs:14-23,47-51
__global__ __launch_bounds__(256) void k_shift_monic(const uint8_t* __restrict__ v, size_t n, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + i * FR_BYTES, i == 0 ? Fr::one() : fp_load<FrParams>(v + (i - 1) * FR_BYTES));
}

// ---- the same builders on ONE BLOCK of a block-sharded vector (gm_psnark_new_time_sharded) ----------------------------
// plookup_set restricted to the outputs [lo, lo + nout) of a vector of n_global + 1 of them: v = the elements [lo, lo + nv) of the
// hashed set that exist, prev = element lo - 1 (the last element of the block below: a 32-byte halo; null at lo = 0):
//   out[t] = (1 + z) y + (t >= 1 ? v[t - 1] : prev) + (t < nv ? z v[t] : 0)
__global__ __launch_bounds__(256) void k_plookup_set_block(const uint8_t* __restrict__ v, size_t nv, const uint32_t* __restrict__ prev8, size_t nout,
                                                           const uint32_t* __restrict__ yz8, uint8_t* __restrict__ out) {
  Fr y1z = fp_load<FrParams>(yz8), z = fp_load<FrParams>(yz8 + 8);
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nout; t += (size_t)gridDim.x * blockDim.x) {
    Fr acc = y1z;
    if (t >= 1) acc = fr_add(acc, fp_load<FrParams>(v + (t - 1) * FR_BYTES));
    else if (prev8) acc = fr_add(acc, fp_load<FrParams>(prev8));
    if (t < nv) acc = fr_add(acc, fr_mul(z, fp_load<FrParams>(v + t * FR_BYTES)));
    fp_store<FrParams>(out + t * FR_BYTES, acc);
  }
}
// right_rotation(monic(v)) on a block: out[0] = first (1 on the lowest block, else the last element of the block below),
// out[t] = v[t - 1]
__global__ __launch_bounds__(256) void k_shift_block(const uint8_t* __restrict__ v, const uint32_t* __restrict__ first8, size_t nout,
                                                     uint8_t* __restrict__ out) {
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nout; t += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + t * FR_BYTES, t == 0 ? fp_load<FrParams>(first8) : fp_load<FrParams>(v + (t - 1) * FR_BYTES));
}

// accumulated_product(monic(v)) (entryproduct/time_prover.rs:25-45): out[i] = prod_{j >= i} v[j], out[n] = 1.
// The same three-phase blocked scan as the division above with the monoid (Fr, *):
//   phase 1: p_c = product of chunk c (ACC_K elements per thread)
//   phase 2: suffix products of p over segments of `seg` chunks (one thread each, in place) + the
//            segment products; the few segment carries are combined on the host
//   phase 3: rerun each chunk from its incoming suffix
constexpr int ACC_K = 64;
__global__ __launch_bounds__(256) void k_accp_phase1(const uint8_t* __restrict__ v, size_t n, uint8_t* __restrict__ chunk_prod) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nch = (n + ACC_K - 1) / ACC_K;
  if (c >= nch) return;
  const size_t lo = c * ACC_K, hi = min(lo + (size_t)ACC_K, n);
  Fr acc = fp_load<FrParams>(v + (hi - 1) * FR_BYTES);
  for (size_t j = hi - 1; j-- > lo;) acc = fr_mul(acc, fp_load<FrParams>(v + j * FR_BYTES));
  fp_store<FrParams>(chunk_prod + c * FR_BYTES, acc);
}
__global__ __launch_bounds__(256) void k_accp_phase2a(uint8_t* __restrict__ prods, size_t nch, size_t seg, uint8_t* __restrict__ seg_prods) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nseg = (nch + seg - 1) / seg;
  if (s >= nseg) return;
  const size_t lo = s * seg, hi = min(lo + seg, nch);
  Fr acc = fp_load<FrParams>(prods + (hi - 1) * FR_BYTES);
  for (size_t c = hi - 1; c-- > lo;) {
    acc = fr_mul(acc, fp_load<FrParams>(prods + c * FR_BYTES));
    fp_store<FrParams>(prods + c * FR_BYTES, acc);  // suffix product within the segment
  }
  fp_store<FrParams>(seg_prods + s * FR_BYTES, acc);
}
__global__ __launch_bounds__(256) void k_accp_phase3(const uint8_t* __restrict__ v, size_t n, const uint8_t* __restrict__ prods,
                                                     const uint8_t* __restrict__ carry, size_t seg, const uint8_t* __restrict__ top_carry,
                                                     int write_monic, uint8_t* __restrict__ out) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nch = (n + ACC_K - 1) / ACC_K;
  if (c >= nch) return;
  // product of everything above chunk c
  Fr state;
  if (c + 1 < nch) {
    const size_t c1 = c + 1, s1 = c1 / seg;
    state = fr_mul(fp_load<FrParams>(prods + c1 * FR_BYTES), fp_load<FrParams>(carry + s1 * FR_BYTES));
  } else {
    // (a block of a sharded vector: the product of the blocks above comes in as carry[nseg]; the monic entry exists on the
    // rank that holds position n of the whole vector only)
    state = top_carry ? fp_load<FrParams>(top_carry) : Fr::one();
    if (write_monic) fp_store<FrParams>(out + n * FR_BYTES, state);  // the monic entry
  }
  const size_t lo = c * ACC_K, hi = min(lo + (size_t)ACC_K, n);
  for (size_t j = hi; j-- > lo;) {
    state = fr_mul(state, fp_load<FrParams>(v + j * FR_BYTES));
    fp_store<FrParams>(out + j * FR_BYTES, state);
  }
}

// y[i] = sum_k vals[k] * x[cols[k]] over row i of a CSR matrix       src/misc.rs:100-110
// (product_matrix_vector; the reference skips the multiplication when the coefficient is one,
// which cannot change the value)
__global__ __launch_bounds__(256) void k_spmv(const uint64_t* __restrict__ rowptr, const uint32_t* __restrict__ cols,
                                              const uint8_t* __restrict__ vals, size_t nrows,
                                              const uint8_t* __restrict__ x, size_t nx, uint8_t* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = Fr::zero();
    for (uint64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
      Fr xv = fr_load_or_zero(x, cols[k], nx);
      acc = fr_add(acc, fr_mul(fp_load<FrParams>(vals + k * FR_BYTES), xv));
    }
    fp_store<FrParams>(y + i * FR_BYTES, acc);
  }
}

// ------------------------------------------------------------------------------------------
// space prover (src/subprotocols/sumcheck/space_prover.rs): nothing but the ORIGINAL big-endian
// streams and the challenges is kept; every message is recomputed from the streams.
// After k folds the folded vector is F[m] = sum_{u < 2^k} f_le[m 2^k + u] w_u with
// w = tensor(challenges) (misc.rs:133-149) -- so a message is a pass over the streams that
// dots each group of 2^k originals with the tensor weights (two half tables lo/hi), no folded
// vector is ever materialised.  f_le[i] = stream[n - 1 - i] (Reverse, src/iterable/slice.rs:17-39).
// ------------------------------------------------------------------------------------------
struct SpArgs {
  const uint8_t* f;  // streams, big-endian
  const uint8_t* g;
  size_t nf, ng;
  const uint8_t* wf_lo;  // tensor(twisted challenges): lo table (2^klo) and hi table (2^(k-klo))
  const uint8_t* wf_hi;
  const uint8_t* wg_lo;  // tensor(challenges)
  const uint8_t* wg_hi;
  uint32_t k, klo;
  uint32_t le;  // inputs are little-endian partially folded vectors (sp_reduce), not the big-endian streams
  uint32_t tau[8];
  PowTable tau2;
  uint32_t log_threads;
  size_t npairs;
};

GM_DEV Fr sp_folded(const uint8_t* stream, size_t n, const uint8_t* lo, const uint8_t* hi, uint32_t k, uint32_t klo, size_t m,
                    bool le = false) {
  // F[m] over the little-endian view; elements past the end are zero
  const size_t span = (size_t)1 << k;
  const size_t first = m << k;
  Fr acc = Fr::zero();
  if (first >= n) return acc;
  const size_t last = min(first + span, n);
  const size_t lomask = ((size_t)1 << klo) - 1;
  for (size_t i = first; i < last; i++) {
    const size_t u = i - first;
    Fr v = fp_load<FrParams>(stream + (le ? i : n - 1 - i) * FR_BYTES);
    if (k > 0) {
      v = fr_mul(v, fp_load<FrParams>(lo + (u & lomask) * FR_BYTES));
      if (k > klo) v = fr_mul(v, fp_load<FrParams>(hi + (u >> klo) * FR_BYTES));
    }
    acc = fr_add(acc, v);
  }
  return acc;
}

__global__ __launch_bounds__(256) void k_sp_message(SpArgs A, uint8_t* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 3 * FR_BYTES];
  const size_t T = (size_t)1 << A.log_threads;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr tau, tw = Fr::one(), step;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    tau.l[i] = A.tau[i];
    step.l[i] = A.tau2.p[A.log_threads][i];
  }
  Fr acc[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
  if (t < A.npairs) tw = pow_from_table(A.tau2, t);
  for (size_t j = t; j < A.npairs; j += T) {
    Fr fe = sp_folded(A.f, A.nf, A.wf_lo, A.wf_hi, A.k, A.klo, 2 * j, A.le);
    Fr fo = sp_folded(A.f, A.nf, A.wf_lo, A.wf_hi, A.k, A.klo, 2 * j + 1, A.le);
    Fr ge = sp_folded(A.g, A.ng, A.wg_lo, A.wg_hi, A.k, A.klo, 2 * j, A.le);
    Fr go = sp_folded(A.g, A.ng, A.wg_lo, A.wg_hi, A.k, A.klo, 2 * j + 1, A.le);
    Fr u = fr_mul(fe, tw), w = fr_mul(fo, tw);
    acc[0] = fr_add(acc[0], fr_mul(u, ge));
    acc[1] = fr_add(acc[1], fr_mul(u, go));
    acc[2] = fr_add(acc[2], fr_mul(w, ge));
    tw = fr_mul(tw, step);
  }
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0) {
    Fr b = fr_add(acc[1], fr_mul(tau, acc[2]));
    fp_store<FrParams>(partials + ((size_t)blockIdx.x * 2) * FR_BYTES, acc[0]);
    fp_store<FrParams>(partials + ((size_t)blockIdx.x * 2 + 1) * FR_BYTES, b);
  }
}

// out[m] = F[m] for m < nout: materialises the folded vector (TimeProver::from(&SpaceProver),
// space_prover.rs:269-307) in little-endian order
__global__ __launch_bounds__(256) void k_sp_materialize(const uint8_t* __restrict__ stream, size_t n, const uint8_t* __restrict__ lo,
                                                        const uint8_t* __restrict__ hi, uint32_t k, uint32_t klo, size_t nout,
                                                        uint8_t* __restrict__ out, bool le) {
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < nout; m += (size_t)gridDim.x * blockDim.x)
    fp_store<FrParams>(out + m * FR_BYTES, sp_folded(stream, n, lo, hi, k, klo, m, le));
}

// ------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------
static void make_pow_table(const gmh::Fr& x, PowTable& t) {
  gmh::Fr cur = x;
  for (int b = 0; b < 40; b++) {
    memcpy(t.p[b], cur.l, 32);
    cur = cur.sqr();
  }
}
static unsigned grid_for(size_t n, unsigned max_blocks = 2048) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (unsigned)b;
}

// A round is an enqueue (the kernel and the asynchronous copy of its per-block partial sums) and a collect (after the
// stream has been waited for: the host adds the partials).  Driven one prover at a time the two run back to back
// (sc_launch); Sumcheck::prove_batch enqueues the round of EVERY prover before it waits once (sc_round_begin / _end).
// sc_prepare: the arguments of the launch (and the prover's buffers it needs); sc_after: the bookkeeping once it is enqueued
static bool sc_twist_one(const ScArgs& A) {
  static const bool off = getenv("GM_SC_TW1") && atoi(getenv("GM_SC_TW1")) == 0;  // A/B knob
  return !off && memcmp(A.tau, gmh::Fr::one().l, 32) == 0;
}
static int sc_prepare(Context* C, Sumcheck* S, bool fold, bool msg, const gmh::Fr& rho, ScArgs& A, unsigned* blocks_out, uint8_t** part_out_p) {
  memset(&A, 0, sizeof A);
  gmh::Fr tau = gmh::Fr::from_limbs(S->twist);
  gmh::Fr tau_msg = fold ? tau.sqr() : tau;
  gmh::Fr rho_tau = rho * tau;
  // herring's bilinear-module prover uses the twist only when folding; its message is the plain
  // a = <f_e, g_e>, b = <f_e, g_o> + <f_o, g_e> (src/herring/time_prover.rs:104-117)
  if (S->herring) tau_msg = gmh::Fr::one();
  if (fold && S->borrowed && S->cur == 1) {  // the second fold would write into the caller's vectors: the slot becomes ours
    S->borrowed = false;
    S->f[0] = S->g[0] = nullptr;
    S->fcap[0] = S->gcap[0] = 0;
    int rcb;
    if ((rcb = C->pool.alloc(((S->nf + 1) / 2) * FR_BYTES, (void**)&S->f[0], &S->fcap[0]))) return rcb;
    if ((rcb = C->pool.alloc(((S->ng + 1) / 2) * FR_BYTES, (void**)&S->g[0], &S->gcap[0]))) return rcb;
  }
  A.f_in = S->f[S->cur];
  A.g_in = S->g[S->cur];
  A.f_out = S->f[S->cur ^ 1];
  A.g_out = S->g[S->cur ^ 1];
  A.nf_in = S->nf;
  A.ng_in = S->ng;
  memcpy(A.rho_tau, rho_tau.l, 32);
  memcpy(A.rho, rho.l, 32);
  memcpy(A.tau, tau_msg.l, 32);
  gmh::Fr tau2 = tau_msg.sqr();
  make_pow_table(tau2, A.tau2);
  const size_t nf = fold ? (S->nf + 1) / 2 : S->nf, ng = fold ? (S->ng + 1) / 2 : S->ng;
  const size_t pf = (nf + 1) / 2, pg = (ng + 1) / 2;
  A.npairs = pf > pg ? pf : pg;
  // shard origin: tau_msg^(2 * first pair) where first pair = (element offset after fold) / 2
  uint64_t elem_off = fold ? S->pair_offset : 2 * S->pair_offset;  // pair_offset counts pairs of the CURRENT vectors
  {
    uint64_t first_pair = elem_off / 2;
    gmh::Fr o = gmh::Fr::one();
    gmh::Fr base = tau2;
    uint64_t e = first_pair;
    while (e) {
      if (e & 1) o = o * base;
      base = base.sqr();
      e >>= 1;
    }
    memcpy(A.origin, o.l, 32);
  }
  // threads: power of two, up to 2^17 (512 blocks of 256)
  uint32_t lt = 8;
  while (lt < 17 && ((size_t)1 << lt) < A.npairs) lt++;
  A.log_threads = lt;
  *blocks_out = (unsigned)(((size_t)1 << lt) / 256);
  const bool zc = (C->zero_copy & 1) != 0;  // the blocks write their partial sums into the pinned buffer themselves
  *part_out_p = zc ? reinterpret_cast<uint8_t*>(S->host_partials) : S->partials;
  return GM_OK;
}
static int sc_after(Context* C, Sumcheck* S, bool fold, bool msg, unsigned blocks) {
  const bool zc = (C->zero_copy & 1) != 0;
  S->pending_blocks = 0;
  if (msg) {
    if (!zc) GM_HIP(hipMemcpyAsync(S->host_partials, S->partials, (size_t)blocks * 2 * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
    S->pending_blocks = blocks;
  }
  if (fold) {
    gmh::Fr t2 = gmh::Fr::from_limbs(S->twist).sqr();  // the real twist, also for herring
    S->cur ^= 1;
    S->nf = (S->nf + 1) / 2;
    S->ng = (S->ng + 1) / 2;
    t2.to_limbs(S->twist);
    S->pair_offset = S->pair_offset / 2;
  }
  return GM_OK;
}
static bool sc_lazy() {
  // lazy reduction of the message's inner products (GM_SC_LAZY=0: the reduced form)
  static const bool lazy = !(getenv("GM_SC_LAZY") && atoi(getenv("GM_SC_LAZY")) == 0);
  return lazy;
}
static int sc_enqueue(Context* C, Sumcheck* S, bool fold, bool msg, const gmh::Fr& rho) {
  ScArgs A;
  unsigned blocks = 0;
  uint8_t* part_out = nullptr;
  int rc = sc_prepare(C, S, fold, msg, rho, A, &blocks, &part_out);
  if (rc) return rc;
  hipStream_t st = C->stream;
  // stage timer of bench.py's sumcheck roofline: only when profiling is on, and then (one event pair) only
  // with provers driven one at a time
  Profiler& prof = C->prof;
  prof.begin(PROF_SC_ROUND, st);
  const bool lazy = sc_lazy();
  if (msg && lazy && sc_twist_one(A)) {
    if (fold) hipLaunchKernelGGL((k_sc_round<true, true, true, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
    else hipLaunchKernelGGL((k_sc_round<false, true, true, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
  } else if (fold && msg && lazy)
    hipLaunchKernelGGL((k_sc_round<true, true, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
  else if (fold && msg)
    hipLaunchKernelGGL((k_sc_round<true, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
  else if (fold)
    hipLaunchKernelGGL((k_sc_round<true, false>), dim3(blocks), dim3(256), 0, st, A, part_out);
  else if (lazy)
    hipLaunchKernelGGL((k_sc_round<false, true, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
  else
    hipLaunchKernelGGL((k_sc_round<false, true>), dim3(blocks), dim3(256), 0, st, A, part_out);
  prof.end(PROF_SC_ROUND, st);
  GM_HIP(hipGetLastError());
  if (prof.on && !msg) GM_HIP(hipStreamSynchronize(st));
  return sc_after(C, S, fold, msg, blocks);
}
// the same for SEVERAL provers with the same (fold, msg): their descriptors go to the device in one small copy, ONE launch
static int sc_enqueue_many(Context* C, Sumcheck** S, size_t k, bool fold, bool msg, const gmh::Fr& rho, size_t scratch_slot) {
  if (k == 1) return sc_enqueue(C, S[0], fold, msg, rho);
  GM_FR_LOCK(C);  // (the descriptors travel through the small-upload area of the vector scratch)
  std::vector<ScMultiDesc> d(k);
  unsigned max_blocks = 0;
  for (size_t j = 0; j < k; j++) {
    memset(&d[j], 0, sizeof d[j]);
    int rc = sc_prepare(C, S[j], fold, msg, rho, d[j].A, &d[j].blocks, &d[j].partials);
    if (rc) return rc;
    d[j].tw1 = msg && sc_twist_one(d[j].A) ? 1u : 0u;
    max_blocks = std::max(max_blocks, d[j].blocks);
  }
  const size_t bytes = k * sizeof(ScMultiDesc), slot = (size_t)32 << 10;
  GM_CHECK(bytes <= slot && C->sc_desc_host, GM_EINVAL, "sumcheck: %zu provers in one launch", k);
  (void)scratch_slot;
  int rc = C->fr_scratch.ensure(1 << 20);
  if (rc) return rc;
  // through PINNED memory of the context, one ring slot per launch: the copy is asynchronous and nothing waits before this call returns.  (The
  // first version used one slot per (fold, message) kind: the two launches of a round in which some provers of a sharded batch leave their
  // sharded phase shared it, and the first launch read the second one's descriptors -- ranks produced different proofs at 8 ranks.)
  const unsigned ring = C->sc_desc_next++ & 7u;
  uint8_t* dd = C->fr_scratch.as<uint8_t>() + ring * slot;
  uint8_t* hs = C->sc_desc_host + ring * slot;
  memcpy(hs, d.data(), bytes);
  GM_HIP(hipMemcpyAsync(dd, hs, bytes, hipMemcpyHostToDevice, C->stream));
  hipStream_t st = C->stream;
  const dim3 grid(max_blocks, (unsigned)k);
  const ScMultiDesc* dp = reinterpret_cast<const ScMultiDesc*>(dd);
  const bool lazy = sc_lazy();
  if (fold && msg && lazy)
    hipLaunchKernelGGL((k_sc_round_multi<true, true, true>), grid, dim3(256), 0, st, dp);
  else if (fold && msg)
    hipLaunchKernelGGL((k_sc_round_multi<true, true>), grid, dim3(256), 0, st, dp);
  else if (fold)
    hipLaunchKernelGGL((k_sc_round_multi<true, false>), grid, dim3(256), 0, st, dp);
  else if (lazy)
    hipLaunchKernelGGL((k_sc_round_multi<false, true, true>), grid, dim3(256), 0, st, dp);
  else
    hipLaunchKernelGGL((k_sc_round_multi<false, true>), grid, dim3(256), 0, st, dp);
  GM_HIP(hipGetLastError());
  for (size_t j = 0; j < k; j++) {
    rc = sc_after(C, S[j], fold, msg, d[j].blocks);
    if (rc) return rc;
  }
  // a fold-only launch has no collect phase that would wait for the stream: wait here, so that its descriptor slot is free for the next launch
  if (!msg) GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

static size_t ceil_log2_sz(size_t n) {
  size_t b = 0, v = n > 1 ? n - 1 : 0;
  while (v) {
    b++;
    v >>= 1;
  }
  return b;
}

int PartialBufs::take(uint8_t** dev, uint64_t** host) {
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!free_pairs.empty()) {
      *dev = free_pairs.back().first;
      *host = free_pairs.back().second;
      free_pairs.pop_back();
      return GM_OK;
    }
  }
  GM_HIP(dev_malloc((void**)dev, PARTIAL_BUF_BYTES));
  GM_HIP(hipHostMalloc((void**)host, PARTIAL_BUF_BYTES, hipHostMallocDefault));
  return GM_OK;
}
void PartialBufs::give(uint8_t* dev, uint64_t* host) {
  if (!dev && !host) return;
  if (dev && host) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_pairs.size() < 64) {
      free_pairs.emplace_back(dev, host);
      return;
    }
  }
  if (dev) (void)gm::raw_free(dev);
  if (host) (void)hipHostFree(host);
}
void PartialBufs::release_all() {
  std::lock_guard<std::mutex> lk(mu);
  for (auto& p : free_pairs) {
    (void)gm::raw_free(p.first);
    (void)hipHostFree(p.second);
  }
  free_pairs.clear();
}

// borrow = true: no copy -- the prover reads the caller's device vectors in place until its first fold (the native provers hand it
// temporaries or vectors that outlive the sumcheck): 2 x n x 32 bytes less to move and to hold per prover
int sc_create(Context* C, const void* f_src, size_t nf, const void* g_src, size_t ng, bool src_is_device,
              const uint64_t twist[4], uint64_t* handle, bool borrow) {
  GM_CHECK(nf >= 1 && ng >= 1, GM_EINVAL, "sumcheck: empty vectors");
  auto S = std::make_unique<Sumcheck>();
  S->nf = nf;
  S->ng = ng;
  memcpy(S->twist, twist, 32);
  S->tot_rounds = ceil_log2_sz(nf > ng ? nf : ng);  // time_prover.rs:35-38
  int rc;
  if (borrow && src_is_device) {
    S->borrowed = true;
    S->f[0] = const_cast<uint8_t*>(static_cast<const uint8_t*>(f_src));
    S->g[0] = const_cast<uint8_t*>(static_cast<const uint8_t*>(g_src));
    if ((rc = C->pool.alloc(((nf + 1) / 2) * FR_BYTES, (void**)&S->f[1], &S->fcap[1]))) return rc;
    if ((rc = C->pool.alloc(((ng + 1) / 2) * FR_BYTES, (void**)&S->g[1], &S->gcap[1]))) return rc;
    if ((rc = C->partial_bufs.take(&S->partials, &S->host_partials))) return rc;
    *handle = put_prover(std::move(S));
    return GM_OK;
  }
  if ((rc = C->pool.alloc(nf * FR_BYTES, (void**)&S->f[0], &S->fcap[0]))) return rc;
  if ((rc = C->pool.alloc(((nf + 1) / 2) * FR_BYTES, (void**)&S->f[1], &S->fcap[1]))) return rc;
  if ((rc = C->pool.alloc(ng * FR_BYTES, (void**)&S->g[0], &S->gcap[0]))) return rc;
  if ((rc = C->pool.alloc(((ng + 1) / 2) * FR_BYTES, (void**)&S->g[1], &S->gcap[1]))) return rc;
  if ((rc = C->partial_bufs.take(&S->partials, &S->host_partials))) return rc;
  hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  GM_HIP(hipMemcpyAsync(S->f[0], f_src, nf * FR_BYTES, kind, C->stream));
  GM_HIP(hipMemcpyAsync(S->g[0], g_src, ng * FR_BYTES, kind, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  *handle = put_prover(std::move(S));
  return GM_OK;
}

void sc_destroy(Sumcheck* S) {
  Context* C = context();
  for (int i = 0; i < 2; i++) {
    if (C && !(i == 0 && S->borrowed)) {
      C->pool.free(S->f[i], S->fcap[i]);
      C->pool.free(S->g[i], S->gcap[i]);
    }
  }
  if (C) C->partial_bufs.give(S->partials, S->host_partials);
  else {
    if (S->partials) (void)gm::raw_free(S->partials);
    if (S->host_partials) (void)hipHostFree(S->host_partials);
  }
  S->partials = nullptr;
  S->host_partials = nullptr;
}

// the stream the round was enqueued on has been waited for
// ---- the tail of a sumcheck on the host -------------------------------------------------------------------------------
// Same arithmetic as k_sc_round (fold f' = f_e + rho tau f_o, g' = g_e + rho g_o, then the message on the folded vectors with the
// twist of the round, time_prover.rs:75-123), on <= SC_HOST_TAIL elements: the last ~8 rounds of EVERY sumcheck -- time provers,
// elastic ones after their switch, the provers of a batch, the replicated tail of a sharded one.
static bool sc_host_ready(Context* C, Sumcheck* S, int* rc) {
  *rc = GM_OK;
  if (S->on_host) return true;
  static const size_t tail = getenv("GM_SC_HOST_TAIL") ? (size_t)strtoull(getenv("GM_SC_HOST_TAIL"), nullptr, 10) : SC_HOST_TAIL;  // A/B knob, 0 = off
  const size_t n = S->nf > S->ng ? S->nf : S->ng;
  if (tail == 0 || n > tail || S->pending_blocks != 0) return false;  // (also while the stage timers are on: the profiled run makes the launches of the timed one)
  S->hf.assign(4 * S->nf, 0);
  S->hg.assign(4 * S->ng, 0);
  hipError_t e = hipSuccess;
  if (S->nf) e = hipMemcpyAsync(S->hf.data(), S->f[S->cur], S->nf * FR_BYTES, hipMemcpyDeviceToHost, C->stream);
  if (e == hipSuccess && S->ng) e = hipMemcpyAsync(S->hg.data(), S->g[S->cur], S->ng * FR_BYTES, hipMemcpyDeviceToHost, C->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
  if (e != hipSuccess) {
    *rc = hip_fail(e, "sumcheck tail: copy to the host", __FILE__, __LINE__);
    return false;
  }
  S->on_host = true;
  return true;
}
static void sc_host_step(Sumcheck* S, bool fold, bool msg, const gmh::Fr& rho) {
  using gmh::Fr;
  const Fr tau = Fr::from_limbs(S->twist);
  Fr tau_msg = fold ? tau.sqr() : tau;
  if (S->herring) tau_msg = Fr::one();
  const uint64_t elem_off = fold ? S->pair_offset : 2 * S->pair_offset;
  if (fold) {
    const Fr rho_tau = rho * tau;
    auto fold_vec = [](std::vector<uint64_t>& v, size_t n, const Fr& c) {
      const size_t m = (n + 1) / 2;
      for (size_t i = 0; i < m; i++) {
        Fr x = Fr::from_limbs(v.data() + 8 * i);
        if (2 * i + 1 < n) x = x + c * Fr::from_limbs(v.data() + 8 * i + 4);
        x.to_limbs(v.data() + 4 * i);  // slot i <= slot 2 i: in place
      }
      v.resize(4 * m);
    };
    fold_vec(S->hf, S->nf, rho_tau);
    fold_vec(S->hg, S->ng, rho);
    S->nf = (S->nf + 1) / 2;
    S->ng = (S->ng + 1) / 2;
    tau.sqr().to_limbs(S->twist);
    S->pair_offset = S->pair_offset / 2;
  }
  if (msg) {
    const Fr tau2 = tau_msg.sqr();
    Fr runner = Fr::one(), base = tau2;
    for (uint64_t e = elem_off / 2; e; e >>= 1) {
      if (e & 1) runner = runner * base;
      base = base.sqr();
    }
    const size_t pf = (S->nf + 1) / 2, pg = (S->ng + 1) / 2, np = pf < pg ? pf : pg;  // a pair with a side missing contributes nothing
    Fr a = Fr::zero(), b = Fr::zero();
    for (size_t i = 0; i < np; i++) {
      const Fr fe = Fr::from_limbs(S->hf.data() + 8 * i), ge = Fr::from_limbs(S->hg.data() + 8 * i);
      const Fr fo = 2 * i + 1 < S->nf ? Fr::from_limbs(S->hf.data() + 8 * i + 4) : Fr::zero();
      const Fr go = 2 * i + 1 < S->ng ? Fr::from_limbs(S->hg.data() + 8 * i + 4) : Fr::zero();
      a = a + fe * ge * runner;
      b = b + (fe * go + ge * fo * tau_msg) * runner;
      runner = runner * tau2;
    }
    a.to_limbs(S->host_msg);
    b.to_limbs(S->host_msg + 4);
    S->host_msg_pending = true;
  }
}

static void sc_collect(Context* C, Sumcheck* S, uint64_t a_out[4], uint64_t b_out[4]) {
  gmh::Fr a = gmh::Fr::zero(), b = gmh::Fr::zero();
  for (unsigned i = 0; i < S->pending_blocks; i++) {
    a = a + gmh::Fr::from_limbs(S->host_partials + (size_t)i * 8);
    b = b + gmh::Fr::from_limbs(S->host_partials + (size_t)i * 8 + 4);
  }
  a.to_limbs(a_out);
  b.to_limbs(b_out);
  S->pending_blocks = 0;
  C->prof.collect();
}

static int sc_launch(Context* C, Sumcheck* S, bool fold, bool msg, const gmh::Fr& rho, uint64_t a_out[4], uint64_t b_out[4]) {
  int rc;
  if (sc_host_ready(C, S, &rc)) {
    sc_host_step(S, fold, msg, rho);
    if (msg) {
      memcpy(a_out, S->host_msg, 32);
      memcpy(b_out, S->host_msg + 4, 32);
      S->host_msg_pending = false;
    }
    return GM_OK;
  }
  if (rc) return rc;
  rc = sc_enqueue(C, S, fold, msg, rho);
  if (rc) return rc;
  if (msg) {
    GM_HIP(hipStreamSynchronize(C->stream));
    sc_collect(C, S, a_out, b_out);
  } else {
    C->prof.collect();
  }
  return GM_OK;
}

// Prover::next_message                                                      time_prover.rs:83-123
int sc_round(Context* C, Sumcheck* S, const uint64_t* challenge, uint64_t a[4], uint64_t b[4], int* has_msg) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_CHECK(S->round <= S->tot_rounds, GM_ESTATE, "More rounds than needed.");
  const bool fold = challenge != nullptr;
  const bool msg = S->round != S->tot_rounds;
  gmh::Fr rho = fold ? gmh::Fr::from_limbs(challenge) : gmh::Fr::zero();
  if (fold || msg) {
    int rc = sc_launch(C, S, fold, msg, rho, a, b);
    if (rc) return rc;
  }
  if (!msg) {
    *has_msg = 0;
    return GM_OK;
  }
  S->round += 1;
  *has_msg = 1;
  return GM_OK;
}

// split-phase round (see sc_enqueue): _begin launches, _end waits for the stream and returns the message
int sc_round_begin(Context* C, Sumcheck* S, const uint64_t* challenge, int* has_msg) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_CHECK(S->round <= S->tot_rounds, GM_ESTATE, "More rounds than needed.");
  GM_CHECK(S->pending_blocks == 0 && !S->host_msg_pending, GM_ESTATE, "sc_round_begin: the previous round has not been collected");
  const bool fold = challenge != nullptr;
  const bool msg = S->round != S->tot_rounds;
  gmh::Fr rho = fold ? gmh::Fr::from_limbs(challenge) : gmh::Fr::zero();
  if (fold || msg) {
    int rc;
    if (sc_host_ready(C, S, &rc)) sc_host_step(S, fold, msg, rho);
    else {
      if (rc) return rc;
      rc = sc_enqueue(C, S, fold, msg, rho);
      if (rc) return rc;
    }
  }
  if (!msg) {
    *has_msg = 0;
    return GM_OK;
  }
  S->round += 1;
  *has_msg = 1;
  return GM_OK;
}
int sc_round_end(Context* C, Sumcheck* S, uint64_t a[4], uint64_t b[4]) {
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->host_msg_pending) {
    memcpy(a, S->host_msg, 32);
    memcpy(b, S->host_msg + 4, 32);
    S->host_msg_pending = false;
    return GM_OK;
  }
  GM_CHECK(S->pending_blocks != 0, GM_ESTATE, "sc_round_end: no round in flight");
  GM_HIP(hipStreamSynchronize(C->stream));
  sc_collect(C, S, a, b);
  return GM_OK;
}

// the split-phase round of k provers with the same challenge (Sumcheck::prove_batch): provers on the device with the same (fold,
// message) share ONE launch; the tails on the host step there.  Every prover is collected with sc_round_end as before.
int sc_round_begin_many(Context* C, Sumcheck** S, size_t k, const uint64_t* challenge, int* has_msg) {
  const bool fold = challenge != nullptr;
  const gmh::Fr rho = fold ? gmh::Fr::from_limbs(challenge) : gmh::Fr::zero();
  std::vector<std::unique_lock<std::mutex>> locks;
  for (size_t j = 0; j < k; j++) {
    for (size_t i = 0; i < j; i++) GM_CHECK(S[i] != S[j], GM_EINVAL, "sc_round_begin_many: a prover appears twice");
    locks.emplace_back(S[j]->mu);
  }
  std::vector<Sumcheck*> dev[2];  // [msg]
  for (size_t j = 0; j < k; j++) {
    Sumcheck* P = S[j];
    GM_CHECK(P->round <= P->tot_rounds, GM_ESTATE, "More rounds than needed.");
    GM_CHECK(P->pending_blocks == 0 && !P->host_msg_pending, GM_ESTATE, "sc_round_begin_many: the previous round has not been collected");
    const bool msg = P->round != P->tot_rounds;
    has_msg[j] = msg ? 1 : 0;
    if (!(fold || msg)) continue;
    int rc;
    if (sc_host_ready(C, P, &rc)) sc_host_step(P, fold, msg, rho);
    else {
      if (rc) return rc;
      dev[msg ? 1 : 0].push_back(P);
    }
  }
  for (int m = 1; m >= 0; m--)
    if (!dev[m].empty()) {
      int rc = sc_enqueue_many(C, dev[m].data(), dev[m].size(), fold, m == 1, rho, (size_t)m);
      if (rc) return rc;
    }
  for (size_t j = 0; j < k; j++)
    if (has_msg[j]) S[j]->round += 1;
  return GM_OK;
}

int sc_set_herring(Sumcheck* S, int on) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_CHECK(S->round == 0, GM_ESTATE, "sc_set_herring: must be called before the first round");
  S->herring = on != 0;
  // herring::Witness::required_rounds uses the SHORTER vector (src/herring/time_prover.rs:36-39)
  S->tot_rounds = ceil_log2_sz(on ? (S->nf < S->ng ? S->nf : S->ng) : (S->nf > S->ng ? S->nf : S->ng));
  return GM_OK;
}

int sc_fold(Context* C, Sumcheck* S, const uint64_t challenge[4]) {
  std::lock_guard<std::mutex> lk(S->mu);
  uint64_t a[4], b[4];
  return sc_launch(C, S, true, false, gmh::Fr::from_limbs(challenge), a, b);
}

int sc_final(Context* C, Sumcheck* S, uint64_t f0[4], uint64_t g0[4], int* has) {
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->round != S->tot_rounds) {
    *has = 0;
    return GM_OK;
  }
  if (S->on_host) {
    memset(f0, 0, 32);
    memset(g0, 0, 32);
    if (S->nf) memcpy(f0, S->hf.data(), 32);
    if (S->ng) memcpy(g0, S->hg.data(), 32);
    *has = 1;
    return GM_OK;
  }
  GM_HIP(hipMemcpyAsync(f0, S->f[S->cur], FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipMemcpyAsync(g0, S->g[S->cur], FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  *has = 1;
  return GM_OK;
}

// ---- space prover -------------------------------------------------------------------------------
// The tensor weights factor level by level: with SP_LV challenges per level,
//   F_k[m] = sum_t W_hi(t) * ( sum_v x[(m 2^(k-SP_LV) + t) 2^SP_LV + v] W_lo(v) ),
// so a round first folds the streams by groups of SP_LV challenges into short little-endian vectors
// (one thread per output, 2^SP_LV terms each -- every level has >= n / 2^SP_LV outputs to spread over
// the GPU) and hands the last <= SP_LV challenges to the message / materialise kernel.  Nothing is kept
// between rounds: the space prover's state stays the streams and the challenges.
constexpr uint32_t SP_LV = 8;
struct SpReduced {
  const uint8_t *f, *g;  // inputs of the final kernel
  size_t nf, ng;
  bool le;
  uint32_t krem;          // challenges left for the final kernel (tables wf_lo / wg_lo of 2^krem entries)
};
static size_t ceil_shift(size_t n, uint32_t k);
static int sp_reduce(Context* C, SpaceProver* S, SpReduced* R) {
  const uint32_t k = (uint32_t)S->challenges.size() / 4;
  const uint32_t levels = k > SP_LV ? (k - 1) / SP_LV : 0;
  const uint32_t krem = k - levels * SP_LV;
  R->f = S->f;
  R->g = S->g;
  R->nf = S->nf;
  R->ng = S->ng;
  R->le = false;
  R->krem = krem;
  if (k == 0) return GM_OK;
  const size_t ntab = (size_t)1 << SP_LV;
  // layout: [k twisted challenges][k plain challenges][per level: twisted table, plain table][final: twisted, plain]
  //         [f ping][f pong][g ping][g pong]
  const size_t f0 = ceil_shift(S->nf, SP_LV), f1 = ceil_shift(S->nf, 2 * SP_LV), g0 = ceil_shift(S->ng, SP_LV), g1 = ceil_shift(S->ng, 2 * SP_LV);
  const size_t tab_bytes = (size_t)2 * k * 32 + ((size_t)levels + 1) * 2 * ntab * FR_BYTES;
  const size_t tables_need = tab_bytes + (f0 + f1 + g0 + g1 + 4) * FR_BYTES + 256;
  if (S->tables_cap < tables_need) {
    if (S->tables) C->pool.free(S->tables, S->tables_cap);
    S->tables = nullptr;
    S->tables_cap = 0;
    int rc = C->pool.alloc(tables_need, (void**)&S->tables, &S->tables_cap);
    if (rc) return rc;
  }
  uint8_t* base = S->tables;
  uint8_t* ch_t = base;
  uint8_t* ch_p = base + (size_t)k * 32;
  uint8_t* tabs = base + (size_t)2 * k * 32;
  uint8_t* bufs = base + tab_bytes;
  uint8_t* fbuf[2] = {bufs, bufs + f0 * FR_BYTES};
  uint8_t* gbuf[2] = {bufs + (f0 + f1) * FR_BYTES, bufs + (f0 + f1 + g0) * FR_BYTES};
  GM_HIP(hipMemcpyAsync(ch_t, S->twisted.data(), (size_t)k * 32, hipMemcpyHostToDevice, C->stream));
  GM_HIP(hipMemcpyAsync(ch_p, S->challenges.data(), (size_t)k * 32, hipMemcpyHostToDevice, C->stream));
  for (uint32_t l = 0; l <= levels; l++) {
    const uint32_t cnt = l < levels ? SP_LV : krem;
    uint8_t* tt = tabs + (size_t)l * 2 * ntab * FR_BYTES;
    hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << cnt)), dim3(256), 0, C->stream,
                       (const uint32_t*)(ch_t + (size_t)l * SP_LV * 32), cnt, tt);
    hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << cnt)), dim3(256), 0, C->stream,
                       (const uint32_t*)(ch_p + (size_t)l * SP_LV * 32), cnt, tt + ntab * FR_BYTES);
  }
  for (uint32_t l = 0; l < levels; l++) {
    const uint8_t* tt = tabs + (size_t)l * 2 * ntab * FR_BYTES;
    const size_t nfo = ceil_shift(R->nf, SP_LV), ngo = ceil_shift(R->ng, SP_LV);
    hipLaunchKernelGGL(k_sp_materialize, dim3(grid_for(nfo, 1u << 20)), dim3(256), 0, C->stream, R->f, R->nf, tt, (const uint8_t*)nullptr, SP_LV,
                       SP_LV, nfo, fbuf[l & 1], R->le);
    hipLaunchKernelGGL(k_sp_materialize, dim3(grid_for(ngo, 1u << 20)), dim3(256), 0, C->stream, R->g, R->ng, tt + ntab * FR_BYTES,
                       (const uint8_t*)nullptr, SP_LV, SP_LV, ngo, gbuf[l & 1], R->le);
    R->f = fbuf[l & 1];
    R->g = gbuf[l & 1];
    R->nf = nfo;
    R->ng = ngo;
    R->le = true;
  }
  const uint8_t* tf = tabs + (size_t)levels * 2 * ntab * FR_BYTES;
  S->wf_lo = const_cast<uint8_t*>(tf);
  S->wg_lo = const_cast<uint8_t*>(tf + ntab * FR_BYTES);
  S->wf_hi = S->wg_hi = nullptr;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

static size_t ceil_shift(size_t n, uint32_t k) { return k >= 63 ? (n ? 1 : 0) : (n + (((size_t)1 << k) - 1)) >> k; }

int sp_create(Context* C, const void* f_stream, size_t nf, const void* g_stream, size_t ng, bool src_is_device,
              const uint64_t twist[4], uint64_t* handle, bool borrow) {
  GM_CHECK(nf >= 1 && ng >= 1, GM_EINVAL, "space prover: empty streams");
  auto S = std::make_unique<SpaceProver>();
  S->nf = nf;
  S->ng = ng;
  memcpy(S->twist, twist, 32);
  S->tot_rounds = ceil_log2_sz(nf < ng ? nf : ng);  // space_prover.rs:74-77: log2(min(len))
  int rc;
  if (borrow && src_is_device) {  // the streams are read in place for the life of the prover (it never writes them)
    S->borrowed = true;
    S->f = const_cast<uint8_t*>(static_cast<const uint8_t*>(f_stream));
    S->g = const_cast<uint8_t*>(static_cast<const uint8_t*>(g_stream));
    if ((rc = C->partial_bufs.take(&S->partials, &S->host_partials))) return rc;
    std::lock_guard<std::mutex> lk(C->mu);
    *handle = C->next_handle++;
    C->space_provers[*handle] = std::move(S);
    return GM_OK;
  }
  if ((rc = C->pool.alloc(nf * FR_BYTES, (void**)&S->f, &S->fcap))) return rc;
  if ((rc = C->pool.alloc(ng * FR_BYTES, (void**)&S->g, &S->gcap))) return rc;
  if ((rc = C->partial_bufs.take(&S->partials, &S->host_partials))) return rc;
  hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  GM_HIP(hipMemcpyAsync(S->f, f_stream, nf * FR_BYTES, kind, C->stream));
  GM_HIP(hipMemcpyAsync(S->g, g_stream, ng * FR_BYTES, kind, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  std::lock_guard<std::mutex> lk(C->mu);
  *handle = C->next_handle++;
  C->space_provers[*handle] = std::move(S);
  return GM_OK;
}

void sp_destroy(Context* C, SpaceProver* S) {
  if (C && !S->borrowed) {
    C->pool.free(S->f, S->fcap);
    C->pool.free(S->g, S->gcap);
  }
  if (S->tables) {
    if (C) C->pool.free(S->tables, S->tables_cap);
    else (void)gm::raw_free(S->tables);
  }
  S->tables = nullptr;
  S->tables_cap = 0;
  if (C) C->partial_bufs.give(S->partials, S->host_partials);
  else {
    if (S->partials) (void)gm::raw_free(S->partials);
    if (S->host_partials) (void)hipHostFree(S->host_partials);
  }
  S->partials = nullptr;
  S->host_partials = nullptr;
}

// Prover::fold for the space prover: store the randomness aside        space_prover.rs:245-249
static void sp_push(SpaceProver* S, const uint64_t r[4]) {
  gmh::Fr rr = gmh::Fr::from_limbs(r), tw = gmh::Fr::from_limbs(S->twist);
  gmh::Fr rt = rr * tw;
  S->challenges.insert(S->challenges.end(), r, r + 4);
  S->twisted.insert(S->twisted.end(), rt.l, rt.l + 4);
  tw.sqr().to_limbs(S->twist);
}

int sp_fold(Context*, SpaceProver* S, const uint64_t challenge[4]) {
  std::lock_guard<std::mutex> lk(S->mu);
  sp_push(S, challenge);
  return GM_OK;
}

// SpaceProver::next_message                                             space_prover.rs:117-240
int sp_round(Context* C, SpaceProver* S, const uint64_t* challenge, uint64_t a_out[4], uint64_t b_out[4], int* has_msg) {
  std::lock_guard<std::mutex> lk(S->mu);
  GM_CHECK(S->round <= S->tot_rounds, GM_ESTATE, "More rounds than needed.");
  if (challenge) sp_push(S, challenge);
  if (S->round == S->tot_rounds) {
    *has_msg = 0;
    return GM_OK;
  }
  SpReduced R;
  int rc = sp_reduce(C, S, &R);
  if (rc) return rc;
  SpArgs A;
  memset(&A, 0, sizeof A);
  A.f = R.f;
  A.g = R.g;
  A.nf = R.nf;
  A.ng = R.ng;
  A.le = R.le ? 1u : 0u;
  A.wf_lo = S->wf_lo;
  A.wf_hi = nullptr;
  A.wg_lo = S->wg_lo;
  A.wg_hi = nullptr;
  A.k = R.krem;
  A.klo = R.krem;
  gmh::Fr tau = gmh::Fr::from_limbs(S->twist);
  memcpy(A.tau, tau.l, 32);
  make_pow_table(tau.sqr(), A.tau2);
  const size_t nfk = ceil_shift(R.nf, A.k), ngk = ceil_shift(R.ng, A.k);
  const size_t pf = (nfk + 1) / 2, pg = (ngk + 1) / 2;
  A.npairs = pf < pg ? pf : pg;  // the streams are aligned at the low end (space_prover.rs:141-153)
  uint32_t lt = 8;
  while (lt < 17 && ((size_t)1 << lt) < A.npairs) lt++;
  A.log_threads = lt;
  const unsigned blocks = (unsigned)(((size_t)1 << lt) / 256);
  const bool zc = (C->zero_copy & 1) != 0;
  hipLaunchKernelGGL(k_sp_message, dim3(blocks), dim3(256), 0, C->stream, A, zc ? reinterpret_cast<uint8_t*>(S->host_partials) : S->partials);
  GM_HIP(hipGetLastError());
  if (!zc) GM_HIP(hipMemcpyAsync(S->host_partials, S->partials, (size_t)blocks * 2 * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  gmh::Fr a = gmh::Fr::zero(), b = gmh::Fr::zero();
  for (unsigned i = 0; i < blocks; i++) {
    a = a + gmh::Fr::from_limbs(S->host_partials + (size_t)i * 8);
    b = b + gmh::Fr::from_limbs(S->host_partials + (size_t)i * 8 + 4);
  }
  a.to_limbs(a_out);
  b.to_limbs(b_out);
  S->round += 1;
  *has_msg = 1;
  return GM_OK;
}

// folded vectors of the current round, little-endian, into fresh pooled buffers
static int sp_materialize(Context* C, SpaceProver* S, uint8_t** f_out, size_t* nf_out, size_t* fcap, uint8_t** g_out, size_t* ng_out,
                          size_t* gcap) {
  SpReduced R;
  int rc = sp_reduce(C, S, &R);
  if (rc) return rc;
  *nf_out = ceil_shift(R.nf, R.krem);
  *ng_out = ceil_shift(R.ng, R.krem);
  if ((rc = C->pool.alloc(*nf_out * FR_BYTES, (void**)f_out, fcap))) return rc;
  if ((rc = C->pool.alloc(*ng_out * FR_BYTES, (void**)g_out, gcap))) return rc;
  hipLaunchKernelGGL(k_sp_materialize, dim3(grid_for(*nf_out, 1u << 20)), dim3(256), 0, C->stream, R.f, R.nf, S->wf_lo, (const uint8_t*)nullptr, R.krem,
                     R.krem, *nf_out, *f_out, R.le);
  hipLaunchKernelGGL(k_sp_materialize, dim3(grid_for(*ng_out, 1u << 20)), dim3(256), 0, C->stream, R.g, R.ng, S->wg_lo, (const uint8_t*)nullptr, R.krem,
                     R.krem, *ng_out, *g_out, R.le);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

int sp_final(Context* C, SpaceProver* S, uint64_t f0[4], uint64_t g0[4], int* has) {
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->round != S->tot_rounds) {
    *has = 0;
    return GM_OK;
  }
  uint8_t *f, *g;
  size_t nf, ng, fc, gc;
  int rc = sp_materialize(C, S, &f, &nf, &fc, &g, &ng, &gc);
  if (rc) return rc;
  GM_HIP(hipMemcpyAsync(f0, f, FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipMemcpyAsync(g0, g, FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  C->pool.free(f, fc);
  C->pool.free(g, gc);
  *has = 1;
  return GM_OK;
}

// From<&SpaceProver> for TimeProver                                      space_prover.rs:269-307
int sp_to_time(Context* C, SpaceProver* S, uint64_t* time_handle) {
  std::lock_guard<std::mutex> lk(S->mu);
  auto T = std::make_unique<Sumcheck>();
  int rc = sp_materialize(C, S, &T->f[0], &T->nf, &T->fcap[0], &T->g[0], &T->ng, &T->gcap[0]);
  if (rc) return rc;
  if ((rc = C->pool.alloc(((T->nf + 1) / 2) * FR_BYTES, (void**)&T->f[1], &T->fcap[1]))) return rc;
  if ((rc = C->pool.alloc(((T->ng + 1) / 2) * FR_BYTES, (void**)&T->g[1], &T->gcap[1]))) return rc;
  {
    int rcp = C->partial_bufs.take(&T->partials, &T->host_partials);
    if (rcp) return rcp;
  }
  memcpy(T->twist, S->twist, 32);
  T->round = S->round;  // "copy other informations such us round(s) and twist"
  T->tot_rounds = S->tot_rounds;
  *time_handle = put_prover(std::move(T));
  return GM_OK;
}

static int upload_small(Context* C, const void* src, size_t bytes, uint8_t** dptr);
int fr_stride_raw(Context* C, const uint8_t* in, size_t start, size_t stride, size_t count, uint8_t* out) {
  GM_FR_LOCK(C);
  if (count) hipLaunchKernelGGL(k_fr_stride, dim3(grid_for(count)), dim3(256), 0, C->stream, in, start, stride, count, out);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
int fr_fold_raw(Context* C, const uint8_t* f, size_t n, const uint64_t r[4], uint8_t* out) {
  GM_FR_LOCK(C);
  uint8_t* dr;
  int rc = upload_small(C, r, 32, &dr);
  if (rc) return rc;
  const size_t m = (n + 1) / 2;
  if (m) hipLaunchKernelGGL(k_fold, dim3(grid_for(m)), dim3(256), 0, C->stream, f, n, (const uint32_t*)dr, out);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

// ---- vector helper entry points (called from capi.hip) ---------------------------------------
static int upload_small(Context* C, const void* src, size_t bytes, uint8_t** dptr) {
  int rc = C->fr_scratch.ensure(1 << 20);
  if (rc) return rc;
  GM_HIP(hipMemcpyAsync(C->fr_scratch.p, src, bytes, hipMemcpyHostToDevice, C->stream));
  *dptr = C->fr_scratch.as<uint8_t>();
  return GM_OK;
}

// foldings_polynomial (tensorcheck/mod.rs:124-133): outs[j] = fold(outs[j - 1], challenge j), outs[-1] = f.  k launches, ONE wait
// (a folding tree is ~20 tiny launches behind the first two: one wait per level was most of their cost)
int fr_fold_chain(Context* C, FrVec* f, const uint64_t* challenges, size_t k, FrVec** outs) {
  GM_FR_LOCK(C);
  FrVec* cur = f;
  for (size_t j = 0; j < k; j++) {
    const size_t m = (cur->len + 1) / 2;
    GM_CHECK(outs[j]->cap >= m, GM_EINVAL, "fold: output capacity %zu < %zu", outs[j]->cap, m);
    GM_CHECK(outs[j] != cur && outs[j] != f, GM_EINVAL, "fold: output must not alias an input");
    uint8_t* dr;
    int rc = upload_small(C, challenges + 4 * j, 32, &dr);  // same staging slot every level: the copy is ordered behind the previous kernel
    if (rc) return rc;
    if (m) hipLaunchKernelGGL(k_fold, dim3(grid_for(m)), dim3(256), 0, C->stream, cur->d, cur->len, (const uint32_t*)dr, outs[j]->d);
    outs[j]->len = m;
    cur = outs[j];
  }
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

int fr_fold(Context* C, FrVec* f, const uint64_t r[4], FrVec* out) {
  GM_FR_LOCK(C);
  const size_t m = (f->len + 1) / 2;
  GM_CHECK(out->cap >= m, GM_EINVAL, "fold: output capacity %zu < %zu", out->cap, m);
  GM_CHECK(out != f, GM_EINVAL, "fold: output must not alias the input");
  uint8_t* dr;
  int rc = upload_small(C, r, 32, &dr);
  if (rc) return rc;
  if (m) hipLaunchKernelGGL(k_fold, dim3(grid_for(m)), dim3(256), 0, C->stream, f->d, f->len, (const uint32_t*)dr, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = m;
  return GM_OK;
}

int fr_powers(Context* C, const uint64_t x[4], size_t n, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out->cap >= n, GM_EINVAL, "powers: output capacity %zu < %zu", out->cap, n);
  PowTable t;
  make_pow_table(gmh::Fr::from_limbs(x), t);
  if (n) hipLaunchKernelGGL(k_powers, dim3(grid_for(n, 512)), dim3(256), 0, C->stream, t, (size_t)0, n, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = n;
  return GM_OK;
}

// out[i] = x^(start + i), i < n, into raw device memory (the segments of a block-sharded key); start + n < 2^40
int fr_powers_at(Context* C, const uint64_t x[4], size_t start, size_t n, uint8_t* out) {
  GM_FR_LOCK(C);
  GM_CHECK(start < ((size_t)1 << 39) && n < ((size_t)1 << 39), GM_EINVAL, "powers_at: exponent range [%zu, %zu + %zu)", start, start, n);
  PowTable t;
  make_pow_table(gmh::Fr::from_limbs(x), t);
  if (n) hipLaunchKernelGGL(k_powers, dim3(grid_for(n, 512)), dim3(256), 0, C->stream, t, start, n, out);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

// out[i] = tensor(rhos)[start + i], i < count: the whole tensor (start = 0, count = 2^k) or a block of it
int fr_tensor_range(Context* C, const uint64_t* rhos, size_t k, size_t start, size_t count, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(k >= 1 && k <= 32, GM_EINVAL, "tensor: need 1 <= k <= 32 elements (got %zu)", k);
  const size_t n = (size_t)1 << k;
  GM_CHECK(start <= n && count <= n - start, GM_EINVAL, "tensor: range [%zu, %zu + %zu) outside 2^%zu entries", start, start, count, k);
  GM_CHECK(out->cap >= count, GM_EINVAL, "tensor: output capacity %zu < %zu", out->cap, count);
  const uint32_t klo = (uint32_t)(k / 2 > 0 ? (k + 1) / 2 : k), khi = (uint32_t)k - klo;
  int rc = C->fr_scratch.ensure((1 << 20) + (((size_t)1 << klo) + ((size_t)1 << khi)) * FR_BYTES);
  if (rc) return rc;
  uint8_t* base = C->fr_scratch.as<uint8_t>();
  GM_HIP(hipMemcpyAsync(base, rhos, k * 32, hipMemcpyHostToDevice, C->stream));
  uint8_t* lo = base + (1 << 20);
  uint8_t* hi = lo + ((size_t)1 << klo) * FR_BYTES;
  hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << klo)), dim3(256), 0, C->stream, (const uint32_t*)base, klo, lo);
  hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << khi)), dim3(256), 0, C->stream,
                     (const uint32_t*)(base + (size_t)klo * 32), khi, hi);
  if (count) hipLaunchKernelGGL(k_tensor, dim3(grid_for(count)), dim3(256), 0, C->stream, lo, hi, klo, start, count, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = count;
  return GM_OK;
}
int fr_tensor(Context* C, const uint64_t* rhos, size_t k, FrVec* out) {
  GM_CHECK(k >= 1 && k <= 32, GM_EINVAL, "tensor: need 1 <= k <= 32 elements (got %zu)", k);
  return fr_tensor_range(C, rhos, k, 0, (size_t)1 << k, out);
}

// out[i] = tensor(rhos)[index[i]] (k_gather_prod2 on the half tables of the tensor)
int fr_tensor_gather(Context* C, const uint64_t* rhos, size_t k, const IdxVec* index, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(k >= 1 && k <= 32, GM_EINVAL, "tensor_gather: need 1 <= k <= 32 elements (got %zu)", k);
  GM_CHECK(out->cap >= index->n, GM_EINVAL, "tensor_gather: output capacity %zu < %zu", out->cap, index->n);
  GM_CHECK(index->max_plus_1 <= ((size_t)1 << k), GM_EINVAL, "tensor_gather: index %zu outside 2^%zu entries", index->max_plus_1 - 1, k);
  const uint32_t klo = (uint32_t)(k / 2 > 0 ? (k + 1) / 2 : k), khi = (uint32_t)k - klo;
  int rc = C->fr_scratch.ensure((1 << 20) + (((size_t)1 << klo) + ((size_t)1 << khi)) * FR_BYTES);
  if (rc) return rc;
  uint8_t* base = C->fr_scratch.as<uint8_t>();
  GM_HIP(hipMemcpyAsync(base, rhos, k * 32, hipMemcpyHostToDevice, C->stream));
  uint8_t* lo = base + (1 << 20);
  uint8_t* hi = lo + ((size_t)1 << klo) * FR_BYTES;
  hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << klo)), dim3(256), 0, C->stream, (const uint32_t*)base, klo, lo);
  hipLaunchKernelGGL(k_tensor_table, dim3(grid_for((size_t)1 << khi)), dim3(256), 0, C->stream, (const uint32_t*)(base + (size_t)klo * 32), khi, hi);
  if (index->n) hipLaunchKernelGGL(k_gather_prod2, dim3(grid_for(index->n)), dim3(256), 0, C->stream, lo, hi, klo, index->d, index->n, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = index->n;
  return GM_OK;
}
// out[i] = x^index[i], index[i] < 2^k: lo[j] = x^j (j < 2^klo), hi[j] = (x^(2^klo))^j
int fr_powers_gather(Context* C, const uint64_t x[4], size_t k, const IdxVec* index, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(k >= 1 && k <= 32, GM_EINVAL, "powers_gather: need 1 <= k <= 32 (got %zu)", k);
  GM_CHECK(out->cap >= index->n, GM_EINVAL, "powers_gather: output capacity %zu < %zu", out->cap, index->n);
  GM_CHECK(index->max_plus_1 <= ((size_t)1 << k), GM_EINVAL, "powers_gather: index %zu outside 2^%zu entries", index->max_plus_1 - 1, k);
  const uint32_t klo = (uint32_t)(k / 2 > 0 ? (k + 1) / 2 : k), khi = (uint32_t)k - klo;
  int rc = C->fr_scratch.ensure((1 << 20) + (((size_t)1 << klo) + ((size_t)1 << khi)) * FR_BYTES);
  if (rc) return rc;
  uint8_t* lo = C->fr_scratch.as<uint8_t>() + (1 << 20);
  uint8_t* hi = lo + ((size_t)1 << klo) * FR_BYTES;
  gmh::Fr xx = gmh::Fr::from_limbs(x), step = xx;
  for (uint32_t b = 0; b < klo; b++) step = step.sqr();  // x^(2^klo)
  PowTable t;
  make_pow_table(xx, t);
  hipLaunchKernelGGL(k_powers, dim3(grid_for((size_t)1 << klo, 512)), dim3(256), 0, C->stream, t, (size_t)0, (size_t)1 << klo, lo);
  make_pow_table(step, t);
  hipLaunchKernelGGL(k_powers, dim3(grid_for((size_t)1 << khi, 512)), dim3(256), 0, C->stream, t, (size_t)0, (size_t)1 << khi, hi);
  if (index->n) hipLaunchKernelGGL(k_gather_prod2, dim3(grid_for(index->n)), dim3(256), 0, C->stream, lo, hi, klo, index->d, index->n, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = index->n;
  return GM_OK;
}

int fr_hadamard(Context* C, FrVec* a, FrVec* b, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(a->len == b->len, GM_EINVAL, "hadamard: lengths differ (%zu vs %zu)", a->len, b->len);
  GM_CHECK(out->cap >= a->len, GM_EINVAL, "hadamard: output capacity %zu < %zu", out->cap, a->len);
  if (a->len) hipLaunchKernelGGL(k_hadamard, dim3(grid_for(a->len)), dim3(256), 0, C->stream, a->d, b->d, a->len, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = a->len;
  return GM_OK;
}

int fr_ip(Context* C, FrVec* a, FrVec* b, uint64_t result[4]) {
  GM_FR_LOCK(C);
  GM_CHECK(a->len == b->len, GM_EINVAL, "ip: lengths differ (%zu vs %zu)", a->len, b->len);
  const unsigned blocks = grid_for(a->len, 512);
  int rc = C->fr_scratch.ensure(1 << 20);
  if (rc) return rc;
  const bool zc = (C->zero_copy & 1) != 0;
  hipLaunchKernelGGL(k_ip, dim3(blocks), dim3(256), 0, C->stream, a->d, b->d, a->len,
                     zc ? reinterpret_cast<uint8_t*>(C->host_small) : C->fr_scratch.as<uint8_t>());
  GM_HIP(hipGetLastError());
  if (!zc) GM_HIP(hipMemcpyAsync(C->host_small, C->fr_scratch.p, (size_t)blocks * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  gmh::Fr s = gmh::Fr::zero();
  for (unsigned i = 0; i < blocks; i++) s = s + gmh::Fr::from_limbs(C->host_small + (size_t)i * 4);
  s.to_limbs(result);
  return GM_OK;
}

// point k == -point j (j < k) as field elements
static void eval_mark_negations(EvalArgs& A, const uint64_t* xs, size_t npoints) {
  for (size_t k = 1; k < npoints; k++)
    for (size_t j = 0; j < k; j++) {
      if (A.neg_of[j]) continue;
      const gmh::Fr s = gmh::Fr::from_limbs(xs + 4 * k) + gmh::Fr::from_limbs(xs + 4 * j);
      if (s.is_zero()) {
        A.neg_of[k] = (uint32_t)j + 1;
        break;
      }
    }
}

int fr_eval_le(Context* C, FrVec* p, const uint64_t* xs, size_t npoints, uint64_t* results) {
  GM_FR_LOCK(C);
  GM_CHECK(npoints >= 1 && npoints <= 3, GM_EINVAL, "eval_le: 1..3 points per pass (got %zu)", npoints);
  EvalArgs A;
  memset(&A, 0, sizeof A);
  for (size_t k = 0; k < npoints; k++) make_pow_table(gmh::Fr::from_limbs(xs + 4 * k), A.xt[k]);
  A.npoints = (uint32_t)npoints;
  eval_mark_negations(A, xs, npoints);
  uint32_t lt = 8;
  while (lt < 17 && ((size_t)1 << lt) < p->len) lt++;
  A.log_threads = lt;
  const unsigned blocks = (unsigned)(((size_t)1 << lt) / 256);
  int rc = C->fr_scratch.ensure(1 << 20);
  if (rc) return rc;
  const bool zc = (C->zero_copy & 1) != 0;
  hipLaunchKernelGGL(k_eval_le, dim3(blocks), dim3(256), 0, C->stream, p->d, p->len, A,
                     zc ? reinterpret_cast<uint8_t*>(C->host_small) : C->fr_scratch.as<uint8_t>());
  GM_HIP(hipGetLastError());
  if (!zc) GM_HIP(hipMemcpyAsync(C->host_small, C->fr_scratch.p, (size_t)blocks * 3 * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  for (size_t k = 0; k < npoints; k++) {
    gmh::Fr s = gmh::Fr::zero();
    for (unsigned i = 0; i < blocks; i++) s = s + gmh::Fr::from_limbs(C->host_small + ((size_t)i * 3 + k) * 4);
    s.to_limbs(results + 4 * k);
  }
  return GM_OK;
}

// k polynomials at the SAME <= 3 points: all kernels and one copy are enqueued before the single wait (the foldings of
// the tensor check are evaluated at +-beta one after the other in the reference, tensorcheck/mod.rs:228-247; 23 waits
// per proof at 2^24).  results: k x npoints x 4 limbs.
int fr_eval_le_batch(Context* C, FrVec* const* ps, size_t k, const uint64_t* xs, size_t npoints, uint64_t* results) {
  GM_FR_LOCK(C);
  GM_CHECK(npoints >= 1 && npoints <= 3, GM_EINVAL, "eval_le_batch: 1..3 points per pass (got %zu)", npoints);
  if (k == 0) return GM_OK;
  EvalArgs A;
  memset(&A, 0, sizeof A);
  for (size_t j = 0; j < npoints; j++) make_pow_table(gmh::Fr::from_limbs(xs + 4 * j), A.xt[j]);
  A.npoints = (uint32_t)npoints;
  eval_mark_negations(A, xs, npoints);
  const size_t slot = (size_t)512 * 3 * FR_BYTES;  // per polynomial: <= 2^17 threads = 512 blocks
  const size_t jobs_bytes = k * sizeof(EvalJob);
  int rc = C->fr_scratch.ensure(k * slot + jobs_bytes);
  if (rc) return rc;
  if (C->host_batch_cap < k * slot + jobs_bytes) {
    if (C->host_batch) (void)hipHostFree(C->host_batch);
    C->host_batch = nullptr;
    C->host_batch_cap = 0;
    GM_HIP(hipHostMalloc((void**)&C->host_batch, k * slot + jobs_bytes, hipHostMallocDefault));
    C->host_batch_cap = k * slot + jobs_bytes;
  }
  std::vector<unsigned> nblocks(k);
  const bool zc = (C->zero_copy & 1) != 0;
  // the job list rides in page-locked memory behind the result slots; one copy, one launch
  EvalJob* h_jobs = reinterpret_cast<EvalJob*>(reinterpret_cast<uint8_t*>(C->host_batch) + k * slot);
  EvalJob* d_jobs = reinterpret_cast<EvalJob*>(C->fr_scratch.as<uint8_t>() + k * slot);
  uint32_t total_blocks = 0;
  for (size_t j = 0; j < k; j++) {
    uint32_t lt = 8;
    while (lt < 17 && ((size_t)1 << lt) < ps[j]->len) lt++;
    nblocks[j] = (unsigned)(((size_t)1 << lt) / 256);
    h_jobs[j].p = ps[j]->d;
    h_jobs[j].n = ps[j]->len;
    h_jobs[j].out_off = j * slot;
    h_jobs[j].log_threads = lt;
    h_jobs[j].blk_start = total_blocks;
    total_blocks += nblocks[j];
  }
  A.log_threads = 0;  // per job
  GM_HIP(hipMemcpyAsync(d_jobs, h_jobs, jobs_bytes, hipMemcpyHostToDevice, C->stream));
  hipLaunchKernelGGL(k_eval_le_multi, dim3(total_blocks), dim3(256), 0, C->stream, d_jobs, (uint32_t)k, A,
                     zc ? reinterpret_cast<uint8_t*>(C->host_batch) : C->fr_scratch.as<uint8_t>());
  if (!zc)
    for (size_t j = 0; j < k; j++)
      GM_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t*>(C->host_batch) + j * slot, C->fr_scratch.as<uint8_t>() + j * slot, (size_t)nblocks[j] * 3 * FR_BYTES,
                            hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  for (size_t j = 0; j < k; j++) {
    const uint64_t* h = C->host_batch + j * (slot / 8);
    for (size_t q = 0; q < npoints; q++) {
      gmh::Fr s = gmh::Fr::zero();
      for (unsigned i = 0; i < nblocks[j]; i++) s = s + gmh::Fr::from_limbs(h + ((size_t)i * 3 + q) * 4);
      s.to_limbs(results + (j * npoints + q) * 4);
    }
  }
  return GM_OK;
}

int fr_trim(Context* C, FrVec* v) {
  GM_FR_LOCK(C);
  // DensePolynomial::from_coefficients_vec strips high zero coefficients
  if (v->len == 0) return GM_OK;
  int rc = C->fr_scratch.ensure(1 << 20);
  if (rc) return rc;
  // the top 2^16 coefficients first: the leading coefficient of a polynomial a prover builds is almost never zero, and a scan of
  // the whole vector reads 512 MB at 2^24 coefficients (0.18 ms, three times per `snark -i 24` proof)
  const size_t top = v->len < ((size_t)1 << 16) ? v->len : ((size_t)1 << 16);
  size_t lo = v->len - top, hi = v->len;
  for (;;) {
    GM_HIP(hipMemsetAsync(C->fr_scratch.p, 0, 8, C->stream));
    hipLaunchKernelGGL(k_high_nonzero, dim3(grid_for(hi - lo)), dim3(256), 0, C->stream, v->d, lo, hi, C->fr_scratch.as<unsigned long long>());
    GM_HIP(hipGetLastError());
    GM_HIP(hipMemcpyAsync(C->host_small, C->fr_scratch.p, 8, hipMemcpyDeviceToHost, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
    if (C->host_small[0] != 0 || lo == 0) break;
    hi = lo;  // nothing up there: the rest of the vector
    lo = 0;
  }
  v->len = (size_t)C->host_small[0];
  return GM_OK;
}

int fr_lincomb(Context* C, FrVec** polys, const uint64_t* coeffs, size_t k, FrVec* out) {
  GM_FR_LOCK(C);
  size_t n = 0;
  for (size_t j = 0; j < k; j++) n = polys[j]->len > n ? polys[j]->len : n;
  GM_CHECK(out->cap >= n, GM_EINVAL, "lincomb: output capacity %zu < %zu", out->cap, n);
  for (size_t j = 0; j < k; j++) GM_CHECK(polys[j] != out, GM_EINVAL, "lincomb: output must not alias an input");
  if (n == 0) {
    out->len = 0;
    return GM_OK;
  }
  for (size_t j0 = 0; j0 < k || j0 == 0; j0 += LINCOMB_MAX) {
    LincombArgs A;
    memset(&A, 0, sizeof A);
    size_t cnt = k - j0 < LINCOMB_MAX ? k - j0 : LINCOMB_MAX;
    A.k = (uint32_t)cnt;
    for (size_t j = 0; j < cnt; j++) {
      A.p[j] = polys[j0 + j]->d;
      A.len[j] = polys[j0 + j]->len;
      memcpy(A.c[j], coeffs + 4 * (j0 + j), 32);
    }
    if (j0 == 0)
      hipLaunchKernelGGL(k_lincomb, dim3(grid_for(n)), dim3(256), 0, C->stream, A, n, out->d);
    else
      hipLaunchKernelGGL(k_lincomb_acc, dim3(grid_for(n)), dim3(256), 0, C->stream, A, n, out->d);
    if (k == 0) break;
  }
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = n;
  return fr_trim(C, out);
}

// out[offsets[j] + i] = c_j * in_j[i], i < len(in_j): several scaled vectors laid out in ONE vector at chosen offsets (the quotients
// of the block-sharded opening against the back-to-back key slices: one division and one MSM instead of one per level).  k launches,
// ONE wait.  The ranges must lie inside out's current length (they may not overlap each other: the launches are not ordered by data)
int fr_scale_into_many(Context* C, FrVec** ins, const uint64_t* coeffs, size_t k, FrVec* out, const size_t* offsets) {
  GM_FR_LOCK(C);
  for (size_t j = 0; j < k; j++) {
    GM_CHECK(ins[j] != out, GM_EINVAL, "scale_into: output must not alias an input");
    GM_CHECK(offsets[j] <= out->len && ins[j]->len <= out->len - offsets[j], GM_EINVAL, "scale_into: [%zu, %zu) outside a vector of length %zu", offsets[j],
             offsets[j] + ins[j]->len, out->len);
  }
  bool any = false;
  for (size_t j = 0; j < k; j++) {
    if (ins[j]->len == 0) continue;
    LincombArgs A;
    memset(&A, 0, sizeof A);
    A.k = 1;
    A.p[0] = ins[j]->d;
    A.len[0] = ins[j]->len;
    memcpy(A.c[0], coeffs + 4 * j, 32);
    hipLaunchKernelGGL(k_lincomb, dim3(grid_for(ins[j]->len)), dim3(256), 0, C->stream, A, ins[j]->len, out->d + offsets[j] * FR_BYTES);
    any = true;
  }
  GM_HIP(hipGetLastError());
  if (any) GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
int fr_scale_into(Context* C, FrVec* in, const uint64_t c[4], FrVec* out, size_t offset) { return fr_scale_into_many(C, &in, c, 1, out, &offset); }

// v[idx[j]] += vals[j]; the positions must be distinct (each is updated by its own lane) and inside the vector
int fr_add_at(Context* C, FrVec* v, const size_t* idx, const uint64_t* vals, size_t k) {
  GM_FR_LOCK(C);
  GM_CHECK(k <= 4096, GM_EINVAL, "add_at: %zu positions (at most 4096)", k);
  if (k == 0) return GM_OK;
  std::vector<unsigned long long> packed(k * 5);
  for (size_t j = 0; j < k; j++) {
    GM_CHECK(idx[j] < v->len, GM_EINVAL, "add_at: position %zu outside a vector of length %zu", idx[j], v->len);
    packed[j] = idx[j];
  }
  {
    std::vector<size_t> sorted(idx, idx + k);
    std::sort(sorted.begin(), sorted.end());
    GM_CHECK(std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end(), GM_EINVAL, "add_at: repeated position");
  }
  memcpy(packed.data() + k, vals, k * 32);
  uint8_t* d;
  int rc = upload_small(C, packed.data(), k * 40, &d);
  if (rc) return rc;
  hipLaunchKernelGGL(k_add_at, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, C->stream, v->d, (const unsigned long long*)d, d + k * 8, (unsigned)k);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

int spm_mul(Context* C, SparseMatrix* M, FrVec* x, FrVec* y) {
  GM_FR_LOCK(C);
  GM_CHECK(y->cap >= M->nrows, GM_EINVAL, "spm_mul: output capacity %zu < %zu rows", y->cap, M->nrows);
  GM_CHECK(y != x, GM_EINVAL, "spm_mul: output must not alias the input");
  if (M->nrows)
    hipLaunchKernelGGL(k_spmv, dim3(grid_for(M->nrows)), dim3(256), 0, C->stream, M->rowptr, M->cols, M->vals, M->nrows, x->d, x->len, y->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  y->len = M->nrows;
  return GM_OK;
}

// ---- entry-product / plookup builders ---------------------------------------------------------
int fr_gather(Context* C, FrVec* src, const IdxVec* index, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out->cap >= index->n, GM_EINVAL, "gather: output capacity %zu < %zu", out->cap, index->n);
  GM_CHECK(out != src, GM_EINVAL, "gather: output must not alias the input");
  GM_CHECK(index->max_plus_1 <= src->len, GM_EINVAL, "gather: index %zu outside the source vector (%zu)", index->max_plus_1 - 1, src->len);
  if (index->n) hipLaunchKernelGGL(k_gather, dim3(grid_for(index->n)), dim3(256), 0, C->stream, src->d, index->d, index->n, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = index->n;
  return GM_OK;
}

int fr_alg_hash(Context* C, FrVec* v, const IdxVec* index, const uint64_t zeta[4], FrVec* out, uint64_t base) {
  GM_FR_LOCK(C);
  // zip semantics of the reference: the shorter of (v, index) decides the length
  const size_t n = index ? std::min(v->len, index->n) : v->len;
  GM_CHECK(out->cap >= n, GM_EINVAL, "alg_hash: output capacity %zu < %zu", out->cap, n);
  gmh::Fr z2 = gmh::Fr::from_limbs(zeta) * gmh::Fr::from_limbs(gmh::FrP::R2);
  uint8_t* dz;
  int rc = upload_small(C, z2.l, 32, &dz);
  if (rc) return rc;
  if (n) hipLaunchKernelGGL(k_alg_hash, dim3(grid_for(n)), dim3(256), 0, C->stream, v->d, index ? index->d : nullptr, n, base, (const uint32_t*)dz, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = n;
  return GM_OK;
}

int fr_plookup_set(Context* C, FrVec* v, const uint64_t y[4], const uint64_t z[4], FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out != v, GM_EINVAL, "plookup_set: output must not alias the input");
  if (v->len == 0) {
    out->len = 0;
    return GM_OK;
  }
  GM_CHECK(out->cap >= v->len + 1, GM_EINVAL, "plookup_set: output capacity %zu < %zu", out->cap, v->len + 1);
  gmh::Fr zz = gmh::Fr::from_limbs(z);
  gmh::Fr y1z = (gmh::Fr::one() + zz) * gmh::Fr::from_limbs(y);
  uint64_t small[8];
  memcpy(small, y1z.l, 32);
  memcpy(small + 4, zz.l, 32);
  uint8_t* d;
  int rc = upload_small(C, small, 64, &d);
  if (rc) return rc;
  hipLaunchKernelGGL(k_plookup_set, dim3(grid_for(v->len + 1)), dim3(256), 0, C->stream, v->d, v->len, (const uint32_t*)d, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = v->len + 1;
  return GM_OK;
}

int fr_add_scalar(Context* C, FrVec* v, const uint64_t y[4], FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out->cap >= v->len, GM_EINVAL, "add_scalar: output capacity %zu < %zu", out->cap, v->len);
  uint8_t* d;
  int rc = upload_small(C, y, 32, &d);
  if (rc) return rc;
  if (v->len) hipLaunchKernelGGL(k_add_scalar, dim3(grid_for(v->len)), dim3(256), 0, C->stream, v->d, v->len, (const uint32_t*)d, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = v->len;
  return GM_OK;
}

int fr_shift_monic(Context* C, FrVec* v, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out->cap >= v->len + 1, GM_EINVAL, "shift_monic: output capacity %zu < %zu", out->cap, v->len + 1);
  GM_CHECK(out != v, GM_EINVAL, "shift_monic: output must not alias the input");
  hipLaunchKernelGGL(k_shift_monic, dim3(grid_for(v->len + 1)), dim3(256), 0, C->stream, v->d, v->len, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = v->len + 1;
  return GM_OK;
}

// One block of a block-sharded vector as well (gm_psnark_new_time_sharded): carry_in = the product of the blocks ABOVE this one
// (null: one), write_monic = whether position n of the whole vector -- the entry 1 -- falls into this block; total (optional) =
// the product of this block WITHOUT the carry; out = nullptr: the product only (phases 1 and 2).
int fr_acc_product_ex(Context* C, FrVec* v, const uint64_t* carry_in, bool write_monic, FrVec* out, uint64_t* total) {
  GM_FR_LOCK(C);
  const size_t n = v->len;
  const gmh::Fr cin = carry_in ? gmh::Fr::from_limbs(carry_in) : gmh::Fr::one();
  if (out) {
    GM_CHECK(out->cap >= n + (write_monic ? 1 : 0), GM_EINVAL, "acc_product: output capacity %zu < %zu", out->cap, n + (write_monic ? 1 : 0));
    GM_CHECK(out != v, GM_EINVAL, "acc_product: output must not alias the input");
    out->len = n + (write_monic ? 1 : 0);
  }
  if (n == 0) {
    if (total) gmh::Fr::one().to_limbs(total);
    if (out && write_monic) {
      GM_HIP(hipMemcpyAsync(out->d, cin.l, 32, hipMemcpyHostToDevice, C->stream));
      GM_HIP(hipStreamSynchronize(C->stream));
    }
    return GM_OK;
  }
  const size_t nch = (n + ACC_K - 1) / ACC_K, seg = 64, nseg = (nch + seg - 1) / seg;
  int rc = C->fr_scratch.ensure((1 << 20) + (nch + 2 * nseg + 8) * FR_BYTES);
  if (rc) return rc;
  uint8_t* prods = C->fr_scratch.as<uint8_t>() + (1 << 20);
  uint8_t* seg_prods = prods + nch * FR_BYTES;
  uint8_t* carry = seg_prods + nseg * FR_BYTES;  // nseg segment carries + the carry of the top chunk
  // page-locked staging as in fr_div_linear_factors: [segment products][carries]
  if (C->host_batch_cap < (2 * nseg + 1) * FR_BYTES) {
    if (C->host_batch) (void)hipHostFree(C->host_batch);
    C->host_batch = nullptr;
    C->host_batch_cap = 0;
    GM_HIP(hipHostMalloc((void**)&C->host_batch, (2 * nseg + 1) * FR_BYTES, hipHostMallocDefault));
    C->host_batch_cap = (2 * nseg + 1) * FR_BYTES;
  }
  uint64_t* hs = C->host_batch;
  uint64_t* hc = hs + nseg * 4;
  const bool zc = (C->zero_copy & 1) != 0;
  hipLaunchKernelGGL(k_accp_phase1, dim3(grid_for(nch, 1u << 22)), dim3(256), 0, C->stream, v->d, n, prods);
  hipLaunchKernelGGL(k_accp_phase2a, dim3(grid_for(nseg, 1u << 22)), dim3(256), 0, C->stream, prods, nch, seg,
                     zc ? reinterpret_cast<uint8_t*>(hs) : seg_prods);
  GM_HIP(hipGetLastError());
  {
    // carry[s] = product of the segments above s (times the carry of the blocks above): nseg <= n / 4096 sequential host multiplications
    if (!zc) GM_HIP(hipMemcpyAsync(hs, seg_prods, nseg * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
    GM_HIP(hipStreamSynchronize(C->stream));
    gmh::Fr acc = cin, plain = gmh::Fr::one();
    cin.to_limbs(hc + 4 * nseg);
    for (size_t s = nseg; s-- > 0;) {
      acc.to_limbs(hc + 4 * s);
      const gmh::Fr p = gmh::Fr::from_limbs(hs + 4 * s);
      acc = acc * p;
      plain = plain * p;
    }
    if (total) plain.to_limbs(total);
    if (!out) return GM_OK;
    GM_HIP(hipMemcpyAsync(carry, hc, (nseg + 1) * FR_BYTES, hipMemcpyHostToDevice, C->stream));
  }
  hipLaunchKernelGGL(k_accp_phase3, dim3(grid_for(nch, 1u << 22)), dim3(256), 0, C->stream, v->d, n, prods, carry, seg,
                     carry_in ? carry + nseg * FR_BYTES : (const uint8_t*)nullptr, write_monic ? 1 : 0, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
int fr_acc_product(Context* C, FrVec* v, FrVec* out) { return fr_acc_product_ex(C, v, nullptr, true, out, nullptr); }

// plookup_set on a block (k_plookup_set_block): v_off / nv = the elements of `v` that are this block's part of the hashed set,
// prev (host, 4 limbs) = the element just below the block or null
int fr_plookup_set_block(Context* C, FrVec* v, size_t v_off, size_t nv, const uint64_t* prev, size_t nout, const uint64_t y[4], const uint64_t z[4],
                         FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out != v, GM_EINVAL, "plookup_set_block: output must not alias the input");
  GM_CHECK(v_off <= v->len && nv <= v->len - v_off, GM_EINVAL, "plookup_set_block: elements [%zu, %zu + %zu) outside a vector of %zu", v_off, v_off, nv, v->len);
  GM_CHECK(out->cap >= nout && nout <= nv + 1, GM_EINVAL, "plookup_set_block: %zu outputs (capacity %zu) from %zu elements", nout, out->cap, nv);
  out->len = nout;
  if (nout == 0) return GM_OK;
  gmh::Fr zz = gmh::Fr::from_limbs(z);
  gmh::Fr y1z = (gmh::Fr::one() + zz) * gmh::Fr::from_limbs(y);
  uint64_t small[12];
  memcpy(small, y1z.l, 32);
  memcpy(small + 4, zz.l, 32);
  if (prev) memcpy(small + 8, prev, 32);
  uint8_t* d;
  int rc = upload_small(C, small, 96, &d);
  if (rc) return rc;
  hipLaunchKernelGGL(k_plookup_set_block, dim3(grid_for(nout)), dim3(256), 0, C->stream, v->d + v_off * FR_BYTES, nv,
                     prev ? (const uint32_t*)(d + 64) : (const uint32_t*)nullptr, nout, (const uint32_t*)d, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}
// out[0] = first, out[t] = v[t - 1], t < nout <= |v| + 1
int fr_shift_block(Context* C, FrVec* v, const uint64_t first[4], size_t nout, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out != v, GM_EINVAL, "shift_block: output must not alias the input");
  GM_CHECK(out->cap >= nout && nout <= v->len + 1, GM_EINVAL, "shift_block: %zu outputs (capacity %zu) from %zu elements", nout, out->cap, v->len);
  out->len = nout;
  if (nout == 0) return GM_OK;
  uint8_t* d;
  int rc = upload_small(C, first, 32, &d);
  if (rc) return rc;
  hipLaunchKernelGGL(k_shift_block, dim3(grid_for(nout)), dim3(256), 0, C->stream, v->d, (const uint32_t*)d, nout, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

int fr_reverse(Context* C, FrVec* in, FrVec* out) {
  GM_FR_LOCK(C);
  GM_CHECK(out->cap >= in->len, GM_EINVAL, "reverse: output capacity %zu < %zu", out->cap, in->len);
  GM_CHECK(out != in, GM_EINVAL, "reverse: output must not alias the input");
  if (in->len) hipLaunchKernelGGL(k_reverse, dim3(grid_for(in->len)), dim3(256), 0, C->stream, in->d, in->len, out->d);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  out->len = in->len;
  return GM_OK;
}

int fr_fill(Context* C, FrVec* v, const uint64_t val[4]) {
  GM_FR_LOCK(C);
  uint8_t* dv;
  int rc = upload_small(C, val, 32, &dv);
  if (rc) return rc;
  if (v->len) hipLaunchKernelGGL(k_fill, dim3(grid_for(v->len)), dim3(256), 0, C->stream, v->d, v->len, (const uint32_t*)dv);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(C->stream));
  return GM_OK;
}

// quotient of f by prod_j (x - points[j]): k successive divisions by linear factors.
// rem_out[j] = value of the j-th intermediate polynomial at points[j] (= f(points[0]) for j = 0).
int fr_div_linear_factors(Context* C, FrVec* f, const uint64_t* points, size_t k, FrVec* q, uint64_t* rem_out) {
  GM_FR_LOCK(C);
  GM_CHECK(k >= 1 && k <= 8, GM_EINVAL, "div_vanishing: 1..8 points (got %zu)", k);
  GM_CHECK(q != f, GM_EINVAL, "div_vanishing: quotient must not alias the dividend");
  if (f->len <= k) {
    // polynomial no longer than the point set: the quotient is empty and the successive remainders are
    // computed on the host (<= 8 coefficients).  CommitterKey::open([c], x) must return c
    // (src/kzg/time.rs:112-131), open_multi_points the polynomial itself as remainder (:134-145).
    std::vector<gmh::Fr> cur(f->len);
    if (f->len) {
      std::vector<uint64_t> h(4 * f->len);
      GM_HIP(hipMemcpyAsync(h.data(), f->d, f->len * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
      GM_HIP(hipStreamSynchronize(C->stream));
      for (size_t i = 0; i < f->len; i++) cur[i] = gmh::Fr::from_limbs(h.data() + 4 * i);
    }
    for (size_t j = 0; j < k; j++) {
      gmh::Fr rem = gmh::Fr::zero();
      if (!cur.empty()) {
        const gmh::Fr alpha = gmh::Fr::from_limbs(points + 4 * j);
        std::vector<gmh::Fr> qn(cur.size() - 1);
        gmh::Fr carry = gmh::Fr::zero();  // synthetic division from the leading coefficient down
        for (size_t i = cur.size(); i-- > 0;) {
          carry = cur[i] + alpha * carry;
          if (i > 0) qn[i - 1] = carry;
        }
        rem = carry;
        cur.swap(qn);
      }
      if (rem_out) rem.to_limbs(rem_out + 4 * j);
    }
    q->len = 0;
    return GM_OK;
  }
  GM_CHECK(q->cap >= f->len - 1, GM_EINVAL, "div_vanishing: quotient capacity %zu < %zu", q->cap, f->len - 1);
  const size_t n0 = f->len;
  const size_t nch0 = (n0 + DIV_K - 1) / DIV_K;
  const size_t seg = 64;
  const size_t nseg0 = (nch0 + seg - 1) / seg;
  // scratch: [1 MiB small][chunk sums][seg sums][carry][tmp poly for odd passes]
  size_t need = (1 << 20) + (nch0 + 2 * nseg0 + 8) * FR_BYTES + n0 * FR_BYTES;
  int rc = C->fr_scratch.ensure(need);
  if (rc) return rc;
  uint8_t* base = C->fr_scratch.as<uint8_t>();
  uint8_t* sums = base + (1 << 20);
  uint8_t* seg_sums = sums + nch0 * FR_BYTES;
  uint8_t* carry = seg_sums + nseg0 * FR_BYTES;
  uint8_t* tmp = carry + (nseg0 + 8) * FR_BYTES;
  // page-locked staging: [alpha_j, alpha_j^64 of every pass][segment sums][carries] -- the parameters of all passes go up in one
  // copy, phase 2a writes its segment sums straight into host memory, and the carries go back from page-locked memory, so a
  // pass waits ONCE (for the segment sums) instead of three times around two pageable copies
  const size_t stage_bytes = k * 64 + 2 * nseg0 * FR_BYTES;
  if (C->host_batch_cap < stage_bytes) {
    if (C->host_batch) (void)hipHostFree(C->host_batch);
    C->host_batch = nullptr;
    C->host_batch_cap = 0;
    GM_HIP(hipHostMalloc((void**)&C->host_batch, stage_bytes, hipHostMallocDefault));
    C->host_batch_cap = stage_bytes;
  }
  uint64_t* h_par = C->host_batch;
  uint64_t* h_seg = h_par + k * 8;
  uint64_t* h_carry = h_seg + nseg0 * 4;
  const bool zc = (C->zero_copy & 1) != 0;
  uint8_t* d_par = base + 4096;  // k <= 8 passes x 64 bytes, inside the 1 MiB head of the scratch
  std::vector<gmh::Fr> ms(k);
  for (size_t j = 0; j < k; j++) {
    const gmh::Fr alpha = gmh::Fr::from_limbs(points + 4 * j);
    gmh::Fr m = alpha;
    for (int s6 = 0; s6 < 6; s6++) m = m.sqr();  // alpha^64 = alpha^DIV_K
    ms[j] = m;
    memcpy(h_par + 8 * j, alpha.l, 32);
    memcpy(h_par + 8 * j + 4, m.l, 32);
  }
  GM_HIP(hipMemcpyAsync(d_par, h_par, k * 64, hipMemcpyHostToDevice, C->stream));
  const uint8_t* src = f->d;
  size_t n = n0;
  for (size_t j = 0; j < k; j++) {
    // ping-pong so the final quotient lands in q: passes write q, tmp, q, ... ending in q
    uint8_t* dst = ((k - 1 - j) % 2 == 0) ? q->d : tmp;
    const gmh::Fr m = ms[j];
    PowTable mt;
    make_pow_table(m, mt);
    const uint32_t* d_alpha = reinterpret_cast<const uint32_t*>(d_par + 64 * j);
    const size_t nch = (n + DIV_K - 1) / DIV_K, nseg = (nch + seg - 1) / seg;
    hipLaunchKernelGGL(k_div_phase1, dim3(grid_for(nch, 1u << 22)), dim3(256), 0, C->stream, src, n, d_alpha, sums);
    hipLaunchKernelGGL(k_div_phase2a, dim3(grid_for(nseg, 1u << 22)), dim3(256), 0, C->stream, sums, nch, seg, d_alpha + 8,
                       zc ? reinterpret_cast<uint8_t*>(h_seg) : seg_sums);
    {
      // phase 2b on the host: nseg (<= n/4096) sequential steps of a first-order recurrence cost
      // microseconds on a CPU core and milliseconds on a single GPU lane
      if (!zc) GM_HIP(hipMemcpyAsync(h_seg, seg_sums, nseg * FR_BYTES, hipMemcpyDeviceToHost, C->stream));
      GM_HIP(hipStreamSynchronize(C->stream));  // (also: phase 3 of the previous pass has read the carries that are overwritten now)
      gmh::Fr acc = gmh::Fr::zero();
      auto mpow = [&](size_t e) {
        gmh::Fr r = gmh::Fr::one(), b = m;
        while (e) {
          if (e & 1) r = r * b;
          b = b.sqr();
          e >>= 1;
        }
        return r;
      };
      const gmh::Fr mseg = mpow(seg);
      for (size_t sgi = nseg; sgi-- > 0;) {
        acc.to_limbs(h_carry + 4 * sgi);
        const size_t len = std::min(seg, nch - sgi * seg);
        acc = acc * (len == seg ? mseg : mpow(len)) + gmh::Fr::from_limbs(h_seg + 4 * sgi);
      }
      GM_HIP(hipMemcpyAsync(carry, h_carry, nseg * FR_BYTES, hipMemcpyHostToDevice, C->stream));
    }
    hipLaunchKernelGGL(k_div_phase3, dim3(grid_for(nch, 1u << 22)), dim3(256), 0, C->stream, src, n, d_alpha, sums, carry, seg, mt, dst,
                       base + 128 + j * 32);
    GM_HIP(hipGetLastError());
    src = dst;
    n -= 1;
  }
  if (rem_out) GM_HIP(hipMemcpyAsync(rem_out, base + 128, k * 32, hipMemcpyDeviceToHost, C->stream));
  GM_HIP(hipStreamSynchronize(C->stream));
  q->len = n;
  return GM_OK;
}

}  // namespace gm

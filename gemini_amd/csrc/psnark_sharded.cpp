// psnark::Proof::new_time (src/psnark/time_prover.rs:69-384) with EVERY vector of the prover block-sharded over the ranks of
// gm_dist (one process per GPU): the field side of BASELINE configs[4] on N GPUs, not only its MSMs.
//
// Layout.  One block size M for the proof (a multiple of a power of two, gm_psnark_shard_block) and a LEVEL per family of vectors: a vector of
// level s lives in blocks of M >> s -- rank r holds its elements [r (M >> s), (r + 1) (M >> s)) that exist -- with s the largest level whose g
// blocks still hold the family's longest member (gm_psnark_shard_level).  The prover's vectors come in several lengths (dummy_r1cs: n + 1 / n + 2
// for the sets and the nnz-long vectors, 2 n + 2 for the sorted ones; a general instance: n, nnz ~ 6 n, n + nnz): with ONE block size the short
// ones would sit on the lower ranks only (the busiest rank at 46 / 33 of the average MSM work); with levels every rank holds ~1 / g of EVERY vector.
// Members of a family (a lookup vector, its accumulated product and its rotation; everything indexed by the joint support) share a level, so
// everything element-wise stays local.  A folding level j of a polynomial of level s is a vector of level s + j (sharded while s + j <= jmax,
// gathered after), committed against key slice s + j: levels and folding levels are the same thing to the key.  Linear combinations ACROSS
// levels (the batched bodies of the tensor check, the opened polynomials) are done per level and the partial sums RE-BLOCKED to the coarsest
// one (gm_dist_reblock_vecs: every element crosses one link once).
// What crosses ranks, all through gm_dist's all-gather:
//   lookups              `lookup(v, index)` (plookup/time_prover.rs:5-8) gathers from tensor(rho), powers(alpha) and z.  The first two are
//                        FUNCTIONS of the index: NO rank ever builds them -- a looked-up element is one multiplication from two half tables
//                        that stay in L2 (gm_fr_tensor_gather / gm_fr_powers_gather), a contiguous range comes from gm_fr_tensor_range /
//                        gm_fr_powers_range -- and z is the instance's (whole on every rank, as for the general matrices of
//                        gm_snark_new_time_sharded): the lookups are local, with this rank's block of the index vectors
//   prefix products      accumulated_product (entryproduct/time_prover.rs:34-45) is a suffix scan: each rank scans its block from the
//                        product of the blocks above -- 9 x 32 bytes all-gathered, once
//   rotations / plookup  right_rotation and plookup_set read element i - 1: a 32-byte halo per vector from the rank below
//   sumchecks            64 bytes per prover and round (the third one: 13 provers in one all-gather), tails gathered once
//   commitments          partial G1 points, 144 bytes each, one all-gather per batch_commit
//   openings             n / g pairs per rank: re-blocked level sums, the carry between blocks interpolated from one all-gather of
//                        evaluations (as gm_snark_new_time_sharded)
// The proof is byte-identical to gm_psnark_new_time's on every rank (tests/test_gpu_dist_native.py).
#include <algorithm>
#include <utility>

#include "ctx.hpp"
#include "prover_common.hpp"

namespace {

using namespace gmprover;

size_t ceil_log2(size_t n) {
  size_t b = 0;
  while (((size_t)1 << b) < n) b++;
  return b;
}
size_t ceil_shift(size_t n, size_t k) { return k >= 63 ? (n ? 1 : 0) : (n + (((size_t)1 << k) - 1)) >> k; }

const uint64_t* identity_point() {
  static uint64_t id[18];
  static bool init = (gm_g1_sum(nullptr, 0, id), true);
  (void)init;
  return id;
}

// levels 0 .. jmax of a folding tree stay block-sharded: blocks of M >> j, EVEN (pairs fold inside a rank) and >= 2^tail_log
size_t sharded_levels(size_t M, size_t tail_log) {
  size_t j = 0;
  while (((M >> j) % 2 == 0) && ((M >> (j + 1)) % 2 == 0) && ((M >> (j + 1)) >= ((size_t)1 << tail_log))) j++;
  return j;
}

// the polynomial of degree < k through (xs[i], ys[i]), k <= 3: coefficients c[0 .. k)
void interp(const Fr* xs, const Fr* ys, size_t k, Fr* c) {
  for (size_t i = 0; i < k; i++) c[i] = Fr::zero();
  for (size_t i = 0; i < k; i++) {
    // numerator prod_{j != i} (x - xs[j]) as coefficients
    Fr num[3] = {Fr::one(), Fr::zero(), Fr::zero()};
    size_t deg = 0;
    Fr den = Fr::one();
    for (size_t j = 0; j < k; j++) {
      if (j == i) continue;
      for (size_t t = deg + 1; t-- > 0;) {
        num[t + 1] = num[t + 1] + num[t];
        num[t] = num[t] * xs[j].neg();
      }
      deg++;
      den = den * (xs[i] - xs[j]);
    }
    const Fr s = ys[i] * den.inv();
    for (size_t t = 0; t <= deg; t++) c[t] = c[t] + s * num[t];
  }
}

struct Sh {  // this rank's view of the block layout
  size_t r = 0, g = 1, M = 0, jmax = 0;
  size_t B(size_t s) const { return M >> s; }
  size_t lo(size_t s) const { return r * (M >> s); }
  size_t cnt(size_t len, size_t s) const { return len > lo(s) ? std::min(B(s), len - lo(s)) : 0; }
  // the level of a family whose longest member has `len` elements: the finest blocks, still foldable, whose g copies hold it
  size_t level(size_t len) const {
    size_t s = 0;
    while (s < jmax && g * (M >> (s + 1)) >= len) s++;
    return s;
  }
};
struct BV {  // a block-sharded vector: this rank's block (0 / length 0: nothing of it here), its whole length, its level
  uint64_t h = 0;
  size_t len = 0, s = 0;
};

// a device vector of n elements that also exists when n = 0 (an empty block)
int alloc_len(Vecs& V, size_t n, uint64_t* out, size_t room = 0) {
  RC(V.alloc(std::max<size_t>(n + room, 1), out));
  return gm_fr_vec_set_len(*out, n);
}
int alloc_zero(Vecs& V, size_t cap, size_t len, uint64_t* out) {
  RC(V.alloc(std::max<size_t>(cap, 1), out));
  uint64_t zero[4] = {0, 0, 0, 0};
  RC(gm_fr_vec_fill(*out, zero));
  return gm_fr_vec_set_len(*out, len);
}


int blk_len(uint64_t v, size_t* n) {
  *n = 0;
  return v ? vec_len(v, n) : GM_OK;
}

// values p(x) = sum_rr x^(rr * B(s)) P_rr(x) of block-sharded polynomials at npts points: local block evaluations, one all-gather
int eval_blocks(const Sh& lay, const std::vector<BV>& blocks, const uint64_t* pts, size_t npts, std::vector<Fr>& vals) {
  const size_t k = blocks.size();
  vals.assign(k * npts, Fr::zero());
  if (k == 0) return GM_OK;
  std::vector<uint64_t> local(4 * k * npts, 0), allr(4 * k * npts * lay.g, 0);
  {
    std::vector<uint64_t> live, res;
    std::vector<size_t> at;
    for (size_t i = 0; i < k; i++) {
      size_t len = 0;
      RC(blk_len(blocks[i].h, &len));
      if (len) {
        live.push_back(blocks[i].h);
        at.push_back(i);
      }
    }
    if (!live.empty()) {
      res.resize(4 * live.size() * npts);
      RC(gm_fr_eval_le_batch(live.data(), live.size(), pts, npts, res.data()));
      for (size_t t = 0; t < live.size(); t++) memcpy(local.data() + 4 * at[t] * npts, res.data() + 4 * t * npts, 32 * npts);
    }
  }
  RC(gm_dist_allgather_host(local.data(), 32 * k * npts, allr.data()));
  for (size_t i = 0; i < k; i++)
    for (size_t q = 0; q < npts; q++) {
      const Fr x = Fr::from_limbs(pts + 4 * q);
      const Fr step = fr_pow(x, lay.B(blocks[i].s));
      Fr acc = Fr::zero(), xp = Fr::one();
      for (size_t rr = 0; rr < lay.g; rr++) {
        acc = acc + xp * Fr::from_limbs(allr.data() + 4 * ((rr * k + i) * npts + q));
        xp = xp * step;
      }
      vals[i * npts + q] = acc;
    }
  return GM_OK;
}

struct Key {
  uint64_t h = 0;
  const size_t* offsets = nullptr;
  const size_t* counts = nullptr;
  size_t segments = 0;
};

// un-normalised MSMs of vecs[i] against the key slice of level levels[i], one pipelined batch (empty blocks: the identity)
int key_commit(const Key& K, const std::vector<size_t>& levels, const std::vector<uint64_t>& vecs, uint64_t* out) {
  std::vector<size_t> offs, ns, at;
  std::vector<uint64_t> live;
  for (size_t i = 0; i < vecs.size(); i++) {
    memcpy(out + 18 * i, identity_point(), 144);
    size_t len = 0;
    RC(blk_len(vecs[i], &len));
    if (levels[i] >= K.segments) return GM_EINVAL;
    len = std::min(len, K.counts[levels[i]]);
    if (!len) continue;
    offs.push_back(K.offsets[levels[i]]);
    ns.push_back(len);
    live.push_back(vecs[i]);
    at.push_back(i);
  }
  if (live.empty()) return GM_OK;
  std::vector<uint64_t> parts(18 * live.size());
  RC(gm_g1_msm_v_batch_at(K.h, offs.data(), 0, live.data(), ns.data(), live.size(), 1, parts.data()));
  for (size_t t = 0; t < live.size(); t++) memcpy(out + 18 * at[t], parts.data() + 18 * t, 144);
  return GM_OK;
}
// all-gather k partial points, add per column, normalise
int gather_sum(const Sh& lay, const uint64_t* parts, size_t k, uint64_t* out) {
  if (k == 0) return GM_OK;
  std::vector<uint64_t> all(18 * k * lay.g), col(18 * lay.g);
  RC(gm_dist_allgather_host_class(parts, 144 * k, all.data(), GM_DIST_CLASS_G1));
  for (size_t j = 0; j < k; j++) {
    for (size_t rr = 0; rr < lay.g; rr++) memcpy(col.data() + 18 * rr, all.data() + 18 * (rr * k + j), 144);
    RC(gm_g1_sum(col.data(), lay.g, out + 18 * j));
  }
  return GM_OK;
}
// ck.batch_commit of block-sharded vectors: a vector of level s against the key slice of level s
int commit_blocks(const Sh& lay, const Key& K, const std::vector<BV>& vecs, uint64_t* out) {
  std::vector<uint64_t> parts(18 * std::max<size_t>(vecs.size(), 1)), hs;
  std::vector<size_t> levels;
  for (const BV& v : vecs) {
    hs.push_back(v.h);
    levels.push_back(v.s);
  }
  RC(key_commit(K, levels, hs, parts.data()));
  return gather_sum(lay, parts.data(), vecs.size(), out);
}

// sum_i c_i v_i of block-sharded vectors of DIFFERENT levels, as a vector of level t = the coarsest level among them (blocks of B(t),
// capacity B(t) + room, zero beyond the data, length B(t)): the members of level t are combined in place; those of every finer level u > t
// are combined per level into one vector of nominal length B(u) and RE-BLOCKED to blocks of B(t) -- one grouped send / recv for all levels,
// every element crosses one link once -- then added.  Collective: every rank calls it with the same levels.
struct Item {
  BV v;
  Fr c;
};
int combine_levels(const Sh& lay, Vecs& V, const std::vector<Item>& items, size_t room, BV* out) {
  if (items.empty()) return GM_EINVAL;
  size_t t = ~(size_t)0, len = 0;
  for (const Item& it : items) {
    t = std::min(t, it.v.s);
    len = std::max(len, it.v.len);
  }
  std::vector<uint64_t> pieces, coeffs, temps;
  auto add = [&](uint64_t h, const Fr& c) {
    pieces.push_back(h);
    coeffs.resize(coeffs.size() + 4);
    c.to_limbs(coeffs.data() + coeffs.size() - 4);
  };
  std::vector<size_t> finer;  // the levels above t that occur (the same list on every rank: levels are whole-vector facts)
  for (const Item& it : items)
    if (it.v.s != t && std::find(finer.begin(), finer.end(), it.v.s) == finer.end()) finer.push_back(it.v.s);
  std::sort(finer.begin(), finer.end());
  for (const Item& it : items) {
    size_t n = 0;
    RC(blk_len(it.v.h, &n));
    if (it.v.s == t && n) add(it.v.h, it.c);
  }
  if (!finer.empty()) {
    std::vector<uint64_t> sums(finer.size()), outs(finer.size());
    for (size_t i = 0; i < finer.size(); i++) {
      const size_t u = finer[i];
      std::vector<uint64_t> live, cf;
      for (const Item& it : items) {
        size_t n = 0;
        RC(blk_len(it.v.h, &n));
        if (it.v.s != u || !n) continue;
        live.push_back(it.v.h);
        cf.resize(cf.size() + 4);
        it.c.to_limbs(cf.data() + cf.size() - 4);
      }
      RC(alloc_zero(V, lay.B(u), 0, &sums[i]));
      if (!live.empty()) RC(gm_fr_lincomb(live.data(), cf.data(), live.size(), sums[i]));
      RC(gm_fr_vec_set_len(sums[i], lay.B(u)));
      RC(alloc_len(V, lay.B(t), &outs[i]));
      temps.push_back(sums[i]);
      temps.push_back(outs[i]);
    }
    RC(gm_dist_reblock_vecs(sums.data(), sums.size(), lay.B(t), outs.data()));
    for (uint64_t o : outs) {
      size_t n = 0;
      RC(vec_len(o, &n));
      if (n) add(o, Fr::one());
    }
  }
  uint64_t res;
  RC(alloc_zero(V, lay.B(t) + room, 0, &res));
  if (!pieces.empty()) RC(gm_fr_lincomb(pieces.data(), coeffs.data(), pieces.size(), res));
  RC(gm_fr_vec_set_len(res, lay.B(t)));
  for (uint64_t h : temps) V.release(h);
  out->h = res;
  out->len = len;
  out->s = t;
  return GM_OK;
}

// ---- Sumcheck::prove / prove_batch over blocks ---------------------------------------------------------------------------
// (src/subprotocols/sumcheck/proof.rs:36-122) k provers whose vectors are block-sharded, each at its own level.  While a prover's blocks
// hold more than `tail` elements and stay pair-aligned, its rounds are shard-local: the partial messages of ALL sharded provers -- 64 bytes
// each -- are all-gathered in ONE call per round and added mod r.  A prover whose blocks get short is gathered once (one all-gather for all
// the provers that switch in that round) and finished replicated on the whole (short) vectors.  batch = false: Sumcheck::prove of ONE
// prover (labels and round count differ).
struct ShProver {
  uint64_t f = 0, g = 0;  // this rank's blocks (length 0: nothing of the vectors falls into the block)
  const uint64_t* twist = nullptr;
  size_t len = 0;  // of the whole vectors
  size_t s = 0;    // their level
};
struct ProverSet {
  std::vector<uint64_t> h;
  ~ProverSet() {
    for (uint64_t p : h)
      if (p) (void)gm_sc_free(p);
  }
};

int sumcheck_blocks(const Sh& lay, uint64_t transcript, bool batch, const std::vector<ShProver>& P, size_t tail, uint64_t* messages, uint64_t* challenges,
                    size_t cap_rounds, uint64_t* final_foldings, size_t* rounds_out) {
  const size_t k = P.size();
  if (k == 0 || (!batch && k != 1)) return GM_EINVAL;
  std::vector<size_t> tot(k);
  size_t max_tot = 0;
  for (size_t j = 0; j < k; j++) {
    tot[j] = ceil_log2(P[j].len);  // time_prover.rs:35-38
    max_tot = std::max(max_tot, tot[j]);
  }
  const size_t rounds = batch ? max_tot + 1 : max_tot;  // "+1 to get the final foldings" (proof.rs:74)
  if (rounds > cap_rounds) return GM_EINVAL;
  std::vector<Fr> coeff(k, Fr::one());
  if (batch)
    for (size_t j = 0; j < k; j++) {
      uint64_t c[4];
      RC(gm_transcript_challenge_fr(transcript, L("batch-sumcheck"), 14, c));
      coeff[j] = Fr::from_limbs(c);
    }
  ProverSet S;
  S.h.assign(k, 0);
  std::vector<size_t> Mc(k);       // current block size of a sharded prover
  std::vector<char> rep(k, 0);     // finished its sharded phase: S.h[j] is a replicated prover over the whole vectors
  for (size_t j = 0; j < k; j++) {
    size_t nf = 0, ng = 0;
    RC(blk_len(P[j].f, &nf));
    RC(blk_len(P[j].g, &ng));
    Mc[j] = lay.B(P[j].s);
    if (lay.g == 1) {
      if (!nf || !ng) return GM_EINVAL;  // "sumcheck: empty vectors"
      RC(gm_sc_new_borrow(P[j].f, P[j].g, P[j].twist, &S.h[j]));
      rep[j] = 1;
      continue;
    }
    if (nf != lay.cnt(P[j].len, P[j].s) || ng != nf) return GM_EINVAL;  // the blocks of a sharded prover tile its vectors
    if (nf) {
      RC(gm_sc_new_borrow(P[j].f, P[j].g, P[j].twist, &S.h[j]));
      RC(gm_sc_set_shard_rounds(S.h[j], lay.lo(P[j].s) / 2, tot[j]));
    }
  }
  size_t rd = 0;  // messages sent so far = folds applied once the pending challenge is in
  const uint64_t* vm = nullptr;
  std::vector<Fr> final_product(k);
  std::vector<char> finished(k, 0);
  for (;;) {
    // provers that leave their sharded phase now: the pending fold is applied shard-locally, then the blocks are gathered -- the
    // replicated prover starts exactly at a message boundary (and must not fold again in this round)
    std::vector<size_t> sw;
    for (size_t j = 0; j < k; j++)
      if (!rep[j] && !(Mc[j] % 4 == 0 && Mc[j] > tail && rd < tot[j])) sw.push_back(j);
    std::vector<char> folded(k, 0);
    if (!sw.empty()) {
      size_t slots = 0;
      std::vector<size_t> off(sw.size()), bs(sw.size());
      for (size_t i = 0; i < sw.size(); i++) {
        const size_t j = sw[i];
        if (vm) {
          if (S.h[j]) RC(gm_sc_fold(S.h[j], vm));
          Mc[j] /= 2;
          folded[j] = 1;
        }
        bs[i] = Mc[j];
        off[i] = slots;
        slots += 4 + 8 * bs[i];  // [count | f | g], limbs
      }
      std::vector<uint64_t> mine(slots, 0), all(slots * lay.g);
      for (size_t i = 0; i < sw.size(); i++) {
        const size_t j = sw[i];
        if (!S.h[j]) continue;
        size_t nf = 0, ng = 0;
        RC(gm_sc_lens(S.h[j], &nf, &ng, nullptr));
        if (nf != ng || nf > bs[i]) return GM_ESTATE;
        mine[off[i]] = nf;
        RC(gm_sc_download(S.h[j], mine.data() + off[i] + 4, mine.data() + off[i] + 4 + 4 * bs[i]));
      }
      RC(gm_dist_allgather_host(mine.data(), 8 * slots, all.data()));
      const size_t folds = rd;  // (with the pending challenge applied)
      for (size_t i = 0; i < sw.size(); i++) {
        const size_t j = sw[i];
        std::vector<uint64_t> fs, gs;
        for (size_t rr = 0; rr < lay.g; rr++) {
          const uint64_t* sl = all.data() + slots * rr + off[i];
          const size_t c = (size_t)sl[0];
          if (c > bs[i]) return GM_ESTATE;
          fs.insert(fs.end(), sl + 4, sl + 4 + 4 * c);
          gs.insert(gs.end(), sl + 4 + 4 * bs[i], sl + 4 + 4 * bs[i] + 4 * c);
        }
        const size_t n = fs.size() / 4;
        if (n != ceil_shift(P[j].len, folds)) return GM_ESTATE;  // the blocks tile the folded vectors
        Fr tw = Fr::from_limbs(P[j].twist);
        for (size_t t = 0; t < folds; t++) tw = tw.sqr();
        uint64_t twl[4];
        tw.to_limbs(twl);
        if (S.h[j]) (void)gm_sc_free(S.h[j]);
        S.h[j] = 0;
        RC(gm_sc_new(fs.data(), n, gs.data(), n, twl, &S.h[j]));
        rep[j] = 1;
      }
    }
    if (batch && rd == rounds) break;
    bool any_sharded = false;
    for (size_t j = 0; j < k; j++) any_sharded = any_sharded || !rep[j];
    std::vector<char> has(k, 0);
    for (int pass = 0; pass < 2; pass++) {
      // one launch for the provers that fold with the pending challenge, one for those whose fold went into their gathering
      std::vector<uint64_t> hs;
      std::vector<size_t> at;
      for (size_t j = 0; j < k; j++)
        if (S.h[j] && !finished[j] && (folded[j] != 0) == (pass == 1)) {
          hs.push_back(S.h[j]);
          at.push_back(j);
        }
      if (hs.empty()) continue;
      std::vector<int> flags(hs.size(), 0);
      RC(gm_sc_round_begin_many(hs.data(), hs.size(), pass == 1 ? nullptr : vm, flags.data()));
      for (size_t t = 0; t < hs.size(); t++) has[at[t]] = (char)flags[t];
    }
    std::vector<uint64_t> part(8 * k, 0);
    bool any = any_sharded;  // (a sharded prover has a message in this round -- rd < tot -- on some rank)
    for (size_t j = 0; j < k; j++) {
      if (S.h[j] && !finished[j] && has[j]) {
        RC(gm_sc_round_end(S.h[j], part.data() + 8 * j, part.data() + 8 * j + 4));
        any = true;
      } else if (rep[j]) {
        if (!batch) continue;  // Sumcheck::prove: no message means the protocol is over
        if (!finished[j]) {
          uint64_t f0[4], g0[4];
          int hf = 0;
          RC(gm_sc_final(S.h[j], f0, g0, &hf));
          if (!hf) return GM_ESTATE;  // "If next_message is None, we expect final foldings to be available"
          final_product[j] = Fr::from_limbs(f0) * Fr::from_limbs(g0);
          finished[j] = 1;
        }
        final_product[j].to_limbs(part.data() + 8 * j);
      }
    }
    if (any_sharded) {
      // ONE all-gather per round for all the provers still in their sharded phase (a replicated prover's slot carries zeros)
      std::vector<uint64_t> mine(8 * k, 0), all(8 * k * lay.g);
      for (size_t j = 0; j < k; j++)
        if (!rep[j]) memcpy(mine.data() + 8 * j, part.data() + 8 * j, 64);
      RC(gm_dist_allgather_host(mine.data(), 64 * k, all.data()));
      for (size_t j = 0; j < k; j++) {
        if (rep[j]) continue;
        Fr sa = Fr::zero(), sb = Fr::zero();
        for (size_t rr = 0; rr < lay.g; rr++) {
          sa = sa + Fr::from_limbs(all.data() + 8 * (rr * k + j));
          sb = sb + Fr::from_limbs(all.data() + 8 * (rr * k + j) + 4);
        }
        sa.to_limbs(part.data() + 8 * j);
        sb.to_limbs(part.data() + 8 * j + 4);
      }
    }
    if (vm)
      for (size_t j = 0; j < k; j++)
        if (!rep[j]) Mc[j] /= 2;
    if (!batch && !any) break;
    if (rd >= cap_rounds) return GM_EINVAL;
    Fr ma = Fr::zero(), mb = Fr::zero();
    for (size_t j = 0; j < k; j++) {
      ma = ma + Fr::from_limbs(part.data() + 8 * j) * coeff[j];
      mb = mb + Fr::from_limbs(part.data() + 8 * j + 4) * coeff[j];
    }
    ma.to_limbs(messages + 8 * rd);
    mb.to_limbs(messages + 8 * rd + 4);
    RC(gm_transcript_append_fr(transcript, L("evaluations"), 11, messages + 8 * rd, 2));
    RC(gm_transcript_challenge_fr(transcript, L("challenge"), 9, challenges + 4 * rd));
    vm = challenges + 4 * rd;
    rd++;
  }
  for (size_t j = 0; j < k; j++) {
    int has = 0;
    RC(gm_sc_final(S.h[j], final_foldings + 8 * j, final_foldings + 8 * j + 4, &has));
    if (!has) return GM_ESTATE;
    if (batch) {
      RC(gm_transcript_append_fr(transcript, L("final-folding-lhs"), 17, final_foldings + 8 * j, 1));
      RC(gm_transcript_append_fr(transcript, L("final-folding-rhs"), 17, final_foldings + 8 * j + 4, 1));
    } else {
      RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings + 8 * j, 1));
      RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings + 8 * j + 4, 1));
    }
  }
  *rounds_out = rd;
  return GM_OK;
}

// ---- batch_open_multi_points over blocks ------------------------------------------------------------------------------------
// (src/kzg/time.rs:149-159) commit((sum_i eta_i p_i) div Z), Z = prod (x - pts[q]), npts <= 3.  F = sum eta_i p_i is ONE polynomial, built
// at the coarsest level t among its pieces (combine_levels; `small`: replicated short polynomials, every rank takes its range): rank r takes ITS
// coefficient range [r B(t), (r + 1) B(t)) of F and commits the quotient of that block against the key slice of level t -- |F| / g pairs per rank.
// The quotient of block r needs the carry from the blocks above: the polynomial c of degree < npts that agrees with
// S_r(x) = sum_{r' > r} x^((r' - r - 1) B) F_r'(x) at the roots of Z (one all-gather of npts evaluations per rank), and leaves a remainder
// that agrees with G = F_r + x^B c at the roots: q_r = (G - rem) / Z exactly, F div Z = sum_r x^(r B) q_r.
struct Piece {
  uint64_t v;
  Fr eta;
};
int open_blocks(const Sh& lay, const Key& K, Vecs& V, const std::vector<Item>& items, const std::vector<Piece>& small, const uint64_t* pts, size_t npts,
                uint64_t out[18]) {
  if (npts < 1 || npts > 3) return GM_EINVAL;
  BV Fv;
  RC(combine_levels(lay, V, items, npts, &Fv));
  const size_t t = Fv.s, B = lay.B(t), r = lay.r, g = lay.g;
  const uint64_t F = Fv.h;
  size_t flen = Fv.len;
  std::vector<uint64_t> owned{F};
  {
    std::vector<uint64_t> pieces, piece_eta;
    for (const Piece& p : small) {
      size_t len = 0;
      RC(vec_len(p.v, &len));
      flen = std::max(flen, len);
      if (r * B >= len) continue;
      const size_t cnt = std::min(B, len - r * B);
      uint64_t part = p.v;
      if (!(r == 0 && len <= B)) {
        RC(alloc_len(V, cnt, &part));
        owned.push_back(part);
        RC(gm_fr_stride(p.v, r * B, 1, cnt, part));
      }
      pieces.push_back(part);
      piece_eta.resize(piece_eta.size() + 4);
      p.eta.to_limbs(piece_eta.data() + piece_eta.size() - 4);
    }
    if (!pieces.empty()) {
      // F += the ranges of the replicated polynomials (a second pass: F itself is one of the terms)
      uint64_t F2;
      RC(alloc_zero(V, B + npts, 0, &F2));
      pieces.push_back(F);
      piece_eta.resize(piece_eta.size() + 4);
      Fr::one().to_limbs(piece_eta.data() + piece_eta.size() - 4);
      RC(gm_fr_lincomb(pieces.data(), piece_eta.data(), pieces.size(), F2));
      RC(gm_fr_vec_set_len(F2, B));
      // (swap: F2 is the polynomial from here on)
      owned[0] = F2;
      V.release(F);
    }
  }
  const uint64_t Fh = owned[0];
  const size_t lf = lay.cnt(flen, t);
  const bool above = (r + 1) * B < flen;  // something of F lives on a higher rank
  // F_r at the roots, all ranks
  std::vector<uint64_t> mine_ev(4 * npts, 0), all_ev(4 * npts * g);
  if (lf) RC(gm_fr_eval_le(Fh, pts, npts, mine_ev.data()));
  RC(gm_dist_allgather_host(mine_ev.data(), 32 * npts, all_ev.data()));
  uint64_t mine[18];
  memcpy(mine, identity_point(), 144);
  if (lf) {
    Fr xs[3], c[3] = {Fr::zero(), Fr::zero(), Fr::zero()}, gv[3], rem[3];
    for (size_t q = 0; q < npts; q++) xs[q] = Fr::from_limbs(pts + 4 * q);
    const size_t len_f = above ? B + npts : std::max(lf, npts);
    RC(gm_fr_vec_set_len(Fh, len_f));  // (the tail beyond the combination is the zero fill)
    std::vector<size_t> pos;
    std::vector<uint64_t> val;
    auto seam = [&](size_t at, const Fr& v) {
      pos.push_back(at);
      val.resize(val.size() + 4);
      v.to_limbs(val.data() + val.size() - 4);
    };
    if (above) {
      Fr ys[3];
      for (size_t q = 0; q < npts; q++) {
        const Fr step = fr_pow(xs[q], B);
        Fr acc = Fr::zero(), xp = Fr::one();
        for (size_t rr = r + 1; rr < g; rr++) {
          acc = acc + xp * Fr::from_limbs(all_ev.data() + 4 * npts * rr + 4 * q);
          xp = xp * step;
        }
        ys[q] = acc;
      }
      interp(xs, ys, npts, c);
      for (size_t q = 0; q < npts; q++) seam(B + q, c[q]);
    }
    for (size_t q = 0; q < npts; q++) {
      Fr cx = Fr::zero();
      for (size_t tt = npts; tt-- > 0;) cx = cx * xs[q] + c[tt];
      gv[q] = Fr::from_limbs(mine_ev.data() + 4 * q) + fr_pow(xs[q], B) * cx;
    }
    interp(xs, gv, npts, rem);
    for (size_t q = 0; q < npts; q++) seam(q, rem[q].neg());
    RC(gm_fr_add_at(Fh, pos.data(), val.data(), pos.size()));
    if (len_f > npts) {
      uint64_t quot, remz[12];
      RC(alloc_len(V, len_f - 1, &quot));  // (the division peels one linear factor at a time: room for the first quotient)
      owned.push_back(quot);
      RC(gm_fr_div_vanishing(Fh, pts, npts, quot, remz));
      for (size_t l = 0; l < 4 * npts; l++)
        if (remz[l] != 0) return GM_ESTATE;  // the block of the opening is not divisible by Z
      size_t lq = 0;
      RC(vec_len(quot, &lq));
      lq = std::min(lq, K.counts[t]);
      if (lq) {
        const size_t off0 = K.offsets[t];
        RC(gm_g1_msm_v_batch_at(K.h, &off0, 0, &quot, &lq, 1, 1, mine));
      }
    }
  }
  for (uint64_t v : owned) V.release(v);
  return gather_sum(lay, mine, 1, out);
}

int shard_layout(const gm_psnark_shard* S, Sh* lay, Key* K) {
  int rank = 0, world = 1;
  RC(gm_dist_info(&rank, &world, nullptr));
  lay->r = (size_t)rank;
  lay->g = (size_t)world;
  lay->M = S->block;
  if (lay->M < 4 || lay->M % 4 != 0 || S->tail_log < 2 || S->tail_log > 40) return GM_EINVAL;
  lay->jmax = sharded_levels(lay->M, S->tail_log);
  if (S->key_segments != lay->jmax + 2 || !S->key_offsets || !S->key_counts) return GM_EINVAL;
  K->h = S->key;
  K->offsets = S->key_offsets;
  K->counts = S->key_counts;
  K->segments = S->key_segments;
  return GM_OK;
}

// the levels of the families the caller hands in blocks of (the same function of the whole lengths on both sides of the ABI)
struct Families {
  size_t rows, w, nnz, ext_row, ext_col;
};
Families family_levels(const Sh& lay, const gm_psnark_shard* S) {
  Families f;
  f.rows = lay.level(S->num_constraints);
  f.w = lay.level(S->w_len);
  f.nnz = lay.level(S->nnz + 1);                   // the joint-support vectors, their lookups, products (nnz + 1) and rotations
  f.ext_row = lay.level(S->ext_fre_row_len + 2);   // a sorted vector, its lookup vector (+ 1), its products and rotation (+ 2)
  f.ext_col = lay.level(S->ext_fre_col_len + 2);
  return f;
}

}  // namespace

extern "C" {

// the block size of a proof over `longest` elements on `world` ranks: ceil(longest / world) rounded up to a multiple of the largest
// power of two that keeps the rounding under 1/64 of a block (the blocks must halve for as many levels as stay sharded)
size_t gm_psnark_shard_block(size_t longest, int world) {
  const size_t g = world > 0 ? (size_t)world : 1, per = (std::max<size_t>(longest, 1) + g - 1) / g;
  size_t e = 2;
  while (((size_t)1 << (e + 1)) * 64 <= per) e++;
  const size_t unit = (size_t)1 << e;
  return (per + unit - 1) / unit * unit;
}

// the level of a family of vectors whose longest member has `len` elements: blocks of block >> level (see the head of this file)
size_t gm_psnark_shard_level(size_t len, size_t block, size_t tail_log, int world) {
  Sh lay;
  lay.g = world > 0 ? (size_t)world : 1;
  lay.M = block;
  lay.jmax = sharded_levels(block, tail_log);
  return lay.level(len);
}

// this rank's slices of a key of n_key powers for a proof with block size `block`: powers [r B_j, (r + 1) B_j) that exist, B_j =
// block >> j, for the sharded levels j = 0 .. jmax, then the prefix every rank commits the gathered levels against -- ONE handle
int gm_psnark_shard_key_new(const uint64_t base_affine[12], const uint64_t tau[4], size_t n_key, size_t block, size_t tail_log, uint64_t* key,
                            size_t offsets[64], size_t counts[64], size_t* segments) {
  GM_CTX();
  GM_CHECK(base_affine && tau && key && offsets && counts && segments, GM_EINVAL, "psnark_shard_key_new: null pointer");
  int rank = 0, world = 1;
  RC(gm_dist_info(&rank, &world, nullptr));
  GM_CHECK(block >= 4 && block % 4 == 0 && tail_log >= 2 && tail_log <= 40 && n_key >= 1, GM_EINVAL, "psnark_shard_key_new: block %zu, tail 2^%zu, %zu powers", block,
           tail_log, n_key);
  const size_t jmax = sharded_levels(block, tail_log);
  GM_CHECK(jmax + 2 <= 64, GM_EINVAL, "psnark_shard_key_new: %zu levels", jmax);
  size_t starts[64], at = 0;
  for (size_t j = 0; j <= jmax; j++) {
    const size_t b = block >> j, lo = (size_t)rank * b;
    starts[j] = std::min(lo, n_key);
    counts[j] = n_key > lo ? std::min(b, n_key - lo) : 0;
  }
  starts[jmax + 1] = 0;
  counts[jmax + 1] = std::min(n_key, (size_t)world * (block >> (jmax + 1)));
  // (a segment may be empty on the top ranks; the generator wants at least one power per segment)
  size_t gen_counts[64];
  for (size_t j = 0; j <= jmax + 1; j++) {
    gen_counts[j] = std::max<size_t>(counts[j], 1);
    if (counts[j] == 0) starts[j] = 0;
    offsets[j] = at;
    at += gen_counts[j];
  }
  *segments = jmax + 2;
  return gm_g1_srs_register_segments(base_affine, tau, starts, gen_counts, jmax + 2, key);
}

// psnark::Proof::index (src/psnark/time_prover.rs:49-64) over blocks: commitments to row, col, val_a, val_b, val_c
int gm_psnark_index_sharded(const gm_psnark_shard* S, uint64_t* out_jac) {
  GM_CTX();
  GM_CHECK(S && out_jac, GM_EINVAL, "psnark_index_sharded: null pointer");
  Sh lay;
  Key K;
  RC(shard_layout(S, &lay, &K));
  const size_t s = family_levels(lay, S).nnz;
  return commit_blocks(lay, K, {{S->row, S->nnz, s}, {S->col, S->nnz, s}, {S->val_a, S->nnz, s}, {S->val_b, S->nnz, s}, {S->val_c, S->nnz, s}}, out_jac);
}

static int psnark_new_time_sharded_impl(const gm_psnark_shard* S, int g1_encoding, size_t cap_rounds, gm_psnark_proof* P);
// (a failure on this rank -- outside a collective as well: a bad input, an allocation -- tells the peers instead of leaving them in their next all-gather)
int gm_psnark_new_time_sharded(const gm_psnark_shard* S, int g1_encoding, size_t cap_rounds, gm_psnark_proof* P) {
  const int rc = psnark_new_time_sharded_impl(S, g1_encoding, cap_rounds, P);
  if (rc) (void)gm_dist_abort();
  return rc;
}
static int psnark_new_time_sharded_impl(const gm_psnark_shard* S, int g1_encoding, size_t cap_rounds, gm_psnark_proof* P) {
  GM_CTX();
  GM_CHECK(S && P && S->index_commitments && P->messages[0] && P->messages[1] && P->messages[2] && P->fold_commitments && P->fold_evaluations, GM_EINVAL,
           "psnark_new_time_sharded: null pointer");
  const auto t_all = Clock::now();
  Sh lay;
  Key K;
  RC(shard_layout(S, &lay, &K));
  const size_t M = lay.M, r = lay.r, g = lay.g, jmax = lay.jmax;
  const size_t tail = (size_t)1 << S->tail_log;
  const size_t nrows = S->num_constraints, nz = S->num_variables, nnz = S->nnz;
  const Families fam = family_levels(lay, S);
  Vecs V;
  size_t zlen = 0;
  RC(vec_len(S->z, &zlen));
  GM_CHECK(zlen == nz, GM_EINVAL, "psnark_new_time_sharded: z has %zu elements, the instance %zu variables (z is whole on every rank)", zlen, nz);
  GM_CHECK(S->key_len >= nnz && S->key_len >= S->ext_fre_row_len && S->key_len >= S->ext_fre_col_len, GM_EINVAL,
           "psnark_new_time_sharded: a key of %zu powers is shorter than the index vectors", S->key_len);
  // every vector of the proof fits g blocks
  {
    const size_t longest = std::max({S->ext_fre_row_len + 2, S->ext_fre_col_len + 2, nz + 2, nrows + 2, nnz + 1});
    GM_CHECK(longest <= g * M, GM_EINVAL, "psnark_new_time_sharded: %zu blocks of %zu elements do not hold the longest vector (%zu)", g, M, longest);
  }
  // room for this rank's share of the proof, or GM_ENOMEM with the numbers, before the first allocation
  RC(gm_psnark_shard_footprint(S->key, nrows, nz, nnz, M, (int)g, 1, nullptr));
  uint64_t one[4];
  Fr::one().to_limbs(one);
  // the blocks the caller hands in tile their vectors at the levels of their families
  {
    const uint64_t vs[6] = {S->w_block, S->row, S->col, S->val_a, S->val_b, S->val_c};
    const size_t lens[6] = {S->w_len, nnz, nnz, nnz, nnz, nnz}, lv[6] = {fam.w, fam.nnz, fam.nnz, fam.nnz, fam.nnz, fam.nnz};
    for (int k = 0; k < 6; k++) {
      size_t n = 0;
      RC(blk_len(vs[k], &n));
      GM_CHECK(n == lay.cnt(lens[k], lv[k]), GM_EINVAL, "psnark_new_time_sharded: input block %d holds %zu elements, the layout says %zu (level %zu, blocks of %zu)", k, n,
               lay.cnt(lens[k], lv[k]), lv[k], lay.B(lv[k]));
    }
  }
  const BV w{S->w_block, S->w_len, fam.w};

  // z_a, z_b, z_c (:74-76): row blocks, global columns
  BV z_abc[3];
  {
    const uint64_t mats[3] = {S->a, S->b, S->c};
    const size_t rows_blk = lay.cnt(nrows, fam.rows);
    for (int k = 0; k < 3; k++) {
      z_abc[k] = BV{0, nrows, fam.rows};
      RC(alloc_len(V, rows_blk, &z_abc[k].h));
      if (!rows_blk) continue;
      size_t rows = 0, cols = 0;
      RC(gm_spm_shape(mats[k], &rows, &cols, nullptr));
      GM_CHECK(rows == rows_blk && cols == nz, GM_EINVAL, "psnark_new_time_sharded: matrix %d is %zu x %zu, expected the row block %zu x %zu", k, rows, cols, rows_blk, nz);
      RC(gm_spm_mul(mats[k], S->z, z_abc[k].h));
    }
  }
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));

  auto t0 = Clock::now();
  RC(commit_blocks(lay, K, {w}, P->witness_commitment));  // :79
  P->spans[0] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));  // :82-86
  RC(gm_transcript_append_message(T.h, L("ck"), 2, S->ck_g2_bytes, S->ck_g2_len));
  RC(gm_transcript_append_g1(T.h, L("instance"), 8, S->index_commitments, 5, 1));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  {
    std::vector<Fr> vals;
    RC(eval_blocks(lay, {z_abc[2]}, alpha, 1, vals));  // :88-89
    vals[0].to_limbs(P->zc_alpha);
  }
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1(4 * cap_rounds), ch2(4 * cap_rounds), ch3(4 * cap_rounds);
  RC(sumcheck_blocks(lay, T.h, false, {ShProver{z_abc[0].h, z_abc[1].h, alpha, nrows, fam.rows}}, tail, P->messages[0], ch1.data(), cap_rounds, P->final_foldings[0],
                     &P->rounds[0]));  // :92
  P->spans[1] = since(t0);
  for (int k = 0; k < 3; k++) V.release(z_abc[k].h);

  t0 = Clock::now();
  const size_t nt = (size_t)1 << P->rounds[0];
  // extend_frequency(compute_frequency(set_len, index)) has set_len + |index| entries (plookup/time_prover.rs:66-79)
  GM_CHECK(S->ext_fre_row_len == nt + nnz && S->ext_fre_col_len == nz + nnz, GM_EINVAL, "psnark_new_time_sharded: extended frequencies of %zu / %zu entries, expected %zu / %zu",
           S->ext_fre_row_len, S->ext_fre_col_len, nt + nnz, nz + nnz);
  // tensor(rho) and powers(alpha) (:95-97) are FUNCTIONS of the index: no rank ever builds them.  A lookup is one multiplication per element from
  // two half tables that stay in L2 (gm_fr_tensor_gather / gm_fr_powers_gather), a contiguous range comes from gm_fr_tensor_range /
  // gm_fr_powers_range; their product is only ever needed at the looked-up positions.  kind 0: tensor(rho), 1: powers(alpha), 2: z (the instance's, whole)
  const size_t k0 = P->rounds[0];
  auto lookup_fn = [&](int kind, uint64_t index, uint64_t out) -> int {
    if (kind == 2) return gm_fr_gather(S->z, index, out);
    if (k0 == 0) {  // one constraint: both vectors are [1]
      size_t n = 0;
      RC(vec_len(out, &n));
      return n ? gm_fr_vec_fill(out, one) : GM_OK;
    }
    return kind == 0 ? gm_fr_tensor_gather(ch1.data(), k0, index, out) : gm_fr_powers_gather(alpha, k0, index, out);
  };
  auto range_fn = [&](int kind, size_t start, size_t count, uint64_t out) -> int {  // elements [start, start + count) of the set of `kind`
    if (kind == 2) return count ? gm_fr_stride(S->z, start, 1, count, out) : gm_fr_vec_set_len(out, 0);
    if (k0 == 0) {
      RC(gm_fr_vec_set_len(out, count));
      return count ? gm_fr_vec_fill(out, one) : GM_OK;
    }
    return kind == 0 ? gm_fr_tensor_range(ch1.data(), k0, start, count, out) : gm_fr_powers_range(alpha, start, count, out);
  };
  P->spans[2] = since(t0);

  const size_t sn = fam.nnz, nnz_blk = lay.cnt(nnz, sn), lo_n = lay.lo(sn);
  BV ralpha_star{0, nnz, sn}, r_star{0, nnz, sn}, alpha_star{0, nnz, sn}, z_star{0, nnz, sn};  // :114-117, this rank's block of the index vectors
  RC(alloc_len(V, nnz_blk, &r_star.h));
  RC(alloc_len(V, nnz_blk, &alpha_star.h));
  RC(alloc_len(V, nnz_blk, &ralpha_star.h));
  RC(alloc_len(V, nnz_blk, &z_star.h));
  if (nnz_blk) {
    RC(gm_fr_vec_set_len(r_star.h, nnz_blk));
    RC(gm_fr_vec_set_len(alpha_star.h, nnz_blk));
    RC(lookup_fn(0, S->row_index, r_star.h));
    RC(lookup_fn(1, S->row_index, alpha_star.h));
    RC(gm_fr_hadamard(r_star.h, alpha_star.h, ralpha_star.h));
    RC(gm_fr_gather(S->z, S->col_index, z_star.h));
    size_t n1 = 0;
    RC(vec_len(r_star.h, &n1));
    GM_CHECK(n1 == nnz_blk, GM_EINVAL, "psnark_new_time_sharded: the block of the row index holds %zu entries, the layout says %zu", n1, nnz_blk);
  }
  const BV row{S->row, nnz, sn}, col{S->col, nnz, sn}, val_a{S->val_a, nnz, sn}, val_b{S->val_b, nnz, sn}, val_c{S->val_c, nnz, sn};

  t0 = Clock::now();
  {
    uint64_t four[4 * 18];
    RC(commit_blocks(lay, K, {ralpha_star, r_star, alpha_star, z_star}, four));  // :119-127
    memcpy(P->r_star_commitments, four, 3 * 144);
    memcpy(P->z_star_commitment, four + 54, 144);
  }
  P->spans[3] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("ra*"), 3, P->r_star_commitments[0], 1, 0));  // :129-132
  RC(gm_transcript_append_g1(T.h, L("rb*"), 3, P->r_star_commitments[1], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("rc*"), 3, P->r_star_commitments[2], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("z*"), 2, P->z_star_commitment, 1, 0));
  uint64_t eta3[12];  // 1, eta, eta^2   :134-135
  memcpy(eta3, one, 32);
  RC(gm_transcript_challenge_fr(T.h, L("chal"), 4, eta3 + 4));
  Fr::from_limbs(eta3 + 4).sqr().to_limbs(eta3 + 8);
  uint64_t r_star_val;
  RC(alloc_len(V, nnz_blk, &r_star_val));
  if (nnz_blk) {
    uint64_t h[3];
    const uint64_t lhs[3] = {ralpha_star.h, r_star.h, alpha_star.h}, rhs[3] = {S->val_a, S->val_b, S->val_c};
    for (int k = 0; k < 3; k++) {
      RC(V.alloc(nnz_blk, &h[k]));
      RC(gm_fr_hadamard(lhs[k], rhs[k], h[k]));
    }
    // (a zero-filled target: the combination of a block may be shorter than the block, the layout is not)
    uint64_t zero[4] = {0, 0, 0, 0};
    RC(gm_fr_vec_fill(r_star_val, zero));
    RC(gm_fr_lincomb(h, eta3, 3, r_star_val));  // :137-144
    RC(gm_fr_vec_set_len(r_star_val, nnz_blk));
    for (int k = 0; k < 3; k++) V.release(h[k]);
  }

  t0 = Clock::now();
  RC(sumcheck_blocks(lay, T.h, false, {ShProver{z_star.h, r_star_val, one, nnz, sn}}, tail, P->messages[1], ch2.data(), cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :147-152
  GM_CHECK(((size_t)1 << P->rounds[1]) >= nnz, GM_ESTATE, "psnark_new_time_sharded: %zu rounds for %zu entries", P->rounds[1], nnz);
  uint64_t second_challenges;  // &tensor(second challenges)[..num_non_zero], this rank's block
  RC(alloc_len(V, nnz_blk, &second_challenges));
  if (nnz_blk) RC(gm_fr_tensor_range(ch2.data(), P->rounds[1], lo_n, nnz_blk, second_challenges));
  V.release(r_star_val);
  P->spans[4] = since(t0);

  uint64_t zeta[4];
  RC(gm_transcript_challenge_fr(T.h, L("zeta"), 4, zeta));  // :157
  const bool hashed = !Fr::from_limbs(zeta).is_zero();

  t0 = Clock::now();
  // sorted_k = lookup(alg_hash(set_k), extended frequency) = set_k[e] + zeta e for e in this rank's block of the extended frequency (:160-173)
  const size_t ext_len[3] = {S->ext_fre_row_len, S->ext_fre_row_len, S->ext_fre_col_len}, ext_lv[3] = {fam.ext_row, fam.ext_row, fam.ext_col};
  const uint64_t ext_idx[3] = {S->ext_fre_row, S->ext_fre_row, S->ext_fre_col};
  const size_t set_len[3] = {nt, nt, nz};
  BV sorted[3];
  for (int k = 0; k < 3; k++) {
    const size_t n = lay.cnt(ext_len[k], ext_lv[k]);
    sorted[k] = BV{0, ext_len[k], ext_lv[k]};
    RC(alloc_len(V, n, &sorted[k].h));
    if (!n) continue;
    uint64_t tmp;
    RC(V.alloc(n, &tmp));
    RC(lookup_fn(k, ext_idx[k], tmp));
    size_t got = 0;
    RC(vec_len(tmp, &got));
    GM_CHECK(got == n, GM_EINVAL, "psnark_new_time_sharded: the block of extended frequency %d holds %zu entries, the layout says %zu", k, got, n);
    RC(gm_fr_alg_hash(tmp, ext_idx[k], zeta, sorted[k].h));
    V.release(tmp);
  }
  RC(commit_blocks(lay, K, {sorted[0], sorted[1], sorted[2]}, &P->sorted_commitments[0][0]));  // :179-183
  P->spans[5] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("sorted_alpha_commitment"), 23, P->sorted_commitments[1], 1, 0));  // :186-188
  RC(gm_transcript_append_g1(T.h, L("sorted_r_commitment"), 19, P->sorted_commitments[0], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("sorted_z_commitment"), 19, P->sorted_commitments[2], 1, 0));
  uint64_t gamma[4], chi[4];
  RC(gm_transcript_challenge_fr(T.h, L("gamma"), 5, gamma));  // :190-191
  RC(gm_transcript_challenge_fr(T.h, L("chi"), 3, chi));

  t0 = Clock::now();
  // the nine lookup vectors (plookup/time_prover.rs:89-112), this rank's block of each: l = whole length, s = level of the FAMILY (the
  // lookup vector, its accumulated product and its rotation, the longest of which has l + 1 entries)   :194-209
  BV lookup_vec[9];
  {
    const BV subsets[3] = {r_star, alpha_star, z_star};
    const uint64_t sub_idx[3] = {S->row_index, S->row_index, S->col_index};
    // halos of the sorted vectors: the last element of every full block
    std::vector<uint64_t> last(12, 0), lasts(12 * g, 0);
    for (int k = 0; k < 3; k++) {
      size_t n = 0;
      RC(vec_len(sorted[k].h, &n));
      if (n && n == lay.B(sorted[k].s)) RC(gm_fr_vec_download(sorted[k].h, n - 1, last.data() + 4 * k, 1));
    }
    RC(gm_dist_allgather_host(last.data(), 96, lasts.data()));
    for (int k = 0; k < 3; k++) {
      const size_t nset = set_len[k];
      // lookup_set = plookup_set(alg_hash(set)): nset + 1 entries; this rank needs the hashed set on [lo - 1, lo + in_set) only -- a RANGE of a
      // function of the index (or of z), hashed with its own first index
      BV& ls = lookup_vec[3 * k];
      ls = BV{0, nset + 1, lay.level(nset + 2)};
      const size_t out_set = lay.cnt(nset + 1, ls.s), in_set = lay.cnt(nset, ls.s), lo_s = lay.lo(ls.s);
      RC(alloc_len(V, out_set, &ls.h));
      if (out_set) {
        const size_t halo = lo_s ? 1 : 0, first = lo_s - halo, cnt = in_set + halo;
        uint64_t rng, set_h;
        RC(alloc_len(V, cnt, &rng));
        RC(range_fn(k, first, cnt, rng));
        set_h = rng;
        if (hashed && cnt) {
          RC(alloc_len(V, cnt, &set_h));
          RC(gm_fr_alg_hash_from(rng, first, zeta, set_h));
          V.release(rng);
        }
        uint64_t prev[4];
        if (halo) RC(gm_fr_vec_download(set_h, 0, prev, 1));
        RC(gm_fr_plookup_set_block(set_h, halo, in_set, halo ? prev : nullptr, out_set, gamma, chi, ls.h));
        V.release(set_h);
      }
      // lookup_subset = alg_hash(subset, index) + y
      BV& lb = lookup_vec[3 * k + 1];
      lb = BV{0, nnz, sn};
      RC(alloc_len(V, nnz_blk, &lb.h));
      if (nnz_blk) {
        if (hashed) {
          uint64_t tmp;
          RC(V.alloc(nnz_blk, &tmp));
          RC(gm_fr_alg_hash(subsets[k].h, sub_idx[k], zeta, tmp));
          RC(gm_fr_add_scalar(tmp, gamma, lb.h));
          V.release(tmp);
        } else {
          RC(gm_fr_add_scalar(subsets[k].h, gamma, lb.h));
        }
      }
      // lookup_sorted = plookup_set(sorted_k): ext_len + 1 entries, the halo from the rank below
      BV& lt = lookup_vec[3 * k + 2];
      lt = BV{0, ext_len[k] + 1, ext_lv[k]};
      const size_t out_srt = lay.cnt(ext_len[k] + 1, lt.s), in_srt = lay.cnt(ext_len[k], lt.s);
      RC(alloc_len(V, out_srt, &lt.h));
      if (out_srt) RC(gm_fr_plookup_set_block(sorted[k].h, 0, in_srt, r ? lasts.data() + 12 * (r - 1) + 4 * k : nullptr, out_srt, gamma, chi, lt.h));
    }
  }
  // accumulated_product(monic(v)) and right_rotation(monic(v)) (entryproduct/time_prover.rs:14-51), l + 1 entries each   :211-214
  BV acc_vec[9], shift_lookup[9];
  {
    // one all-gather: the products of the blocks (the carries of the suffix scans) and their last elements (the halos of the rotations)
    std::vector<uint64_t> mine(72, 0), all(72 * g);
    for (int k = 0; k < 9; k++) {
      RC(gm_fr_product(lookup_vec[k].h, mine.data() + 4 * k));
      size_t n = 0;
      RC(vec_len(lookup_vec[k].h, &n));
      if (n && n == lay.B(lookup_vec[k].s)) RC(gm_fr_vec_download(lookup_vec[k].h, n - 1, mine.data() + 36 + 4 * k, 1));
    }
    RC(gm_dist_allgather_host(mine.data(), 576, all.data()));
    for (int k = 0; k < 9; k++) {
      Fr carry = Fr::one(), total = Fr::one();
      for (size_t rr = g; rr-- > 0;) {
        if (rr == r) carry = total;
        total = total * Fr::from_limbs(all.data() + 72 * rr + 4 * k);
      }
      total.to_limbs(P->products[k]);  // the full product is the first accumulated entry
      const size_t lk = lookup_vec[k].len, sk = lookup_vec[k].s, n_out = lay.cnt(lk + 1, sk), n_in = lay.cnt(lk, sk);
      uint64_t cl[4];
      carry.to_limbs(cl);
      acc_vec[k] = BV{0, lk + 1, sk};
      shift_lookup[k] = BV{0, lk + 1, sk};
      RC(alloc_len(V, n_out, &acc_vec[k].h));
      if (n_out) RC(gm_fr_acc_product_block(lookup_vec[k].h, cl, n_out == n_in + 1, acc_vec[k].h));
      RC(alloc_len(V, n_out, &shift_lookup[k].h));
      if (n_out) RC(gm_fr_shift_block(lookup_vec[k].h, r ? all.data() + 72 * (r - 1) + 36 + 4 * k : one, n_out, shift_lookup[k].h));
      V.release(lookup_vec[k].h);
    }
  }
  P->spans[6] = since(t0);
  RC(gm_transcript_append_fr(T.h, L("set_r_ep"), 8, P->products[3], 1));  // :216-221 (labels as in the reference)
  RC(gm_transcript_append_fr(T.h, L("subset_r_ep"), 11, P->products[4], 1));
  RC(gm_transcript_append_fr(T.h, L("set_r_ep"), 8, P->products[0], 1));
  RC(gm_transcript_append_fr(T.h, L("subset_r_ep"), 11, P->products[1], 1));
  RC(gm_transcript_append_fr(T.h, L("set_z_ep"), 8, P->products[6], 1));
  RC(gm_transcript_append_fr(T.h, L("subset_z_ep"), 11, P->products[7], 1));

  // EntryProduct::new_time_batch (entryproduct/time_prover.rs:53-114)   :223-239
  t0 = Clock::now();
  uint64_t psi[4];
  std::vector<Fr> acc_psi;
  {
    RC(commit_blocks(lay, K, std::vector<BV>(acc_vec, acc_vec + 9), &P->acc_v_commitments[0][0]));
    for (int k = 0; k < 9; k++) RC(gm_transcript_append_g1(T.h, L("acc_v"), 5, P->acc_v_commitments[k], 1, 0));
    RC(gm_transcript_challenge_fr(T.h, L("ep-chal"), 7, psi));
    RC(eval_blocks(lay, std::vector<BV>(acc_vec, acc_vec + 9), psi, 1, acc_psi));
    const Fr ci = Fr::from_limbs(psi);
    for (int k = 0; k < 9; k++) (acc_psi[k] * ci + Fr::from_limbs(P->products[k]) - fr_pow(ci, acc_vec[k].len)).to_limbs(P->claimed_sumchecks[k]);
  }
  P->spans[7] = since(t0);

  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));  // :241-242
  t0 = Clock::now();
  {
    std::vector<Item> polys;  // :244-251
    Fr e = Fr::one();
    const Fr oc = Fr::from_limbs(open_chal);
    polys.push_back({ralpha_star, e});
    for (int k = 0; k < 9; k++) {
      e = e * oc;
      polys.push_back({acc_vec[k], e});
    }
    RC(open_blocks(lay, K, V, polys, {}, psi, 1, P->ralpha_star_acc_mu_proof));
    std::vector<Fr> v0;
    RC(eval_blocks(lay, {ralpha_star}, psi, 1, v0));
    v0[0].to_limbs(P->ralpha_star_acc_mu_evals[0]);
    for (int k = 0; k < 9; k++) acc_psi[k].to_limbs(P->ralpha_star_acc_mu_evals[1 + k]);
  }
  P->spans[8] = since(t0);
  {
    uint64_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // :253-254
    if (nnz_blk) {
      uint64_t h_a, h_b;
      RC(V.alloc(nnz_blk, &h_a));
      RC(gm_fr_hadamard(ralpha_star.h, S->val_a, h_a));
      RC(V.alloc(nnz_blk, &h_b));
      RC(gm_fr_hadamard(r_star.h, S->val_b, h_b));
      RC(gm_fr_ip(h_a, second_challenges, mine));
      RC(gm_fr_ip(h_b, second_challenges, mine + 4));
      V.release(h_a);
      V.release(h_b);
    }
    std::vector<uint64_t> all(8 * g);
    RC(gm_dist_allgather_host(mine, 64, all.data()));
    Fr sa = Fr::zero(), sb = Fr::zero();
    for (size_t rr = 0; rr < g; rr++) {
      sa = sa + Fr::from_limbs(all.data() + 8 * rr);
      sb = sb + Fr::from_limbs(all.data() + 8 * rr + 4);
    }
    sa.to_limbs(P->rstars_vals[0]);
    sb.to_limbs(P->rstars_vals[1]);
  }
  for (int k = 0; k < 10; k++) RC(gm_transcript_append_fr(T.h, L("ralpha_star_acc_mu"), 18, P->ralpha_star_acc_mu_evals[k], 1));  // :258-261
  RC(gm_transcript_append_g1(T.h, L("ralpha_star_mu_proof"), 20, P->ralpha_star_acc_mu_proof, 1, 0));
  std::vector<uint64_t> borrowed_tmp;
  std::vector<ShProver> third;
  for (int k = 0; k < 9; k++) third.push_back({acc_vec[k].h, shift_lookup[k].h, psi, acc_vec[k].len, acc_vec[k].s});  // :223-239
  {
    const uint64_t lhs[3] = {ralpha_star.h, r_star.h, alpha_star.h}, rhs[3] = {S->val_a, S->val_b, S->val_c};  // :263-290
    for (int k = 0; k < 3; k++) {
      uint64_t h;
      RC(alloc_len(V, nnz_blk, &h));
      if (nnz_blk) RC(gm_fr_hadamard(lhs[k], second_challenges, h));
      third.push_back({h, nnz_blk ? rhs[k] : h, one, nnz, sn});
      borrowed_tmp.push_back(h);
    }
    third.push_back({r_star.h, alpha_star.h, psi, nnz, sn});
  }
  t0 = Clock::now();
  RC(sumcheck_blocks(lay, T.h, true, third, tail, P->messages[2], ch3.data(), cap_rounds, &P->third_final_foldings[0][0], &P->rounds[2]));  // :293
  V.release(second_challenges);
  for (uint64_t v : borrowed_tmp) V.release(v);
  P->spans[9] = since(t0);

  // ---- TensorcheckProof::new_time(transcript, ck, 22 base polynomials, 4 bodies)   :296-367, tensorcheck/mod.rs:190-275
  t0 = Clock::now();
  std::vector<BV> base = {w, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c, sorted[0], sorted[1], sorted[2]};
  base.insert(base.end(), acc_vec, acc_vec + 9);
  const size_t n3 = P->rounds[2], n2 = P->rounds[1];
  struct Body {
    std::vector<BV> polys;
    std::vector<uint64_t> challenges;  // 4 limbs each
    size_t len = 0, s = 0;             // of the batched polynomial: its whole length, its level (the coarsest among its members)
    std::vector<uint64_t> sharded;     // levels 1 .. of the folding tree that stay sharded (this rank's blocks, nominal length, zero-padded; 0: nothing here)
    std::vector<uint64_t> small;       // the gathered levels, replicated
    size_t first = 0;                  // index of its first folding in the proof
  };
  std::vector<Body> bodies(4);
  {
    bodies[0].polys.assign(acc_vec, acc_vec + 9);  // accumulated_vec + [r_star], challenges third_ch[j] * psi^(2^j)   :334-349
    bodies[0].polys.push_back(r_star);
    bodies[0].challenges.resize(4 * n3);
    Fr tw = Fr::from_limbs(psi);
    for (size_t j = 0; j < n3; j++) {
      (Fr::from_limbs(ch3.data() + 4 * j) * tw).to_limbs(bodies[0].challenges.data() + 4 * j);
      tw = tw.sqr();
    }
    bodies[1].polys.assign(shift_lookup, shift_lookup + 9);  // shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star], challenges third_ch
    bodies[1].polys.insert(bodies[1].polys.end(), {val_a, val_b, val_c, alpha_star});
    bodies[1].challenges.assign(ch3.begin(), ch3.begin() + 4 * n3);
    bodies[2].polys = {z_star};  // challenges second_ch
    bodies[2].challenges.assign(ch2.begin(), ch2.begin() + 4 * n2);
    bodies[3].polys = {ralpha_star, r_star, alpha_star};  // challenges second_ch[j] * third_ch[j]
    const size_t nh = n2 < n3 ? n2 : n3;
    bodies[3].challenges.resize(4 * nh);
    for (size_t j = 0; j < nh; j++)
      (Fr::from_limbs(ch2.data() + 4 * j) * Fr::from_limbs(ch3.data() + 4 * j)).to_limbs(bodies[3].challenges.data() + 4 * j);
  }
  uint64_t batch_challenge[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, batch_challenge));
  size_t max_group = 0;
  for (auto& b : bodies) max_group = std::max(max_group, b.polys.size());
  std::vector<Fr> bc(max_group);  // powers(batch_challenge, max_len)
  {
    Fr acc = Fr::one();
    const Fr c = Fr::from_limbs(batch_challenge);
    for (size_t k = 0; k < max_group; k++) {
      bc[k] = acc;
      acc = acc * c;
    }
  }
  // foldings_polynomial (tensorcheck/mod.rs:124-133) of every batched body: all challenges but the last.  The batched body is built at the
  // coarsest level s among its members (combine_levels); folding level j is a vector of level s + j: it keeps the block layout while
  // s + j <= jmax (zero-padded to its nominal length: the padding is beyond the end of the polynomial), level jmax + 1 is gathered, the
  // rest is folded replicated.  A body that does not reach this rank's block contributes nothing here.
  size_t nfold = 0;
  for (auto& b : bodies) {
    const size_t nch = b.challenges.size() / 4, nlev = nch ? nch - 1 : 0;
    b.first = nfold;
    nfold += nlev;
    std::vector<Item> items;
    for (size_t k = 0; k < b.polys.size(); k++) items.push_back({b.polys[k], bc[k]});
    b.s = ~(size_t)0;
    for (const BV& p : b.polys) {
      b.s = std::min(b.s, p.s);
      b.len = std::max(b.len, p.len);
    }
    if (!nlev) continue;
    BV batched;
    RC(combine_levels(lay, V, items, 0, &batched));  // (collective: every rank, whether the body reaches its block or not)
    const bool present = lay.lo(b.s) < b.len;
    uint64_t cur = batched.h;
    for (size_t j = 1; j <= nlev; j++) {
      const size_t lv = b.s + j;
      if (lv <= jmax + 1) {
        const size_t nominal = lay.B(lv);
        uint64_t nxt = 0;
        if (present) {
          RC(alloc_len(V, nominal, &nxt));
          RC(gm_fr_fold(cur, b.challenges.data() + 4 * (j - 1), nxt));
          RC(gm_fr_vec_set_len(nxt, nominal));
        }
        if (lv <= jmax) {
          b.sharded.push_back(nxt);  // (0: absent)
          cur = nxt;
          continue;
        }
        // lv = jmax + 1: gathered.  Every rank takes part, an absent block as zeros
        if (!present) RC(alloc_zero(V, nominal, nominal, &nxt));
        uint64_t full = nxt;
        if (g > 1) {
          RC(V.alloc(nominal * g, &full));
          RC(gm_dist_allgather_vec(nxt, full));
          V.release(nxt);
        }
        RC(gm_fr_vec_set_len(full, std::min(nominal * g, ceil_shift(b.len, j))));
        b.small.push_back(full);
        cur = full;
      } else {
        size_t len = 0;
        RC(vec_len(cur, &len));
        uint64_t nxt;
        RC(alloc_len(V, (len + 1) / 2, &nxt));
        RC(gm_fr_fold(cur, b.challenges.data() + 4 * (j - 1), nxt));
        b.small.push_back(nxt);
        cur = nxt;
      }
    }
    V.release(batched.h);
  }
  P->nfold = nfold;
  GM_CHECK(nfold <= P->cap_folds, GM_EINVAL, "psnark_new_time_sharded: %zu foldings exceed capacity %zu", nfold, P->cap_folds);
  if (nfold) {
    // one pipelined batch: the sharded levels against their slices, the gathered ones against the replicated prefix
    std::vector<size_t> levels, where;
    std::vector<uint64_t> vecs;
    size_t n_sharded = 0;
    for (auto& b : bodies)
      for (size_t i = 0; i < b.sharded.size(); i++) {
        levels.push_back(b.s + 1 + i);
        vecs.push_back(b.sharded[i]);
        where.push_back(b.first + i);
        n_sharded++;
      }
    for (auto& b : bodies)
      for (size_t i = 0; i < b.small.size(); i++) {
        levels.push_back(jmax + 1);
        vecs.push_back(b.small[i]);
        where.push_back(b.first + b.sharded.size() + i);
      }
    std::vector<uint64_t> parts(18 * vecs.size()), sums(18 * std::max<size_t>(n_sharded, 1));
    RC(key_commit(K, levels, vecs, parts.data()));
    RC(gather_sum(lay, parts.data(), n_sharded, sums.data()));
    for (size_t i = 0; i < vecs.size(); i++) {
      if (i < n_sharded) memcpy(P->fold_commitments + 18 * where[i], sums.data() + 18 * i, 144);
      else RC(gm_g1_sum(parts.data() + 18 * i, 1, P->fold_commitments + 18 * where[i]));
    }
  }
  for (size_t k = 0; k < nfold; k++) RC(gm_transcript_append_g1(T.h, L("commitment"), 10, P->fold_commitments + 18 * k, 1, 0));
  uint64_t pts[12];  // beta^2, beta, -beta
  RC(gm_transcript_challenge_fr(T.h, L("evaluation-chal"), 15, pts + 4));
  {
    const Fr beta = Fr::from_limbs(pts + 4);
    beta.sqr().to_limbs(pts);
    beta.neg().to_limbs(pts + 8);
  }
  {
    std::vector<Fr> vals;
    RC(eval_blocks(lay, base, pts, 3, vals));
    for (size_t k = 0; k < base.size(); k++)
      for (int q = 0; q < 3; q++) vals[3 * k + q].to_limbs(&P->base_evaluations[k][4 * q]);
    std::vector<BV> sh;
    std::vector<size_t> where;
    for (auto& b : bodies)
      for (size_t i = 0; i < b.sharded.size(); i++) {
        sh.push_back(BV{b.sharded[i], ceil_shift(b.len, 1 + i), b.s + 1 + i});
        where.push_back(b.first + i);
      }
    RC(eval_blocks(lay, sh, pts + 4, 2, vals));
    for (size_t i = 0; i < sh.size(); i++) {
      vals[2 * i].to_limbs(P->fold_evaluations + 8 * where[i]);
      vals[2 * i + 1].to_limbs(P->fold_evaluations + 8 * where[i] + 4);
    }
    for (auto& b : bodies)
      if (!b.small.empty()) RC(gm_fr_eval_le_batch(b.small.data(), b.small.size(), pts + 4, 2, P->fold_evaluations + 8 * (b.first + b.sharded.size())));
  }
  for (size_t k = 0; k < 3 * base.size(); k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, &P->base_evaluations[0][0] + 4 * k, 1));
  for (size_t k = 0; k < 2 * nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal2[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal2));
  {
    // all = base ++ foldings, eta_i = open_chal^i: base polynomials and sharded folding levels of every level go into ONE combination
    // (per level, re-blocked to the coarsest: by linearity at most jmax vectors cross the links, not 4 jmax + 22)
    const Fr oc = Fr::from_limbs(open_chal2);
    std::vector<Fr> etas(base.size() + nfold);
    {
      Fr acc = Fr::one();
      for (auto& e : etas) {
        e = acc;
        acc = acc * oc;
      }
    }
    std::vector<Item> items;
    std::vector<Piece> small;
    for (size_t k = 0; k < base.size(); k++) items.push_back({base[k], etas[k]});
    for (auto& b : bodies) {
      for (size_t i = 0; i < b.sharded.size(); i++)
        items.push_back({BV{b.sharded[i], ceil_shift(b.len, 1 + i), b.s + 1 + i}, etas[base.size() + b.first + i]});
      for (size_t i = 0; i < b.small.size(); i++) small.push_back({b.small[i], etas[base.size() + b.first + b.sharded.size() + i]});
    }
    RC(open_blocks(lay, K, V, items, small, pts, 3, P->evaluation_proof));
  }
  P->spans[10] = since(t0);
  P->spans[11] = since(t_all);
  return GM_OK;
}

}  // extern "C"

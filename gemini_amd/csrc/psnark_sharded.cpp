// psnark::Proof::new_time (src/psnark/time_prover.rs:69-384) with EVERY vector of the prover block-sharded over the ranks of
// gm_dist (one process per GPU): the field side of BASELINE configs[4] on N GPUs, not only its MSMs.
//
// Layout.  One block size M for the whole proof (a multiple of a power of two, gm_psnark_shard_block): rank r holds the elements
// [r M, (r + 1) M) that exist of every vector -- the joint-matrix vectors and index vectors of the instance (nnz), the lookup
// vectors (2^rounds + 1, nnz, 2^rounds + nnz + 1, ...), their accumulated products and rotations (+ 1), the folding levels
// (M / 2^j while that is even and >= 2^tail_log, gathered after).  Vectors of different lengths share the block size, so every linear
// combination of the protocol (the batched bodies of the tensor check, the opened polynomials) is a LOCAL pass.
// What crosses ranks, all through gm_dist's all-gather:
//   lookups              `lookup(v, index)` (plookup/time_prover.rs:5-8) gathers from tensor(rho), powers(alpha) and z.  The first two are
//                        FUNCTIONS of the index -- every rank computes them whole, an O(n) pass at HBM speed, cheaper than n elements over
//                        xGMI -- and z is the instance's (whole on every rank, as for the general matrices of gm_snark_new_time_sharded):
//                        the gathers are local, with this rank's block of the index vectors
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
  size_t r = 0, g = 1, M = 0;
  size_t lo() const { return r * M; }
  size_t cnt(size_t len) const { return len > r * M ? std::min(M, len - r * M) : 0; }
  size_t cnt_of(size_t rank, size_t len, size_t block) const { return len > rank * block ? std::min(block, len - rank * block) : 0; }
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

// values p(x) = sum_rr x^(rr * blen) P_rr(x) of block-sharded polynomials at npts points: local block evaluations, one all-gather
int eval_blocks(const Sh& lay, const std::vector<uint64_t>& blocks, const uint64_t* pts, size_t npts, const std::vector<size_t>& blen, std::vector<Fr>& vals) {
  const size_t k = blocks.size();
  vals.assign(k * npts, Fr::zero());
  if (k == 0) return GM_OK;
  std::vector<uint64_t> local(4 * k * npts, 0), allr(4 * k * npts * lay.g, 0);
  {
    std::vector<uint64_t> live, res;
    std::vector<size_t> at;
    for (size_t i = 0; i < k; i++) {
      size_t len = 0;
      if (blocks[i]) RC(vec_len(blocks[i], &len));
      if (len) {
        live.push_back(blocks[i]);
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
      const Fr step = fr_pow(x, blen[i]);
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
    if (vecs[i]) RC(vec_len(vecs[i], &len));
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
// ck.batch_commit of block-sharded vectors (level-0 slices)
int commit_blocks(const Sh& lay, const Key& K, const std::vector<uint64_t>& vecs, uint64_t* out) {
  std::vector<uint64_t> parts(18 * std::max<size_t>(vecs.size(), 1));
  RC(key_commit(K, std::vector<size_t>(vecs.size(), 0), vecs, parts.data()));
  return gather_sum(lay, parts.data(), vecs.size(), out);
}

// ---- Sumcheck::prove / prove_batch over blocks ---------------------------------------------------------------------------
// (src/subprotocols/sumcheck/proof.rs:36-122) k provers whose vectors are block-sharded with the SAME block size M.  While the
// blocks hold more than `tail` elements and stay pair-aligned, a round is shard-local: every rank's partial messages -- 64 bytes per
// prover -- are all-gathered in ONE call and added mod r.  Then the blocks are gathered once and every rank finishes the protocol
// on the whole (short) vectors.  batch = false: Sumcheck::prove of ONE prover (labels and round count differ).
struct ShProver {
  uint64_t f = 0, g = 0;  // this rank's blocks (length 0: nothing of the vectors falls into the block)
  const uint64_t* twist = nullptr;
  size_t len = 0;  // of the whole vectors
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
  size_t max_tot = 0, min_tot = ~(size_t)0;
  for (size_t j = 0; j < k; j++) {
    tot[j] = ceil_log2(P[j].len);  // time_prover.rs:35-38
    max_tot = std::max(max_tot, tot[j]);
    min_tot = std::min(min_tot, tot[j]);
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
  std::vector<size_t> blk(k, 0);
  for (size_t j = 0; j < k; j++) {
    size_t nf = 0, ng = 0;
    if (P[j].f) RC(vec_len(P[j].f, &nf));
    if (P[j].g) RC(vec_len(P[j].g, &ng));
    if (lay.g > 1 && (nf != lay.cnt(P[j].len) || ng != nf)) return GM_EINVAL;  // the blocks of a sharded prover tile its vectors
    blk[j] = std::min(nf, ng);
    if (lay.g == 1 ? (nf && ng) : blk[j] != 0) {
      RC(gm_sc_new_borrow(P[j].f, P[j].g, P[j].twist, &S.h[j]));
      if (lay.g > 1) RC(gm_sc_set_shard_rounds(S.h[j], lay.lo() / 2, tot[j]));
    } else if (lay.g == 1) {
      return GM_EINVAL;  // "sumcheck: empty vectors"
    }
  }
  size_t rd = 0, folds = 0, Mcur = lay.M;
  const uint64_t* vm = nullptr;
  bool replicated = lay.g == 1;
  std::vector<Fr> final_product(k);
  std::vector<char> finished(k, 0);
  for (;;) {
    if (!replicated && !(Mcur % 4 == 0 && Mcur > tail && rd < min_tot)) {
      // apply the pending fold shard-locally, then gather: the replicated provers start exactly at a message boundary
      if (vm) {
        for (size_t j = 0; j < k; j++)
          if (S.h[j]) RC(gm_sc_fold(S.h[j], vm));
        Mcur /= 2;
        folds++;
        vm = nullptr;
      }
      const size_t slot = 4 + 8 * Mcur;  // [count | f | g] per prover, limbs
      std::vector<uint64_t> mine(slot * k, 0), all(slot * k * lay.g);
      for (size_t j = 0; j < k; j++) {
        if (!S.h[j]) continue;
        size_t nf = 0, ng = 0;
        RC(gm_sc_lens(S.h[j], &nf, &ng, nullptr));
        if (nf != ng || nf > Mcur) return GM_ESTATE;
        mine[slot * j] = nf;
        RC(gm_sc_download(S.h[j], mine.data() + slot * j + 4, mine.data() + slot * j + 4 + 4 * Mcur));
      }
      RC(gm_dist_allgather_host(mine.data(), 8 * slot * k, all.data()));
      for (size_t j = 0; j < k; j++) {
        std::vector<uint64_t> fs, gs;
        for (size_t rr = 0; rr < lay.g; rr++) {
          const uint64_t* s = all.data() + slot * (rr * k + j);
          const size_t c = (size_t)s[0];
          if (c > Mcur) return GM_ESTATE;
          fs.insert(fs.end(), s + 4, s + 4 + 4 * c);
          gs.insert(gs.end(), s + 4 + 4 * Mcur, s + 4 + 4 * Mcur + 4 * c);
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
      }
      replicated = true;
    }
    if (batch && rd == rounds) break;
    Fr ma = Fr::zero(), mb = Fr::zero();
    std::vector<char> has(k, 0);
    for (size_t j = 0; j < k; j++) {
      if (!S.h[j] || finished[j]) continue;
      int h = 0;
      RC(gm_sc_round_begin(S.h[j], vm, &h));
      has[j] = (char)h;
    }
    std::vector<uint64_t> part(8 * k, 0);
    bool any = false;
    for (size_t j = 0; j < k; j++) {
      if (S.h[j] && !finished[j] && has[j]) {
        RC(gm_sc_round_end(S.h[j], part.data() + 8 * j, part.data() + 8 * j + 4));
        any = true;
      } else if (replicated) {
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
    if (!replicated) {
      any = true;  // (rd < min_tot: every prover has a message in this round, on some rank)
      std::vector<uint64_t> all(8 * k * lay.g);
      RC(gm_dist_allgather_host(part.data(), 64 * k, all.data()));
      for (size_t j = 0; j < k; j++) {
        Fr sa = Fr::zero(), sb = Fr::zero();
        for (size_t rr = 0; rr < lay.g; rr++) {
          sa = sa + Fr::from_limbs(all.data() + 8 * (rr * k + j));
          sb = sb + Fr::from_limbs(all.data() + 8 * (rr * k + j) + 4);
        }
        sa.to_limbs(part.data() + 8 * j);
        sb.to_limbs(part.data() + 8 * j + 4);
      }
    }
    if (vm) {
      if (!replicated) Mcur /= 2;
      folds++;
    }
    if (!batch && !any) break;
    if (rd >= cap_rounds) return GM_EINVAL;
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
// (src/kzg/time.rs:149-159) commit((sum_i eta_i p_i) div Z), Z = prod (x - pts[q]), npts <= 3.  F = sum eta_i p_i is ONE polynomial:
// rank r takes ITS coefficient range [r M, (r + 1) M) of F and commits the quotient of that block against the level-0 key slice it holds
// -- |F| / g pairs per rank.  The pieces of F:
//   at_M     blocks in the proof's own layout (block size M), coefficients eta
//   levels   level_sums[j - 1] = this rank's block (nominal M >> j elements, zero-padded) of sum_i eta_i p_i over the polynomials sharded in
//            blocks of M >> j (the folding levels): RE-BLOCKED to blocks of M first (gm_dist_reblock_vecs, one grouped send / recv)
//   small    replicated short polynomials: every rank takes its range
// The quotient of block r needs the carry from the blocks above: the polynomial c of degree < npts that agrees with
// S_r(x) = sum_{r' > r} x^((r' - r - 1) M) F_r'(x) at the roots of Z (one all-gather of npts evaluations per rank), and leaves a remainder
// that agrees with G = F_r + x^M c at the roots: q_r = (G - rem) / Z exactly, F div Z = sum_r x^(r M) q_r.
struct Piece {
  uint64_t v;
  Fr eta;
};
int open_blocks(const Sh& lay, const Key& K, Vecs& V, const std::vector<Piece>& at_M, const std::vector<uint64_t>& level_sums, const std::vector<Piece>& small,
                const uint64_t* pts, size_t npts, uint64_t out[18]) {
  if (npts < 1 || npts > 3) return GM_EINVAL;
  const size_t M = lay.M, r = lay.r, g = lay.g;
  std::vector<uint64_t> pieces, piece_eta, owned;
  auto add_piece = [&](uint64_t v, const Fr& eta) {
    pieces.push_back(v);
    piece_eta.resize(piece_eta.size() + 4);
    eta.to_limbs(piece_eta.data() + piece_eta.size() - 4);
  };
  for (const Piece& p : at_M) {
    size_t len = 0;
    if (p.v) RC(vec_len(p.v, &len));
    if (len) add_piece(p.v, p.eta);
  }
  if (!level_sums.empty()) {
    if (g == 1) {
      for (uint64_t v : level_sums) add_piece(v, Fr::one());
    } else {
      std::vector<uint64_t> outs(level_sums.size());
      for (size_t i = 0; i < outs.size(); i++) {
        RC(alloc_len(V, M, &outs[i]));
        owned.push_back(outs[i]);
      }
      RC(gm_dist_reblock_vecs(level_sums.data(), level_sums.size(), M, outs.data()));
      for (uint64_t v : outs) {
        size_t len = 0;
        RC(vec_len(v, &len));
        if (len) add_piece(v, Fr::one());
      }
    }
  }
  for (const Piece& p : small) {
    size_t len = 0;
    RC(vec_len(p.v, &len));
    if (r * M >= len) continue;
    if (r == 0 && len <= M) {
      add_piece(p.v, p.eta);
      continue;
    }
    const size_t cnt = std::min(M, len - r * M);
    uint64_t part;
    RC(alloc_len(V, cnt, &part));
    owned.push_back(part);
    RC(gm_fr_stride(p.v, r * M, 1, cnt, part));
    add_piece(part, p.eta);
  }
  uint64_t F;
  RC(alloc_zero(V, M + npts, 0, &F));
  owned.push_back(F);
  size_t lf = 0;
  if (!pieces.empty()) {
    RC(gm_fr_lincomb(pieces.data(), piece_eta.data(), pieces.size(), F));
    RC(vec_len(F, &lf));
  }
  if (lf > M) return GM_ESTATE;
  // F_r at the roots + its length, all ranks
  std::vector<uint64_t> mine_ev(4 * (npts + 1), 0), all_ev(4 * (npts + 1) * g);
  if (lf) RC(gm_fr_eval_le(F, pts, npts, mine_ev.data()));
  mine_ev[4 * npts] = lf;
  RC(gm_dist_allgather_host(mine_ev.data(), 32 * (npts + 1), all_ev.data()));
  bool above = false;  // something of F lives on a higher rank
  for (size_t rr = r + 1; rr < g; rr++) above = above || all_ev[(npts + 1) * 4 * rr + 4 * npts] != 0;
  uint64_t mine[18];
  memcpy(mine, identity_point(), 144);
  if (lf || above) {
    Fr xs[3], c[3] = {Fr::zero(), Fr::zero(), Fr::zero()}, gv[3], rem[3];
    for (size_t q = 0; q < npts; q++) xs[q] = Fr::from_limbs(pts + 4 * q);
    const size_t len_f = above ? M + npts : std::max(lf, npts);
    RC(gm_fr_vec_set_len(F, len_f));  // (the tail beyond the combination is the zero fill)
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
        const Fr step = fr_pow(xs[q], M);
        Fr acc = Fr::zero(), xp = Fr::one();
        for (size_t rr = r + 1; rr < g; rr++) {
          acc = acc + xp * Fr::from_limbs(all_ev.data() + (npts + 1) * 4 * rr + 4 * q);
          xp = xp * step;
        }
        ys[q] = acc;
      }
      interp(xs, ys, npts, c);
      for (size_t q = 0; q < npts; q++) seam(M + q, c[q]);
    }
    for (size_t q = 0; q < npts; q++) {
      Fr cx = Fr::zero();
      for (size_t t = npts; t-- > 0;) cx = cx * xs[q] + c[t];
      gv[q] = Fr::from_limbs(mine_ev.data() + 4 * q) + fr_pow(xs[q], M) * cx;
    }
    interp(xs, gv, npts, rem);
    for (size_t q = 0; q < npts; q++) seam(q, rem[q].neg());
    RC(gm_fr_add_at(F, pos.data(), val.data(), pos.size()));
    if (len_f > npts) {
      uint64_t quot, remz[12];
      RC(alloc_len(V, len_f - 1, &quot));  // (the division peels one linear factor at a time: room for the first quotient)
      owned.push_back(quot);
      RC(gm_fr_div_vanishing(F, pts, npts, quot, remz));
      for (size_t l = 0; l < 4 * npts; l++)
        if (remz[l] != 0) return GM_ESTATE;  // the block of the opening is not divisible by Z
      size_t lq = 0;
      RC(vec_len(quot, &lq));
      lq = std::min(lq, K.counts[0]);
      if (lq) {
        const size_t off0 = K.offsets[0];
        RC(gm_g1_msm_v_batch_at(K.h, &off0, 0, &quot, &lq, 1, 1, mine));
      }
    }
  }
  for (uint64_t v : owned) V.release(v);
  return gather_sum(lay, mine, 1, out);
}

int shard_layout(const gm_psnark_shard* S, Sh* lay, Key* K, size_t* jmax) {
  int rank = 0, world = 1;
  RC(gm_dist_info(&rank, &world, nullptr));
  lay->r = (size_t)rank;
  lay->g = (size_t)world;
  lay->M = S->block;
  if (lay->M < 4 || lay->M % 4 != 0 || S->tail_log < 2 || S->tail_log > 40) return GM_EINVAL;
  *jmax = sharded_levels(lay->M, S->tail_log);
  if (S->key_segments != *jmax + 2 || !S->key_offsets || !S->key_counts) return GM_EINVAL;
  K->h = S->key;
  K->offsets = S->key_offsets;
  K->counts = S->key_counts;
  K->segments = S->key_segments;
  return GM_OK;
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
  size_t jmax = 0;
  RC(shard_layout(S, &lay, &K, &jmax));
  return commit_blocks(lay, K, {S->row, S->col, S->val_a, S->val_b, S->val_c}, out_jac);
}

int gm_psnark_new_time_sharded(const gm_psnark_shard* S, int g1_encoding, size_t cap_rounds, gm_psnark_proof* P) {
  GM_CTX();
  GM_CHECK(S && P && S->index_commitments && P->messages[0] && P->messages[1] && P->messages[2] && P->fold_commitments && P->fold_evaluations, GM_EINVAL,
           "psnark_new_time_sharded: null pointer");
  const auto t_all = Clock::now();
  Sh lay;
  Key K;
  size_t jmax = 0;
  RC(shard_layout(S, &lay, &K, &jmax));
  const size_t M = lay.M, r = lay.r, g = lay.g, lo = lay.lo();
  const size_t tail = (size_t)1 << S->tail_log;
  const size_t nrows = S->num_constraints, nz = S->num_variables, nnz = S->nnz;
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
  uint64_t one[4];
  Fr::one().to_limbs(one);
  auto blk_len = [&](uint64_t v, size_t* n) -> int {
    *n = 0;
    return v ? vec_len(v, n) : GM_OK;
  };
  // the blocks the caller hands in tile their vectors
  {
    const uint64_t vs[6] = {S->w_block, S->row, S->col, S->val_a, S->val_b, S->val_c};
    const size_t lens[6] = {S->w_len, nnz, nnz, nnz, nnz, nnz};
    for (int k = 0; k < 6; k++) {
      size_t n = 0;
      RC(blk_len(vs[k], &n));
      GM_CHECK(n == lay.cnt(lens[k]), GM_EINVAL, "psnark_new_time_sharded: input block %d holds %zu elements, the layout says %zu", k, n, lay.cnt(lens[k]));
    }
  }

  // z_a, z_b, z_c (:74-76): row blocks, global columns
  uint64_t z_abc[3];
  {
    const uint64_t mats[3] = {S->a, S->b, S->c};
    const size_t rows_blk = lay.cnt(nrows);
    for (int k = 0; k < 3; k++) {
      RC(alloc_len(V, rows_blk, &z_abc[k]));
      if (!rows_blk) continue;
      size_t rows = 0, cols = 0;
      RC(gm_spm_shape(mats[k], &rows, &cols, nullptr));
      GM_CHECK(rows == rows_blk && cols == nz, GM_EINVAL, "psnark_new_time_sharded: matrix %d is %zu x %zu, expected the row block %zu x %zu", k, rows, cols, rows_blk, nz);
      RC(gm_spm_mul(mats[k], S->z, z_abc[k]));
    }
  }
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));

  auto t0 = Clock::now();
  RC(commit_blocks(lay, K, {S->w_block}, P->witness_commitment));  // :79
  P->spans[0] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));  // :82-86
  RC(gm_transcript_append_message(T.h, L("ck"), 2, S->ck_g2_bytes, S->ck_g2_len));
  RC(gm_transcript_append_g1(T.h, L("instance"), 8, S->index_commitments, 5, 1));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  {
    std::vector<Fr> vals;
    RC(eval_blocks(lay, {z_abc[2]}, alpha, 1, {M}, vals));  // :88-89
    vals[0].to_limbs(P->zc_alpha);
  }
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1(4 * cap_rounds), ch2(4 * cap_rounds), ch3(4 * cap_rounds);
  RC(sumcheck_blocks(lay, T.h, false, {ShProver{z_abc[0], z_abc[1], alpha, nrows}}, tail, P->messages[0], ch1.data(), cap_rounds, P->final_foldings[0], &P->rounds[0]));  // :92
  P->spans[1] = since(t0);
  for (int k = 0; k < 3; k++) V.release(z_abc[k]);

  t0 = Clock::now();
  const size_t nt = (size_t)1 << P->rounds[0];
  // extend_frequency(compute_frequency(set_len, index)) has set_len + |index| entries (plookup/time_prover.rs:66-79)
  GM_CHECK(S->ext_fre_row_len == nt + nnz && S->ext_fre_col_len == nz + nnz, GM_EINVAL, "psnark_new_time_sharded: extended frequencies of %zu / %zu entries, expected %zu / %zu",
           S->ext_fre_row_len, S->ext_fre_col_len, nt + nnz, nz + nnz);
  // tensor(rho) and powers(alpha) WHOLE on every rank (:95-97): functions of the index, an O(n) pass each; their product is only ever
  // needed at the looked-up positions
  uint64_t b_ch, c_ch;
  RC(V.alloc(nt, &b_ch));
  RC(gm_fr_tensor(ch1.data(), P->rounds[0], b_ch));
  RC(V.alloc(nt, &c_ch));
  RC(gm_fr_powers(alpha, nt, c_ch));
  P->spans[2] = since(t0);

  const size_t nnz_blk = lay.cnt(nnz);
  uint64_t ralpha_star, r_star, alpha_star, z_star;  // :114-117, this rank's block of the index vectors
  RC(alloc_len(V, nnz_blk, &r_star));
  RC(alloc_len(V, nnz_blk, &alpha_star));
  RC(alloc_len(V, nnz_blk, &ralpha_star));
  RC(alloc_len(V, nnz_blk, &z_star));
  if (nnz_blk) {
    RC(gm_fr_gather(b_ch, S->row_index, r_star));
    RC(gm_fr_gather(c_ch, S->row_index, alpha_star));
    RC(gm_fr_hadamard(r_star, alpha_star, ralpha_star));
    RC(gm_fr_gather(S->z, S->col_index, z_star));
    size_t n1 = 0;
    RC(vec_len(r_star, &n1));
    GM_CHECK(n1 == nnz_blk, GM_EINVAL, "psnark_new_time_sharded: the block of the row index holds %zu entries, the layout says %zu", n1, nnz_blk);
  }

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
    const uint64_t lhs[3] = {ralpha_star, r_star, alpha_star}, rhs[3] = {S->val_a, S->val_b, S->val_c};
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
  RC(sumcheck_blocks(lay, T.h, false, {ShProver{z_star, r_star_val, one, nnz}}, tail, P->messages[1], ch2.data(), cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :147-152
  GM_CHECK(((size_t)1 << P->rounds[1]) >= nnz, GM_ESTATE, "psnark_new_time_sharded: %zu rounds for %zu entries", P->rounds[1], nnz);
  uint64_t second_challenges;  // &tensor(second challenges)[..num_non_zero], this rank's block
  RC(alloc_len(V, nnz_blk, &second_challenges));
  if (nnz_blk) RC(gm_fr_tensor_range(ch2.data(), P->rounds[1], lo, nnz_blk, second_challenges));
  V.release(r_star_val);
  P->spans[4] = since(t0);

  uint64_t zeta[4];
  RC(gm_transcript_challenge_fr(T.h, L("zeta"), 4, zeta));  // :157
  const bool hashed = !Fr::from_limbs(zeta).is_zero();

  t0 = Clock::now();
  // sorted_k = lookup(alg_hash(set_k), extended frequency) = set_k[e] + zeta e for e in this rank's block of the extended frequency (:160-173)
  const size_t ext_len[3] = {S->ext_fre_row_len, S->ext_fre_row_len, S->ext_fre_col_len};
  const uint64_t ext_idx[3] = {S->ext_fre_row, S->ext_fre_row, S->ext_fre_col};
  const uint64_t set_src[3] = {b_ch, c_ch, S->z};
  uint64_t sorted[3];
  for (int k = 0; k < 3; k++) {
    const size_t n = lay.cnt(ext_len[k]);
    RC(alloc_len(V, n, &sorted[k]));
    if (!n) continue;
    uint64_t tmp;
    RC(V.alloc(n, &tmp));
    RC(gm_fr_gather(set_src[k], ext_idx[k], tmp));
    size_t got = 0;
    RC(vec_len(tmp, &got));
    GM_CHECK(got == n, GM_EINVAL, "psnark_new_time_sharded: the block of extended frequency %d holds %zu entries, the layout says %zu", k, got, n);
    RC(gm_fr_alg_hash(tmp, ext_idx[k], zeta, sorted[k]));
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
  // the nine lookup vectors (plookup/time_prover.rs:89-112), this rank's block of each; l[k] = their whole lengths   :194-209
  uint64_t lookup_vec[9];
  size_t l[9];
  {
    const uint64_t subsets[3] = {r_star, alpha_star, z_star}, sub_idx[3] = {S->row_index, S->row_index, S->col_index};
    // halos of the sorted vectors: the last element of every full block
    std::vector<uint64_t> last(12, 0), lasts(12 * g, 0);
    for (int k = 0; k < 3; k++) {
      size_t n = 0;
      RC(vec_len(sorted[k], &n));
      if (n == M) RC(gm_fr_vec_download(sorted[k], M - 1, last.data() + 4 * k, 1));
    }
    RC(gm_dist_allgather_host(last.data(), 96, lasts.data()));
    for (int k = 0; k < 3; k++) {
      size_t nset = 0;
      RC(vec_len(set_src[k], &nset));
      // lookup_set = plookup_set(alg_hash(set)): nset + 1 entries, from the WHOLE hashed set (a replicated O(n) pass, transient)
      l[3 * k] = nset + 1;
      const size_t out_set = lay.cnt(nset + 1), in_set = lay.cnt(nset);
      RC(alloc_len(V, out_set, &lookup_vec[3 * k]));
      if (out_set) {
        uint64_t set_h = set_src[k];
        if (hashed) {
          RC(V.alloc(nset, &set_h));
          RC(gm_fr_alg_hash(set_src[k], 0, zeta, set_h));
        }
        uint64_t prev[4];
        if (lo) RC(gm_fr_vec_download(set_h, lo - 1, prev, 1));
        RC(gm_fr_plookup_set_block(set_h, in_set ? lo : 0, in_set, lo ? prev : nullptr, out_set, gamma, chi, lookup_vec[3 * k]));
        if (hashed) V.release(set_h);
      }
      // lookup_subset = alg_hash(subset, index) + y
      l[3 * k + 1] = nnz;
      RC(alloc_len(V, nnz_blk, &lookup_vec[3 * k + 1]));
      if (nnz_blk) {
        if (hashed) {
          uint64_t tmp;
          RC(V.alloc(nnz_blk, &tmp));
          RC(gm_fr_alg_hash(subsets[k], sub_idx[k], zeta, tmp));
          RC(gm_fr_add_scalar(tmp, gamma, lookup_vec[3 * k + 1]));
          V.release(tmp);
        } else {
          RC(gm_fr_add_scalar(subsets[k], gamma, lookup_vec[3 * k + 1]));
        }
      }
      // lookup_sorted = plookup_set(sorted_k): ext_len + 1 entries, the halo from the rank below
      l[3 * k + 2] = ext_len[k] + 1;
      const size_t out_srt = lay.cnt(ext_len[k] + 1), in_srt = lay.cnt(ext_len[k]);
      RC(alloc_len(V, out_srt, &lookup_vec[3 * k + 2]));
      if (out_srt) RC(gm_fr_plookup_set_block(sorted[k], 0, in_srt, r ? lasts.data() + 12 * (r - 1) + 4 * k : nullptr, out_srt, gamma, chi, lookup_vec[3 * k + 2]));
    }
  }
  V.release(b_ch);
  V.release(c_ch);
  // accumulated_product(monic(v)) and right_rotation(monic(v)) (entryproduct/time_prover.rs:14-51), l + 1 entries each   :211-214
  uint64_t acc_vec[9];
  std::vector<uint64_t> shift_lookup(9);
  {
    // one all-gather: the products of the blocks (the carries of the suffix scans) and their last elements (the halos of the rotations)
    std::vector<uint64_t> mine(72, 0), all(72 * g);
    for (int k = 0; k < 9; k++) {
      RC(gm_fr_product(lookup_vec[k], mine.data() + 4 * k));
      size_t n = 0;
      RC(vec_len(lookup_vec[k], &n));
      if (n == M) RC(gm_fr_vec_download(lookup_vec[k], M - 1, mine.data() + 36 + 4 * k, 1));
    }
    RC(gm_dist_allgather_host(mine.data(), 576, all.data()));
    for (int k = 0; k < 9; k++) {
      Fr carry = Fr::one(), total = Fr::one();
      for (size_t rr = g; rr-- > 0;) {
        if (rr == r) carry = total;
        total = total * Fr::from_limbs(all.data() + 72 * rr + 4 * k);
      }
      total.to_limbs(P->products[k]);  // the full product is the first accumulated entry
      const size_t n_out = lay.cnt(l[k] + 1), n_in = lay.cnt(l[k]);
      uint64_t cl[4];
      carry.to_limbs(cl);
      RC(alloc_len(V, n_out, &acc_vec[k]));
      if (n_out) RC(gm_fr_acc_product_block(lookup_vec[k], cl, n_out == n_in + 1, acc_vec[k]));
      RC(alloc_len(V, n_out, &shift_lookup[k]));
      if (n_out) RC(gm_fr_shift_block(lookup_vec[k], r ? all.data() + 72 * (r - 1) + 36 + 4 * k : one, n_out, shift_lookup[k]));
      V.release(lookup_vec[k]);
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
    RC(commit_blocks(lay, K, std::vector<uint64_t>(acc_vec, acc_vec + 9), &P->acc_v_commitments[0][0]));
    for (int k = 0; k < 9; k++) RC(gm_transcript_append_g1(T.h, L("acc_v"), 5, P->acc_v_commitments[k], 1, 0));
    RC(gm_transcript_challenge_fr(T.h, L("ep-chal"), 7, psi));
    RC(eval_blocks(lay, std::vector<uint64_t>(acc_vec, acc_vec + 9), psi, 1, std::vector<size_t>(9, M), acc_psi));
    const Fr ci = Fr::from_limbs(psi);
    for (int k = 0; k < 9; k++) (acc_psi[k] * ci + Fr::from_limbs(P->products[k]) - fr_pow(ci, l[k] + 1)).to_limbs(P->claimed_sumchecks[k]);
  }
  P->spans[7] = since(t0);

  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));  // :241-242
  t0 = Clock::now();
  {
    std::vector<Piece> polys;  // :244-251
    Fr e = Fr::one();
    const Fr oc = Fr::from_limbs(open_chal);
    polys.push_back({ralpha_star, e});
    for (int k = 0; k < 9; k++) {
      e = e * oc;
      polys.push_back({acc_vec[k], e});
    }
    RC(open_blocks(lay, K, V, polys, {}, {}, psi, 1, P->ralpha_star_acc_mu_proof));
    std::vector<Fr> v0;
    RC(eval_blocks(lay, {ralpha_star}, psi, 1, {M}, v0));
    v0[0].to_limbs(P->ralpha_star_acc_mu_evals[0]);
    for (int k = 0; k < 9; k++) acc_psi[k].to_limbs(P->ralpha_star_acc_mu_evals[1 + k]);
  }
  P->spans[8] = since(t0);
  {
    uint64_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // :253-254
    if (nnz_blk) {
      uint64_t h_a, h_b;
      RC(V.alloc(nnz_blk, &h_a));
      RC(gm_fr_hadamard(ralpha_star, S->val_a, h_a));
      RC(V.alloc(nnz_blk, &h_b));
      RC(gm_fr_hadamard(r_star, S->val_b, h_b));
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
  for (int k = 0; k < 9; k++) third.push_back({acc_vec[k], shift_lookup[k], psi, l[k] + 1});  // :223-239
  {
    const uint64_t lhs[3] = {ralpha_star, r_star, alpha_star}, rhs[3] = {S->val_a, S->val_b, S->val_c};  // :263-290
    for (int k = 0; k < 3; k++) {
      uint64_t h;
      RC(alloc_len(V, nnz_blk, &h));
      if (nnz_blk) RC(gm_fr_hadamard(lhs[k], second_challenges, h));
      third.push_back({h, nnz_blk ? rhs[k] : h, one, nnz});
      borrowed_tmp.push_back(h);
    }
    third.push_back({r_star, alpha_star, psi, nnz});
  }
  t0 = Clock::now();
  RC(sumcheck_blocks(lay, T.h, true, third, tail, P->messages[2], ch3.data(), cap_rounds, &P->third_final_foldings[0][0], &P->rounds[2]));  // :293
  V.release(second_challenges);
  for (uint64_t v : borrowed_tmp) V.release(v);
  P->spans[9] = since(t0);

  // ---- TensorcheckProof::new_time(transcript, ck, 22 base polynomials, 4 bodies)   :296-367, tensorcheck/mod.rs:190-275
  t0 = Clock::now();
  std::vector<uint64_t> base = {S->w_block, ralpha_star, r_star, alpha_star, z_star, S->row, S->col, S->val_a, S->val_b, S->val_c, sorted[0], sorted[1], sorted[2]};
  base.insert(base.end(), acc_vec, acc_vec + 9);
  std::vector<size_t> base_len = {S->w_len, nnz, nnz, nnz, nnz, nnz, nnz, nnz, nnz, nnz, ext_len[0], ext_len[1], ext_len[2]};
  for (int k = 0; k < 9; k++) base_len.push_back(l[k] + 1);
  const size_t n3 = P->rounds[2], n2 = P->rounds[1];
  struct Body {
    std::vector<uint64_t> polys;
    std::vector<size_t> lens;          // whole lengths
    std::vector<uint64_t> challenges;  // 4 limbs each
    size_t len = 0;                    // of the batched polynomial
    std::vector<uint64_t> sharded;     // levels 1 .. of the folding tree that stay sharded (this rank's blocks, nominal length, zero-padded)
    std::vector<uint64_t> small;       // the gathered levels, replicated
    size_t first = 0;                  // index of its first folding in the proof
  };
  std::vector<Body> bodies(4);
  {
    bodies[0].polys.assign(acc_vec, acc_vec + 9);  // accumulated_vec + [r_star], challenges third_ch[j] * psi^(2^j)   :334-349
    for (int k = 0; k < 9; k++) bodies[0].lens.push_back(l[k] + 1);
    bodies[0].polys.push_back(r_star);
    bodies[0].lens.push_back(nnz);
    bodies[0].challenges.resize(4 * n3);
    Fr tw = Fr::from_limbs(psi);
    for (size_t j = 0; j < n3; j++) {
      (Fr::from_limbs(ch3.data() + 4 * j) * tw).to_limbs(bodies[0].challenges.data() + 4 * j);
      tw = tw.sqr();
    }
    bodies[1].polys = shift_lookup;  // shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star], challenges third_ch
    for (int k = 0; k < 9; k++) bodies[1].lens.push_back(l[k] + 1);
    bodies[1].polys.insert(bodies[1].polys.end(), {S->val_a, S->val_b, S->val_c, alpha_star});
    bodies[1].lens.insert(bodies[1].lens.end(), {nnz, nnz, nnz, nnz});
    bodies[1].challenges.assign(ch3.begin(), ch3.begin() + 4 * n3);
    bodies[2].polys = {z_star};  // challenges second_ch
    bodies[2].lens = {nnz};
    bodies[2].challenges.assign(ch2.begin(), ch2.begin() + 4 * n2);
    bodies[3].polys = {ralpha_star, r_star, alpha_star};  // challenges second_ch[j] * third_ch[j]
    bodies[3].lens = {nnz, nnz, nnz};
    const size_t nh = n2 < n3 ? n2 : n3;
    bodies[3].challenges.resize(4 * nh);
    for (size_t j = 0; j < nh; j++)
      (Fr::from_limbs(ch2.data() + 4 * j) * Fr::from_limbs(ch3.data() + 4 * j)).to_limbs(bodies[3].challenges.data() + 4 * j);
  }
  uint64_t batch_challenge[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, batch_challenge));
  size_t max_group = 0;
  for (auto& b : bodies) max_group = std::max(max_group, b.polys.size());
  std::vector<uint64_t> bc(4 * max_group);  // powers(batch_challenge, max_len)
  {
    Fr acc = Fr::one();
    const Fr c = Fr::from_limbs(batch_challenge);
    for (size_t k = 0; k < max_group; k++) {
      acc.to_limbs(bc.data() + 4 * k);
      acc = acc * c;
    }
  }
  // foldings_polynomial (tensorcheck/mod.rs:124-133) of every batched body: all challenges but the last.  Levels 1 .. jmax keep the
  // block layout (blocks of M >> j, zero-padded to their nominal length: the padding is beyond the end of the polynomial), level
  // jmax + 1 is gathered, the rest is folded replicated.  A body none of whose polynomials reaches this block contributes nothing here.
  size_t nfold = 0;
  for (auto& b : bodies) {
    for (size_t len : b.lens) b.len = std::max(b.len, len);
    const size_t nch = b.challenges.size() / 4, nlev = nch ? nch - 1 : 0;
    b.first = nfold;
    nfold += nlev;
    if (!nlev) continue;
    const bool present = lo < b.len;
    uint64_t cur = 0;
    if (present) {
      std::vector<uint64_t> live, cf;
      for (size_t k = 0; k < b.polys.size(); k++) {
        size_t n = 0;
        RC(blk_len(b.polys[k], &n));
        if (!n) continue;
        live.push_back(b.polys[k]);
        cf.insert(cf.end(), bc.begin() + 4 * k, bc.begin() + 4 * k + 4);
      }
      RC(alloc_zero(V, M, 0, &cur));
      if (!live.empty()) RC(gm_fr_lincomb(live.data(), cf.data(), live.size(), cur));
      RC(gm_fr_vec_set_len(cur, M));
    }
    uint64_t batched = cur;
    for (size_t j = 1; j <= nlev; j++) {
      const size_t nominal = M >> j;
      if (j <= jmax + 1) {
        uint64_t nxt = 0;
        if (present) {
          RC(alloc_len(V, nominal, &nxt));
          RC(gm_fr_fold(cur, b.challenges.data() + 4 * (j - 1), nxt));
          RC(gm_fr_vec_set_len(nxt, nominal));
        }
        if (j <= jmax) {
          b.sharded.push_back(nxt);  // (0: absent)
          cur = nxt;
          continue;
        }
        // j = jmax + 1: gathered.  Every rank takes part, an absent block as zeros
        if (!present) RC(alloc_zero(V, nominal, nominal, &nxt));
        uint64_t full;
        RC(V.alloc(nominal * g, &full));
        if (g > 1) {
          RC(gm_dist_allgather_vec(nxt, full));
          V.release(nxt);
        } else {
          V.release(full);
          full = nxt;
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
    if (batched) V.release(batched);
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
        levels.push_back(1 + i);
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
    RC(eval_blocks(lay, base, pts, 3, std::vector<size_t>(base.size(), M), vals));
    for (size_t k = 0; k < base.size(); k++)
      for (int q = 0; q < 3; q++) vals[3 * k + q].to_limbs(&P->base_evaluations[k][4 * q]);
    std::vector<uint64_t> sh;
    std::vector<size_t> blen, where;
    for (auto& b : bodies)
      for (size_t i = 0; i < b.sharded.size(); i++) {
        sh.push_back(b.sharded[i]);
        blen.push_back(M >> (1 + i));
        where.push_back(b.first + i);
      }
    RC(eval_blocks(lay, sh, pts + 4, 2, blen, vals));
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
    // all = base ++ foldings, eta_i = open_chal^i.  The sharded levels of the four trees are summed PER LEVEL before they are re-blocked
    // (by linearity): jmax vectors cross the links instead of 4 jmax
    const Fr oc = Fr::from_limbs(open_chal2);
    std::vector<Fr> etas(base.size() + nfold);
    {
      Fr acc = Fr::one();
      for (auto& e : etas) {
        e = acc;
        acc = acc * oc;
      }
    }
    std::vector<Piece> at_M, small;
    for (size_t k = 0; k < base.size(); k++) at_M.push_back({base[k], etas[k]});
    std::vector<uint64_t> level_sums;
    for (size_t j = 1; j <= jmax; j++) {
      std::vector<uint64_t> live, cf;
      bool exists = false;
      for (auto& b : bodies) {
        if (b.sharded.size() < j) continue;
        exists = true;
        if (!b.sharded[j - 1]) continue;
        live.push_back(b.sharded[j - 1]);
        cf.resize(cf.size() + 4);
        etas[base.size() + b.first + j - 1].to_limbs(cf.data() + cf.size() - 4);
      }
      if (!exists) break;
      uint64_t sum;
      RC(alloc_zero(V, M >> j, 0, &sum));
      if (!live.empty()) RC(gm_fr_lincomb(live.data(), cf.data(), live.size(), sum));
      RC(gm_fr_vec_set_len(sum, M >> j));
      level_sums.push_back(sum);
    }
    for (auto& b : bodies)
      for (size_t i = 0; i < b.small.size(); i++) small.push_back({b.small[i], etas[base.size() + b.first + b.sharded.size() + i]});
    RC(open_blocks(lay, K, V, at_M, level_sums, small, pts, 3, P->evaluation_proof));
  }
  P->spans[10] = since(t0);
  P->spans[11] = since(t_all);
  return GM_OK;
}

}  // extern "C"

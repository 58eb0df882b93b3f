// psnark::Proof::new_elastic (src/psnark/elastic_prover.rs:60-634) as ONE entry point of the library: the preprocessing SNARK over
// device-resident STREAMS (big-endian reversed vectors), with
//   * CommitterKeyStream::{commit, open, open_multi_points, commit_folding, open_folding} (src/kzg/space.rs:95-285) as chunked
//     stream MSMs -- `Reverse(powers_of_g)` + advance_by is the reversed / offset addressing of gm_ck_msm, a flush every
//     max(max_msm_buffer [/ depth], min_device_chunk) pairs;
//   * Sumcheck::{new_space, new_elastic} (sumcheck/proof.rs:133-154) and prove_batch (:69-122) over ElasticProvers: a SpaceProver
//     that becomes a TimeProver once fewer than SPACE_TIME_THRESHOLD rounds remain (elastic_prover.rs:44-57);
//   * EntryProduct::new_elastic_batch (entryproduct/elastic_prover.rs:66-127), the plookup streams (:227-232 of the prover) and the
//     tensor check over FoldedPolynomialTrees (:384-600).
// Same transcript, same bytes as gm_psnark_new_time (src/psnark/tests.rs:56-124 asserts time == elastic); the sequence is the one
// of gemini_amd/psnark.py::Proof.new_elastic, which the tests hold byte for byte equal to this.  Pure orchestration over the
// library's own C ABI, like snark.cpp / psnark.cpp.  The reference re-streams every vector from its source to stay in O(log n)
// memory; with the streams resident in HBM they are materialised once.
#include <algorithm>

#include "prover_common.hpp"

namespace {

using namespace gmprover;

// `impl Prover for ElasticProver` (sumcheck/elastic_prover.rs:20-95) with split-phase rounds for prove_batch
struct ElasticSc {
  uint64_t space = 0, time = 0;
  bool allow_switch = true;  // false: Sumcheck::new_space, a SpaceProver to the end
  // round in flight
  bool pending_time = false, have_msg = false;
  uint64_t a[4], b[4];
  ElasticSc() = default;
  ElasticSc(const ElasticSc&) = delete;
  ElasticSc& operator=(const ElasticSc&) = delete;
  ~ElasticSc() { reset(); }
  void reset() {
    if (time) (void)gm_sc_free(time);
    if (space) (void)gm_sp_free(space);
    time = space = 0;
  }
  int init(uint64_t f_stream, uint64_t g_stream, const uint64_t twist[4], bool elastic) {
    allow_switch = elastic;
    // no copy: the space prover reads the caller's streams until it is freed (every caller below keeps them that long)
    return gm_sp_new_borrow(f_stream, g_stream, twist, &space);
  }
  // The RESIDENT schedule: the little-endian vectors behind the streams are in HBM anyway, so the prover is a time prover from
  // its first round (it reads them in place until its first fold) instead of a space prover that re-derives every message from
  // the whole streams until SPACE_TIME_THRESHOLD rounds remain.  The messages are the same field elements (sumcheck/tests.rs:42-87:
  // space == time), and it needs LESS memory than the reversed stream copies a device-side space prover reads (0.75 of them).
  int init_resident(uint64_t f_le, uint64_t g_le, const uint64_t twist[4]) {
    allow_switch = true;
    return gm_sc_new_borrow(f_le, g_le, twist, &time);
  }
  int rounds(size_t* tot) const { return time ? gm_sc_rounds(time, tot, nullptr) : gm_sp_rounds(space, tot, nullptr); }
  // next_message(vm), first half: fold (switching to the time prover when it is time), launch the round
  int begin(const uint64_t* vm, int* has) {
    if (vm && !time) {
      size_t tot = 0, rnd = 0;
      RC(gm_sp_rounds(space, &tot, &rnd));
      if (allow_switch && tot - rnd < SPACE_TIME_THRESHOLD) {
        RC(gm_sp_to_time(space, &time));
        RC(gm_sc_fold(time, vm));
        (void)gm_sp_free(space);
        space = 0;
      } else {
        RC(gm_sp_fold(space, vm));
      }
      vm = nullptr;
    }
    if (time) {
      RC(gm_sc_round_begin(time, vm, has));
      pending_time = *has != 0;
      have_msg = false;
    } else {
      RC(gm_sp_round(space, nullptr, a, b, has));
      pending_time = false;
      have_msg = *has != 0;
    }
    return GM_OK;
  }
  // a prover that is a time prover already takes part in the ONE launch of its round (gm_sc_round_begin_many)
  bool is_time() const { return time != 0; }
  uint64_t time_handle() const { return time; }
  void begun_as_time(int has) {
    pending_time = has != 0;
    have_msg = false;
  }
  int end(uint64_t out_a[4], uint64_t out_b[4]) {
    if (pending_time) {
      pending_time = false;
      return gm_sc_round_end(time, out_a, out_b);
    }
    if (!have_msg) return GM_ESTATE;
    memcpy(out_a, a, 32);
    memcpy(out_b, b, 32);
    have_msg = false;
    return GM_OK;
  }
  int final(uint64_t f0[4], uint64_t g0[4], int* has) { return time ? gm_sc_final(time, f0, g0, has) : gm_sp_final(space, f0, g0, has); }
};

// Sumcheck::prove (proof.rs:36-66) over one ElasticSc
int prove_one(uint64_t transcript, ElasticSc& S, uint64_t* messages, std::vector<uint64_t>& challenges, size_t cap_rounds, uint64_t final_foldings[8],
              size_t* rounds) {
  challenges.assign(cap_rounds * 4, 0);
  size_t k = 0;
  const uint64_t* vm = nullptr;
  for (;;) {
    int has = 0;
    RC(S.begin(vm, &has));
    if (!has) break;
    if (k >= cap_rounds) return GM_EINVAL;
    RC(S.end(messages + 8 * k, messages + 8 * k + 4));
    RC(gm_transcript_append_fr(transcript, L("evaluations"), 11, messages + 8 * k, 2));
    RC(gm_transcript_challenge_fr(transcript, L("challenge"), 9, challenges.data() + 4 * k));
    vm = challenges.data() + 4 * k;
    k++;
  }
  int has = 0;
  RC(S.final(final_foldings, final_foldings + 4, &has));
  if (!has) return GM_ESTATE;
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings, 1));
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings + 4, 1));
  *rounds = k;
  challenges.resize(k * 4);
  return GM_OK;
}

// Sumcheck::prove_batch (proof.rs:69-122) over ElasticProvers: the round of every live prover is enqueued before the first wait
int prove_batch(uint64_t transcript, std::vector<ElasticSc>& provers, uint64_t* messages, uint64_t* challenges, size_t cap_rounds, uint64_t* final_foldings,
                size_t* rounds_out) {
  const size_t k = provers.size();
  size_t rounds = 0;
  for (auto& p : provers) {
    size_t t = 0;
    RC(p.rounds(&t));
    rounds = std::max(rounds, t);
  }
  rounds += 1;
  if (rounds > cap_rounds) return GM_EINVAL;
  std::vector<Fr> coeff(k), final_product(k);
  for (size_t j = 0; j < k; j++) {
    uint64_t c[4];
    RC(gm_transcript_challenge_fr(transcript, L("batch-sumcheck"), 14, c));
    coeff[j] = Fr::from_limbs(c);
  }
  std::vector<char> finished(k, 0), has(k, 0);
  const uint64_t* vm = nullptr;
  for (size_t r = 0; r < rounds; r++) {
    Fr ma = Fr::zero(), mb = Fr::zero();
    {
      // the time provers among the live ones share one launch; a space prover (the literal schedule) steps on its own
      std::vector<uint64_t> th;
      std::vector<size_t> at;
      for (size_t j = 0; j < k; j++) {
        if (finished[j]) continue;
        if (provers[j].is_time()) {
          th.push_back(provers[j].time_handle());
          at.push_back(j);
          continue;
        }
        int h = 0;
        RC(provers[j].begin(vm, &h));
        has[j] = (char)h;
      }
      std::vector<int> hs(th.size(), 0);
      RC(gm_sc_round_begin_many(th.data(), th.size(), vm, hs.data()));
      for (size_t t = 0; t < th.size(); t++) {
        provers[at[t]].begun_as_time(hs[t]);
        has[at[t]] = (char)hs[t];
      }
    }
    for (size_t j = 0; j < k; j++) {
      Fr fa, fb;
      if (!finished[j] && has[j]) {
        uint64_t a[4], b[4];
        RC(provers[j].end(a, b));
        fa = Fr::from_limbs(a);
        fb = Fr::from_limbs(b);
      } else {
        if (!finished[j]) {
          uint64_t f0[4], g0[4];
          int hf = 0;
          RC(provers[j].final(f0, g0, &hf));
          if (!hf) return GM_ESTATE;
          final_product[j] = Fr::from_limbs(f0) * Fr::from_limbs(g0);
          finished[j] = 1;
        }
        fa = final_product[j];
        fb = Fr::zero();
      }
      ma = ma + fa * coeff[j];
      mb = mb + fb * coeff[j];
    }
    ma.to_limbs(messages + 8 * r);
    mb.to_limbs(messages + 8 * r + 4);
    RC(gm_transcript_append_fr(transcript, L("evaluations"), 11, messages + 8 * r, 2));
    RC(gm_transcript_challenge_fr(transcript, L("challenge"), 9, challenges + 4 * r));
    vm = challenges + 4 * r;
  }
  for (size_t j = 0; j < k; j++) {
    int hf = 0;
    RC(provers[j].final(final_foldings + 8 * j, final_foldings + 8 * j + 4, &hf));
    if (!hf) return GM_ESTATE;
    RC(gm_transcript_append_fr(transcript, L("final-folding-lhs"), 17, final_foldings + 8 * j, 1));
    RC(gm_transcript_append_fr(transcript, L("final-folding-rhs"), 17, final_foldings + 8 * j + 4, 1));
  }
  *rounds_out = rounds;
  return GM_OK;
}

struct StreamKey {
  uint64_t ck;
  size_t nck, max_msm_buffer, min_chunk;
  size_t flush(size_t wanted) const { return std::max<size_t>(std::max<size_t>(wanted, 1), min_chunk); }
  // ck.commit(stream of a little-endian vector): msm_chunks of 2^20 (space.rs:169-177).  The vector is reversed into its stream
  // and walked against Reverse(powers_of_g)
  int commit_le(Vecs& V, uint64_t le, uint64_t out[18]) const { return msm_le(V, le, (size_t)1 << 20, out); }
  int msm_le(Vecs& V, uint64_t le, size_t wanted_flush, uint64_t out[18]) const {
    size_t n = 0;
    RC(vec_len(le, &n));
    if (n > nck) return GM_EINVAL;  // the streaming committer insists on a key as long as the stream (space.rs:169-175)
    if (n == 0) return gm_g1_sum(nullptr, 0, out);
    uint64_t s;
    RC(V.alloc(n, &s));
    RC(gm_fr_reverse(le, s));
    const int rc = stream_msm(ck, s, n, n - 1, flush(wanted_flush), out);
    V.release(s);
    return rc;
  }
  // several commitments: when no vector is cut by the flush size they are sum_i v[i] tau^i g whichever way the pairs are walked
  // -- one pipelined batch; otherwise stream by stream
  int commit_many(Vecs& V, const std::vector<uint64_t>& les, size_t wanted_flush, uint64_t* out) const {
    bool cut = false;
    std::vector<size_t> ns(les.size());
    for (size_t k = 0; k < les.size(); k++) {
      RC(vec_len(les[k], &ns[k]));
      if (ns[k] > nck) return GM_EINVAL;
      cut = cut || ns[k] > flush(wanted_flush);
    }
    if (!cut) return gm_ck_msm_batch(ck, les.data(), ns.data(), les.size(), out);
    for (size_t k = 0; k < les.size(); k++) RC(msm_le(V, les[k], wanted_flush, out + 18 * k));
    return GM_OK;
  }
};

int reversed(Vecs& V, uint64_t v, uint64_t* out) {
  size_t n = 0;
  RC(vec_len(v, &n));
  RC(V.alloc(n, out));
  return gm_fr_reverse(v, *out);
}

}  // namespace

extern "C" int gm_psnark_new_elastic(const gm_psnark_instance* I, uint64_t z_stream, uint64_t w_stream, uint64_t za_stream, uint64_t zb_stream,
                                     uint64_t zc_stream, uint64_t ck_bases, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding,
                                     size_t cap_rounds, gm_psnark_proof* P) {
  if (!I || !P || !I->index_commitments || !P->messages[0] || !P->messages[1] || !P->messages[2] || !P->fold_commitments || !P->fold_evaluations)
    return GM_EINVAL;
  const auto t_all = Clock::now();
  Vecs V;
  StreamKey K{ck_bases, 0, max_msm_buffer, min_device_chunk};
  RC(gm_ck_len(ck_bases, &K.nck));
  size_t nz = 0;
  RC(vec_len(z_stream, &nz));
  const size_t nnz = I->nnz;
  RC(gm_footprint_admit(1, ck_bases, nz, nnz, min_device_chunk > 1 ? 1 : 2));
  uint64_t one[4];
  Fr::one().to_limbs(one);
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));

  auto t0 = Clock::now();
  {
    size_t nw = 0;
    RC(vec_len(w_stream, &nw));
    if (nw > K.nck) return GM_EINVAL;
    RC(stream_msm(ck_bases, w_stream, nw, nw ? nw - 1 : 0, K.flush((size_t)1 << 20), P->witness_commitment));  // :82
  }
  P->spans[0] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));  // :86-89
  RC(gm_transcript_append_message(T.h, L("ck"), 2, I->ck_g2_bytes, I->ck_g2_len));
  RC(gm_transcript_append_g1(T.h, L("instance"), 8, I->index_commitments, 5, 1));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  {
    uint64_t zc_le;
    RC(reversed(V, zc_stream, &zc_le));
    RC(gm_fr_eval_le(zc_le, alpha, 1, P->zc_alpha));  // evaluate_be(z_c, alpha) :92-93
    V.release(zc_le);
  }
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  // min_device_chunk > 1 (the default, 2^26): the embedder lets the device merge flushes, i.e. treats max_msm_buffer as advisory
  // because everything is resident -- then the sumchecks take the resident schedule too (ElasticSc::init_resident).  min_device_chunk
  // = 1 is the LITERAL elastic prover: 2^20-pair flushes, space provers until SPACE_TIME_THRESHOLD rounds remain.
  const bool resident = min_device_chunk > 1;
  t0 = Clock::now();
  std::vector<uint64_t> ch1, ch2;
  if (resident) {
    uint64_t za_le, zb_le;
    RC(reversed(V, za_stream, &za_le));
    RC(reversed(V, zb_stream, &zb_le));
    RC(sumcheck_new_time(T.h, za_le, zb_le, alpha, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));
    V.release(za_le);
    V.release(zb_le);
  } else {
    ElasticSc S1;
    RC(S1.init(za_stream, zb_stream, alpha, false));  // Sumcheck::new_space :97
    RC(prove_one(T.h, S1, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));
  }
  P->spans[1] = since(t0);

  t0 = Clock::now();
  const size_t nt = (size_t)1 << P->rounds[0];
  if (I->ext_fre_row_len != nt + nnz || I->ext_fre_col_len != nz + nnz) return GM_EINVAL;
  uint64_t z_le, w_le;
  RC(reversed(V, z_stream, &z_le));
  RC(reversed(V, w_stream, &w_le));
  uint64_t a_ch, b_ch, c_ch;  // Tensor(r_short), powers of alpha, their product   :150-157
  RC(V.alloc(nt, &b_ch));
  RC(gm_fr_tensor(ch1.data(), P->rounds[0], b_ch));
  RC(V.alloc(nt, &c_ch));
  RC(gm_fr_powers(alpha, nt, c_ch));
  RC(V.alloc(nt, &a_ch));
  RC(gm_fr_hadamard(b_ch, c_ch, a_ch));
  P->spans[2] = since(t0);
  uint64_t ralpha_star, r_star, alpha_star, z_star;  // lookup streams :148,159-161
  RC(V.alloc(nnz, &z_star));
  RC(gm_fr_gather(z_le, I->col_index, z_star));
  RC(V.alloc(nnz, &ralpha_star));
  RC(gm_fr_gather(a_ch, I->row_index, ralpha_star));
  V.release(a_ch);
  RC(V.alloc(nnz, &r_star));
  RC(gm_fr_gather(b_ch, I->row_index, r_star));
  RC(V.alloc(nnz, &alpha_star));
  RC(gm_fr_gather(c_ch, I->row_index, alpha_star));

  t0 = Clock::now();
  {
    uint64_t four[4 * 18];
    RC(K.commit_many(V, {ralpha_star, r_star, alpha_star, z_star}, (size_t)1 << 20, four));  // :164-172
    memcpy(P->r_star_commitments, four, 3 * 144);
    memcpy(P->z_star_commitment, four + 54, 144);
  }
  P->spans[3] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("ra*"), 3, P->r_star_commitments[0], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("rb*"), 3, P->r_star_commitments[1], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("rc*"), 3, P->r_star_commitments[2], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("z*"), 2, P->z_star_commitment, 1, 0));
  uint64_t eta3[12];  // :181-192
  memcpy(eta3, one, 32);
  RC(gm_transcript_challenge_fr(T.h, L("chal"), 4, eta3 + 4));
  Fr::from_limbs(eta3 + 4).sqr().to_limbs(eta3 + 8);
  uint64_t rhs;
  {
    uint64_t h[3];
    const uint64_t lhs3[3] = {ralpha_star, r_star, alpha_star}, vals[3] = {I->val_a, I->val_b, I->val_c};
    for (int k = 0; k < 3; k++) {
      RC(V.alloc(nnz, &h[k]));
      RC(gm_fr_hadamard(lhs3[k], vals[k], h[k]));
    }
    RC(V.alloc(nnz, &rhs));
    RC(gm_fr_lincomb(h, eta3, 3, rhs));
    for (int k = 0; k < 3; k++) V.release(h[k]);
  }
  t0 = Clock::now();
  if (resident) {
    RC(sumcheck_new_time(T.h, z_star, rhs, one, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));
  } else {
    uint64_t zs, rs;
    RC(reversed(V, z_star, &zs));
    RC(reversed(V, rhs, &rs));
    ElasticSc S2;
    RC(S2.init(zs, rs, one, true));  // Sumcheck::new_elastic :195
    RC(prove_one(T.h, S2, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));
    S2.reset();
    V.release(zs);
    V.release(rs);
  }
  V.release(rhs);
  P->spans[4] = since(t0);

  uint64_t zeta[4];
  RC(gm_transcript_challenge_fr(T.h, L("zeta"), 4, zeta));  // :199
  t0 = Clock::now();
  uint64_t ahp[3], sorted[3];  // :205-214
  RC(V.alloc(nt, &ahp[0]));
  RC(gm_fr_alg_hash(b_ch, 0, zeta, ahp[0]));
  RC(V.alloc(nt, &ahp[1]));
  RC(gm_fr_alg_hash(c_ch, 0, zeta, ahp[1]));
  RC(V.alloc(nz, &ahp[2]));
  RC(gm_fr_alg_hash(z_le, 0, zeta, ahp[2]));
  RC(V.alloc(I->ext_fre_row_len, &sorted[0]));
  RC(gm_fr_gather(ahp[0], I->ext_fre_row, sorted[0]));
  RC(V.alloc(I->ext_fre_row_len, &sorted[1]));
  RC(gm_fr_gather(ahp[1], I->ext_fre_row, sorted[1]));
  RC(V.alloc(I->ext_fre_col_len, &sorted[2]));
  RC(gm_fr_gather(ahp[2], I->ext_fre_col, sorted[2]));
  for (int k = 0; k < 3; k++) V.release(ahp[k]);
  RC(K.commit_many(V, {sorted[0], sorted[1], sorted[2]}, (size_t)1 << 20, &P->sorted_commitments[0][0]));
  P->spans[5] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("sorted_alpha_commitment"), 23, P->sorted_commitments[1], 1, 0));  // :220-222
  RC(gm_transcript_append_g1(T.h, L("sorted_r_commitment"), 19, P->sorted_commitments[0], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("sorted_z_commitment"), 19, P->sorted_commitments[2], 1, 0));
  uint64_t gamma[4], chi[4];
  RC(gm_transcript_challenge_fr(T.h, L("gamma"), 5, gamma));
  RC(gm_transcript_challenge_fr(T.h, L("chi"), 3, chi));

  t0 = Clock::now();
  uint64_t pls[9], accs[9], shifts[9];  // plookup streams, ProductStream, RightRotationStreamer   :227-243
  RC(plookup(V, r_star, b_ch, I->row_index, nnz, I->ext_fre_row, I->ext_fre_row_len, gamma, chi, zeta, pls));
  RC(plookup(V, alpha_star, c_ch, I->row_index, nnz, I->ext_fre_row, I->ext_fre_row_len, gamma, chi, zeta, pls + 3));
  RC(plookup(V, z_star, z_le, I->col_index, nnz, I->ext_fre_col, I->ext_fre_col_len, gamma, chi, zeta, pls + 6));
  for (int k = 0; k < 9; k++) {
    size_t l = 0;
    RC(vec_len(pls[k], &l));
    RC(V.alloc(l + 1, &accs[k]));
    RC(gm_fr_acc_product(pls[k], accs[k]));
    RC(gm_fr_vec_download(accs[k], 0, P->products[k], 1));
    RC(V.alloc(l + 1, &shifts[k]));
    RC(gm_fr_shift_monic(pls[k], shifts[k]));
    V.release(pls[k]);
  }
  V.release(b_ch);
  V.release(c_ch);
  V.release(z_le);
  P->spans[6] = since(t0);
  RC(gm_transcript_append_fr(T.h, L("set_r_ep"), 8, P->products[3], 1));  // :245-250 (labels as in the reference)
  RC(gm_transcript_append_fr(T.h, L("subset_r_ep"), 11, P->products[4], 1));
  RC(gm_transcript_append_fr(T.h, L("set_r_ep"), 8, P->products[0], 1));
  RC(gm_transcript_append_fr(T.h, L("subset_r_ep"), 11, P->products[1], 1));
  RC(gm_transcript_append_fr(T.h, L("set_z_ep"), 8, P->products[6], 1));
  RC(gm_transcript_append_fr(T.h, L("subset_z_ep"), 11, P->products[7], 1));
  if (((size_t)1 << P->rounds[1]) < nnz) return GM_EINVAL;
  uint64_t ep_r;  // Tensor(&sumcheck2.challenges), cut to the looked-up length   :254-257
  RC(V.alloc((size_t)1 << P->rounds[1], &ep_r));
  RC(gm_fr_tensor(ch2.data(), P->rounds[1], ep_r));
  RC(gm_fr_vec_set_len(ep_r, nnz));

  // EntryProduct::new_elastic_batch (entryproduct/elastic_prover.rs:66-127)
  t0 = Clock::now();
  std::vector<ElasticSc> provers(13);
  std::vector<uint64_t> batch_streams;  // reversed streams the 13 provers read in place
  uint64_t psi[4];
  {
    RC(K.commit_many(V, std::vector<uint64_t>(accs, accs + 9), (size_t)1 << 20, &P->acc_v_commitments[0][0]));
    for (int k = 0; k < 9; k++) RC(gm_transcript_append_g1(T.h, L("acc_v"), 5, P->acc_v_commitments[k], 1, 0));
    RC(gm_transcript_challenge_fr(T.h, L("ep-chal"), 7, psi));
    uint64_t acc_chal[9][4];
    RC(gm_fr_eval_le_batch(accs, 9, psi, 1, &acc_chal[0][0]));  // evaluate_be of the streams
    const Fr ci = Fr::from_limbs(psi);
    for (int k = 0; k < 9; k++) {
      size_t l = 0;
      RC(vec_len(accs[k], &l));
      (Fr::from_limbs(acc_chal[k]) * ci + Fr::from_limbs(P->products[k]) - fr_pow(ci, l)).to_limbs(P->claimed_sumchecks[k]);
      if (resident) {
        RC(provers[k].init_resident(accs[k], shifts[k], psi));
        continue;
      }
      uint64_t as, ss;
      RC(reversed(V, accs[k], &as));
      RC(reversed(V, shifts[k], &ss));
      RC(provers[k].init(as, ss, psi, true));  // the provers read their streams in place: released after the batch
      batch_streams.push_back(as);
      batch_streams.push_back(ss);
    }
  }
  P->spans[7] = since(t0);

  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));  // :313-330
  t0 = Clock::now();
  {
    std::vector<uint64_t> polys = {ralpha_star};
    polys.insert(polys.end(), accs, accs + 9);
    std::vector<uint64_t> oc(40);
    Fr acc = Fr::one();
    size_t longest = 0;
    for (int k = 0; k < 10; k++) {
      acc.to_limbs(oc.data() + 4 * k);
      acc = acc * Fr::from_limbs(open_chal);
      size_t l = 0;
      RC(vec_len(polys[k], &l));
      longest = std::max(longest, l);
    }
    uint64_t poly, q;
    RC(V.alloc(longest, &poly));
    RC(gm_fr_lincomb(polys.data(), oc.data(), 10, poly));
    size_t lp = 0;
    RC(vec_len(poly, &lp));
    // CommitterKeyStream::open (space.rs:95-125): the quotient by (x - psi), committed as a stream
    RC(V.alloc(lp ? lp - 1 : 0, &q));
    uint64_t rem[4];
    RC(gm_fr_div_vanishing(poly, psi, 1, q, rem));
    RC(K.msm_le(V, q, max_msm_buffer, P->ralpha_star_acc_mu_proof));
    V.release(q);
    V.release(poly);
    RC(gm_fr_eval_le_batch(polys.data(), 10, psi, 1, &P->ralpha_star_acc_mu_evals[0][0]));  // :332-343
  }
  P->spans[8] = since(t0);
  {
    const uint64_t lhs3[3] = {ralpha_star, r_star, alpha_star}, vals[3] = {I->val_a, I->val_b, I->val_c};
    uint64_t lh[3];
    for (int k = 0; k < 3; k++) {
      RC(V.alloc(nnz, &lh[k]));
      RC(gm_fr_hadamard(lhs3[k], ep_r, lh[k]));
    }
    RC(gm_fr_ip(lh[0], I->val_a, P->rstars_vals[0]));  // :348-349
    RC(gm_fr_ip(lh[1], I->val_b, P->rstars_vals[1]));
    for (int k = 0; k < 10; k++) RC(gm_transcript_append_fr(T.h, L("ralpha_star_acc_mu"), 18, P->ralpha_star_acc_mu_evals[k], 1));
    RC(gm_transcript_append_g1(T.h, L("ralpha_star_mu_proof"), 20, P->ralpha_star_acc_mu_proof, 1, 0));
    for (int k = 0; k < 3; k++) {  // :358-377
      if (resident) {
        RC(provers[9 + k].init_resident(lh[k], vals[k], one));
        batch_streams.push_back(lh[k]);  // read in place until the first fold: released after the batch
        continue;
      }
      uint64_t ls, vs;
      RC(reversed(V, lh[k], &ls));
      RC(reversed(V, vals[k], &vs));
      RC(provers[9 + k].init(ls, vs, one, true));
      batch_streams.push_back(ls);
      batch_streams.push_back(vs);
      V.release(lh[k]);
    }
    if (resident) {
      RC(provers[12].init_resident(r_star, alpha_star, psi));
    } else {
      uint64_t rs, as;
      RC(reversed(V, r_star, &rs));
      RC(reversed(V, alpha_star, &as));
      RC(provers[12].init(rs, as, psi, true));
      batch_streams.push_back(rs);
      batch_streams.push_back(as);
    }
  }
  V.release(ep_r);
  t0 = Clock::now();
  std::vector<uint64_t> ch3(cap_rounds * 4, 0);
  RC(prove_batch(T.h, provers, P->messages[2], ch3.data(), cap_rounds, &P->third_final_foldings[0][0], &P->rounds[2]));  // :380
  provers.clear();
  for (uint64_t v : batch_streams) V.release(v);
  P->spans[9] = since(t0);

  // ---- tensorcheck (:384-600)
  t0 = Clock::now();
  const size_t n3 = P->rounds[2], n2 = P->rounds[1];
  uint64_t tc_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, tc_chal));
  std::vector<uint64_t> tcc(4 * 13);
  {
    Fr acc = Fr::one();
    for (int k = 0; k < 13; k++) {
      acc.to_limbs(tcc.data() + 4 * k);
      acc = acc * Fr::from_limbs(tc_chal);
    }
  }
  struct Tree {
    uint64_t body;
    std::vector<uint64_t> challenges;  // every challenge but the last
    std::vector<uint64_t> levels;
  };
  std::vector<Tree> trees(4);
  {
    auto lincomb = [&](std::vector<uint64_t> polys, uint64_t* out) -> int {
      size_t longest = 0;
      for (uint64_t p : polys) {
        size_t l = 0;
        RC(vec_len(p, &l));
        longest = std::max(longest, l);
      }
      RC(V.alloc(longest, out));
      return gm_fr_lincomb(polys.data(), tcc.data(), polys.size(), *out);
    };
    std::vector<uint64_t> g0(accs, accs + 9), g1(shifts, shifts + 9);
    g0.push_back(r_star);
    g1.insert(g1.end(), {I->val_a, I->val_b, I->val_c, alpha_star});
    RC(lincomb(g0, &trees[0].body));
    RC(lincomb(g1, &trees[1].body));
    trees[2].body = z_star;
    RC(lincomb({ralpha_star, r_star, alpha_star}, &trees[3].body));
    if (n3 == 0 || n2 == 0) return GM_EINVAL;
    Fr tw = Fr::from_limbs(psi);
    trees[0].challenges.resize(4 * (n3 - 1));
    for (size_t j = 0; j + 1 < n3; j++) {
      (Fr::from_limbs(ch3.data() + 4 * j) * tw).to_limbs(trees[0].challenges.data() + 4 * j);
      tw = tw.sqr();
    }
    trees[1].challenges.assign(ch3.begin(), ch3.begin() + 4 * (n3 - 1));
    trees[2].challenges.assign(ch2.begin(), ch2.begin() + 4 * (n2 - 1));
    const size_t nh = std::min(n2, n3);  // zip(ch2, ch3[:len(ch2)]) without its last element
    trees[3].challenges.resize(4 * (nh - 1));
    for (size_t j = 0; j + 1 < nh; j++)
      (Fr::from_limbs(ch2.data() + 4 * j) * Fr::from_limbs(ch3.data() + 4 * j)).to_limbs(trees[3].challenges.data() + 4 * j);
  }
  for (int k = 0; k < 9; k++) V.release(shifts[k]);
  size_t nfold = 0;
  for (auto& t : trees) {  // FoldedPolynomialTree: the levels, little-endian
    size_t len = 0;
    RC(vec_len(t.body, &len));
    for (size_t k = 0; k < t.challenges.size() / 4; k++) {
      uint64_t nxt;
      len = (len + 1) / 2;
      RC(V.alloc(len, &nxt));
      t.levels.push_back(nxt);
    }
    RC(gm_fr_fold_chain(t.body, t.challenges.data(), t.levels.size(), t.levels.data()));
    nfold += t.levels.size();
  }
  P->nfold = nfold;
  if (nfold > P->cap_folds) return GM_EINVAL;
  std::vector<uint64_t> all_levels;
  for (auto& t : trees) all_levels.insert(all_levels.end(), t.levels.begin(), t.levels.end());
  {
    // commit_folding (space.rs:192-223): one ChunkedPippenger of max_msm_buffer / depth per level of a tree.  When no level of any
    // tree is cut by its flush size every commitment is sum_i level[i] tau^i g whichever way the pairs are walked: the levels of
    // all four trees go through ONE pipelined batch; otherwise tree by tree, level by level, as streams
    bool cut = false;
    for (auto& t : trees)
      for (uint64_t lv : t.levels) {
        size_t l = 0;
        RC(vec_len(lv, &l));
        cut = cut || l > K.flush(max_msm_buffer / t.levels.size());
      }
    if (!cut) {
      if (nfold) RC(K.commit_many(V, all_levels, (size_t)1 << 62, P->fold_commitments));
    } else {
      size_t at = 0;
      for (auto& t : trees) {
        const size_t depth = t.levels.size();
        if (depth) RC(K.commit_many(V, t.levels, max_msm_buffer / depth, P->fold_commitments + 18 * at));
        at += depth;
      }
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
  RC(gm_fr_eval_le_batch(all_levels.data(), nfold, pts + 4, 2, P->fold_evaluations));  // evaluate_folding, tree by tree
  std::vector<uint64_t> base = {w_le, ralpha_star, r_star, alpha_star, z_star, I->row, I->col, I->val_a, I->val_b, I->val_c, sorted[0], sorted[1], sorted[2]};
  base.insert(base.end(), accs, accs + 9);
  RC(gm_fr_eval_le_batch(base.data(), base.size(), pts, 3, &P->base_evaluations[0][0]));
  for (size_t k = 0; k < 3 * base.size(); k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, &P->base_evaluations[0][0] + 4 * k, 1));
  for (size_t k = 0; k < 2 * nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal2[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal2));
  {
    const Fr oc = Fr::from_limbs(open_chal2);
    std::vector<uint64_t> ocs(4 * (base.size() + nfold));
    Fr acc = Fr::one();
    for (size_t k = 0; k < base.size() + nfold; k++) {
      acc.to_limbs(ocs.data() + 4 * k);
      acc = acc * oc;
    }
    // The reference sums five openings: open_multi_points(partial_eval) (space.rs:128-166) and one open_folding per tree (:229-285),
    // each sum_i eta_i commit(p_i div Z), the HashMapPippenger of open_folding merging the scalars of equal bases.  Division by the
    // same Z is linear and all five walk the same key, so the merge extends over all of them: ONE linear combination of the 22 base
    // polynomials and every level, ONE division, one stream MSM (flushed every max_msm_buffer pairs) -- instead of a division
    // per level (~100 latency-bound three-phase scans)
    std::vector<uint64_t> all = base;
    all.insert(all.end(), all_levels.begin(), all_levels.end());
    size_t longest = 0;
    for (uint64_t p : all) {
      size_t l = 0;
      RC(vec_len(p, &l));
      longest = std::max(longest, l);
    }
    uint64_t pe, q, rem[12];
    RC(V.alloc(longest, &pe));
    RC(gm_fr_lincomb(all.data(), ocs.data(), all.size(), pe));
    size_t lp = 0;
    RC(vec_len(pe, &lp));
    RC(V.alloc(lp ? lp - 1 : 0, &q));
    RC(gm_fr_div_vanishing(pe, pts, 3, q, rem));
    V.release(pe);
    RC(K.msm_le(V, q, max_msm_buffer, P->evaluation_proof));
    V.release(q);
  }
  P->spans[10] = since(t0);
  P->spans[11] = since(t_all);
  return GM_OK;
}

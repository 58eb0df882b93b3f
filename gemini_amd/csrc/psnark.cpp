// psnark::Proof::new_time (src/psnark/time_prover.rs:69-384) with EntryProduct::new_time_batch
// (src/subprotocols/entryproduct/time_prover.rs:53-114), the plookup vector builders
// (src/subprotocols/plookup/time_prover.rs:89-112), Sumcheck::{new_time, prove_batch} (sumcheck/proof.rs:69-130),
// TensorcheckProof::new_time (tensorcheck/mod.rs:190-275) and CommitterKey::{commit, batch_commit,
// batch_open_multi_points} (src/kzg/time.rs:81-159) as ONE entry point of the library.
//
// Pure orchestration over the library's own C ABI, like snark.cpp: every O(n) step is a gm_* call, the sequence is the
// one of gemini_amd/psnark.py::Proof.new_time (the tests hold the two byte for byte equal).  Driven from Python the
// prover makes ~1000 FFI round trips per proof, most of which wait for the device; here the GPU is fed from one thread
// that never leaves the library.
#include <algorithm>

#include "prover_common.hpp"

namespace {

using namespace gmprover;

}  // namespace

extern "C" int gm_psnark_new_time(const gm_psnark_instance* I, uint64_t ck_bases, int g1_encoding, size_t cap_rounds, gm_psnark_proof* P) {
  if (!I || !P || !I->index_commitments || !P->messages[0] || !P->messages[1] || !P->messages[2] || !P->fold_commitments || !P->fold_evaluations)
    return GM_EINVAL;
  const auto t_all = Clock::now();
  Vecs V;
  size_t nz = 0, nck = 0;
  RC(vec_len(I->z, &nz));
  RC(gm_ck_len(ck_bases, &nck));
  const size_t nnz = I->nnz;
  if (nck < nnz || nck < I->ext_fre_row_len || nck < I->ext_fre_col_len) return GM_EINVAL;  // index_by zips need as many powers as indices (:119-127,179-183)
  RC(gm_footprint_admit(1, ck_bases, nz, nnz, 0));  // room for the whole proof, or GM_ENOMEM with the numbers, before the first allocation
  uint64_t one[4];
  Fr::one().to_limbs(one);

  // z_a, z_b, z_c (:74-76)
  uint64_t z_abc[3];
  const uint64_t mats[3] = {I->a, I->b, I->c};
  for (int k = 0; k < 3; k++) {
    size_t rows = 0, cols = 0;
    RC(gm_spm_shape(mats[k], &rows, &cols, nullptr));
    if (cols != nz) return GM_EINVAL;
    RC(V.alloc(rows, &z_abc[k]));
    RC(gm_spm_mul(mats[k], I->z, z_abc[k]));
  }
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));

  auto t0 = Clock::now();
  RC(commit(ck_bases, nck, I->w, P->witness_commitment));  // :79
  P->spans[0] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));  // :82-86
  RC(gm_transcript_append_message(T.h, L("ck"), 2, I->ck_g2_bytes, I->ck_g2_len));
  RC(gm_transcript_append_g1(T.h, L("instance"), 8, I->index_commitments, 5, 1));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  RC(gm_fr_eval_le(z_abc[2], alpha, 1, P->zc_alpha));  // :88-89
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1, ch2;
  RC(sumcheck_new_time(T.h, z_abc[0], z_abc[1], alpha, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));  // :92
  P->spans[1] = since(t0);
  // Vectors are released at their LAST USE, not at the end of a phase: at 2^26 constraints every vector of n elements is 2 GiB and
  // the third sumcheck holds ~70 of them (gm_psnark_footprint); round 4 held ~93 there and the proof peaked at 308 GB of a 309 GB device
  for (int k = 0; k < 3; k++) V.release(z_abc[k]);

  t0 = Clock::now();
  const size_t nt = (size_t)1 << P->rounds[0];
  // extend_frequency(compute_frequency(set_len, index)) has set_len + |index| entries (plookup/time_prover.rs:66-79)
  if (I->ext_fre_row_len != nt + nnz || I->ext_fre_col_len != nz + nnz) return GM_EINVAL;
  uint64_t a_ch, b_ch, c_ch;
  RC(V.alloc(nt, &b_ch));
  RC(gm_fr_tensor(ch1.data(), P->rounds[0], b_ch));  // :95-97
  RC(V.alloc(nt, &c_ch));
  RC(gm_fr_powers(alpha, nt, c_ch));
  RC(V.alloc(nt, &a_ch));
  RC(gm_fr_hadamard(b_ch, c_ch, a_ch));
  P->spans[2] = since(t0);  // "joint matrices": resident with the instance

  uint64_t ralpha_star, r_star, alpha_star, z_star;  // :114-117
  RC(V.alloc(nnz, &ralpha_star));
  RC(gm_fr_gather(a_ch, I->row_index, ralpha_star));
  V.release(a_ch);
  RC(V.alloc(nnz, &r_star));
  RC(gm_fr_gather(b_ch, I->row_index, r_star));
  RC(V.alloc(nnz, &alpha_star));
  RC(gm_fr_gather(c_ch, I->row_index, alpha_star));
  RC(V.alloc(nnz, &z_star));
  RC(gm_fr_gather(I->z, I->col_index, z_star));

  t0 = Clock::now();
  RC(batch_commit(ck_bases, nck, {ralpha_star, r_star, alpha_star}, &P->r_star_commitments[0][0]));  // :119-127
  RC(commit(ck_bases, nck, z_star, P->z_star_commitment));
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
  {
    uint64_t h[3];
    const uint64_t lhs[3] = {ralpha_star, r_star, alpha_star}, rhs[3] = {I->val_a, I->val_b, I->val_c};
    for (int k = 0; k < 3; k++) {
      RC(V.alloc(nnz, &h[k]));
      RC(gm_fr_hadamard(lhs[k], rhs[k], h[k]));
    }
    RC(V.alloc(nnz, &r_star_val));
    RC(gm_fr_lincomb(h, eta3, 3, r_star_val));  // :137-144
    for (int k = 0; k < 3; k++) V.release(h[k]);
  }

  t0 = Clock::now();
  RC(sumcheck_new_time(T.h, z_star, r_star_val, one, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :147-152
  uint64_t second_challenges;
  RC(V.alloc((size_t)1 << P->rounds[1], &second_challenges));
  RC(gm_fr_tensor(ch2.data(), P->rounds[1], second_challenges));
  if (((size_t)1 << P->rounds[1]) < nnz) return GM_EINVAL;
  RC(gm_fr_vec_set_len(second_challenges, nnz));  // &second_challenges[..num_non_zero]
  V.release(r_star_val);
  P->spans[4] = since(t0);

  uint64_t zeta[4];
  RC(gm_transcript_challenge_fr(T.h, L("zeta"), 4, zeta));  // :157

  t0 = Clock::now();
  uint64_t ahp[3];  // alg_hash of b_challenges, c_challenges, z   :160-164
  RC(V.alloc(nt, &ahp[0]));
  RC(gm_fr_alg_hash(b_ch, 0, zeta, ahp[0]));
  RC(V.alloc(nt, &ahp[1]));
  RC(gm_fr_alg_hash(c_ch, 0, zeta, ahp[1]));
  RC(V.alloc(nz, &ahp[2]));
  RC(gm_fr_alg_hash(I->z, 0, zeta, ahp[2]));
  uint64_t sorted[3];  // :169-173
  RC(V.alloc(I->ext_fre_row_len, &sorted[0]));
  RC(gm_fr_gather(ahp[0], I->ext_fre_row, sorted[0]));
  RC(V.alloc(I->ext_fre_row_len, &sorted[1]));
  RC(gm_fr_gather(ahp[1], I->ext_fre_row, sorted[1]));
  RC(V.alloc(I->ext_fre_col_len, &sorted[2]));
  RC(gm_fr_gather(ahp[2], I->ext_fre_col, sorted[2]));
  for (int k = 0; k < 3; k++) V.release(ahp[k]);
  RC(batch_commit(ck_bases, nck, {sorted[0], sorted[1], sorted[2]}, &P->sorted_commitments[0][0]));  // :179-183
  P->spans[5] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("sorted_alpha_commitment"), 23, P->sorted_commitments[1], 1, 0));  // :186-188
  RC(gm_transcript_append_g1(T.h, L("sorted_r_commitment"), 19, P->sorted_commitments[0], 1, 0));
  RC(gm_transcript_append_g1(T.h, L("sorted_z_commitment"), 19, P->sorted_commitments[2], 1, 0));
  uint64_t gamma[4], chi[4];
  RC(gm_transcript_challenge_fr(T.h, L("gamma"), 5, gamma));  // :190-191
  RC(gm_transcript_challenge_fr(T.h, L("chi"), 3, chi));

  t0 = Clock::now();
  uint64_t lookup_vec[9];  // r: set, subset, sorted; alpha: ..; z: ..   :194-209
  RC(plookup(V, r_star, b_ch, I->row_index, nnz, I->ext_fre_row, I->ext_fre_row_len, gamma, chi, zeta, lookup_vec));
  RC(plookup(V, alpha_star, c_ch, I->row_index, nnz, I->ext_fre_row, I->ext_fre_row_len, gamma, chi, zeta, lookup_vec + 3));
  RC(plookup(V, z_star, I->z, I->col_index, nnz, I->ext_fre_col, I->ext_fre_col_len, gamma, chi, zeta, lookup_vec + 6));
  V.release(b_ch);
  V.release(c_ch);
  uint64_t acc_vec[9];  // accumulated_product(monic(v))   :211-214
  // the right rotation of every lookup vector (shift_monic) is needed twice -- as the g side of the entry-product sumchecks (:223-239)
  // and as a body of the tensor check (:323-326) -- and the lookup vectors themselves for nothing else: build it here, once, and let
  // the 12 n elements of the lookup vectors go
  std::vector<uint64_t> shift_lookup(9);
  for (int k = 0; k < 9; k++) {
    size_t l = 0;
    RC(vec_len(lookup_vec[k], &l));
    RC(V.alloc(l + 1, &acc_vec[k]));
    RC(gm_fr_acc_product(lookup_vec[k], acc_vec[k]));
    RC(gm_fr_vec_download(acc_vec[k], 0, P->products[k], 1));  // the full product is the first accumulated entry
    RC(V.alloc(l + 1, &shift_lookup[k]));
    RC(gm_fr_shift_monic(lookup_vec[k], shift_lookup[k]));
    V.release(lookup_vec[k]);
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
  std::vector<uint64_t> provers, borrowed_tmp;
  struct ProverGuard {
    std::vector<uint64_t>& p;
    ~ProverGuard() {
      for (uint64_t h : p) (void)gm_sc_free(h);
    }
  } prover_guard{provers};
  uint64_t psi[4];
  {
    RC(batch_commit(ck_bases, nck, std::vector<uint64_t>(acc_vec, acc_vec + 9), &P->acc_v_commitments[0][0]));
    for (int k = 0; k < 9; k++) RC(gm_transcript_append_g1(T.h, L("acc_v"), 5, P->acc_v_commitments[k], 1, 0));
    RC(gm_transcript_challenge_fr(T.h, L("ep-chal"), 7, psi));
    for (int k = 0; k < 9; k++) {
      uint64_t h = 0;
      RC(gm_sc_new_borrow(acc_vec[k], shift_lookup[k], psi, &h));  // both read in place until the first fold, never written
      provers.push_back(h);
    }
    uint64_t acc_chal[9][4];
    RC(gm_fr_eval_le_batch(acc_vec, 9, psi, 1, &acc_chal[0][0]));
    const Fr ci = Fr::from_limbs(psi);
    for (int k = 0; k < 9; k++) {
      size_t l = 0;
      RC(vec_len(acc_vec[k], &l));
      (Fr::from_limbs(acc_chal[k]) * ci + Fr::from_limbs(P->products[k]) - fr_pow(ci, l)).to_limbs(P->claimed_sumchecks[k]);
    }
  }
  P->spans[7] = since(t0);

  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));  // :241-242
  t0 = Clock::now();
  std::vector<uint64_t> polys = {ralpha_star};  // :244-251
  polys.insert(polys.end(), acc_vec, acc_vec + 9);
  RC(batch_open(V, ck_bases, nck, polys, psi, 1, open_chal, P->ralpha_star_acc_mu_proof));
  RC(gm_fr_eval_le_batch(polys.data(), 10, psi, 1, &P->ralpha_star_acc_mu_evals[0][0]));
  P->spans[8] = since(t0);
  {
    uint64_t h_a, h_b;  // :253-254
    RC(V.alloc(nnz, &h_a));
    RC(gm_fr_hadamard(ralpha_star, I->val_a, h_a));
    RC(V.alloc(nnz, &h_b));
    RC(gm_fr_hadamard(r_star, I->val_b, h_b));
    RC(gm_fr_ip(h_a, second_challenges, P->rstars_vals[0]));
    RC(gm_fr_ip(h_b, second_challenges, P->rstars_vals[1]));
    V.release(h_a);
    V.release(h_b);
  }
  for (int k = 0; k < 10; k++) RC(gm_transcript_append_fr(T.h, L("ralpha_star_acc_mu"), 18, P->ralpha_star_acc_mu_evals[k], 1));  // :258-261
  RC(gm_transcript_append_g1(T.h, L("ralpha_star_mu_proof"), 20, P->ralpha_star_acc_mu_proof, 1, 0));
  {
    const uint64_t lhs[3] = {ralpha_star, r_star, alpha_star}, rhs[3] = {I->val_a, I->val_b, I->val_c};  // :263-290
    for (int k = 0; k < 3; k++) {
      uint64_t h, pr = 0;
      RC(V.alloc(nnz, &h));
      RC(gm_fr_hadamard(lhs[k], second_challenges, h));
      RC(gm_sc_new_borrow(h, rhs[k], one, &pr));
      provers.push_back(pr);
      borrowed_tmp.push_back(h);
    }
    uint64_t pr = 0;
    RC(gm_sc_new_borrow(r_star, alpha_star, psi, &pr));
    provers.push_back(pr);
  }
  V.release(second_challenges);
  t0 = Clock::now();
  std::vector<uint64_t> ch3(cap_rounds * 4, 0);
  RC(gm_sumcheck_prove_batch(T.h, provers.data(), provers.size(), P->messages[2], ch3.data(), cap_rounds, &P->third_final_foldings[0][0], &P->rounds[2]));  // :293
  for (uint64_t h : provers) (void)gm_sc_free(h);
  provers.clear();
  for (uint64_t v : borrowed_tmp) V.release(v);
  P->spans[9] = since(t0);

  // ---- TensorcheckProof::new_time(transcript, ck, 22 base polynomials, 4 bodies)   :296-367, tensorcheck/mod.rs:190-275
  t0 = Clock::now();
  std::vector<uint64_t> base = {I->w, ralpha_star, r_star, alpha_star, z_star, I->row, I->col, I->val_a, I->val_b, I->val_c, sorted[0], sorted[1], sorted[2]};
  base.insert(base.end(), acc_vec, acc_vec + 9);
  const size_t n3 = P->rounds[2], n2 = P->rounds[1];
  struct Body {
    std::vector<uint64_t> polys;
    std::vector<uint64_t> challenges;  // 4 limbs each
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
    bodies[1].polys = shift_lookup;  // shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star], challenges third_ch
    bodies[1].polys.insert(bodies[1].polys.end(), {I->val_a, I->val_b, I->val_c, alpha_star});
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
  for (auto& b : bodies) max_group = b.polys.size() > max_group ? b.polys.size() : max_group;
  std::vector<uint64_t> bc(4 * max_group);  // powers(batch_challenge, max_len)
  {
    Fr acc = Fr::one();
    const Fr c = Fr::from_limbs(batch_challenge);
    for (size_t k = 0; k < max_group; k++) {
      acc.to_limbs(bc.data() + 4 * k);
      acc = acc * c;
    }
  }
  std::vector<uint64_t> foldings;
  for (auto& b : bodies) {
    size_t longest = 0;
    for (uint64_t p : b.polys) {
      size_t l = 0;
      RC(vec_len(p, &l));
      longest = l > longest ? l : longest;
    }
    uint64_t batched;
    RC(V.alloc(longest, &batched));
    RC(gm_fr_lincomb(b.polys.data(), bc.data(), b.polys.size(), batched));
    size_t len = 0;
    RC(vec_len(batched, &len));
    const size_t nch = b.challenges.size() / 4, first_level = foldings.size();
    for (size_t k = 0; k + 1 < nch; k++) {  // foldings_polynomial: all challenges but the last (:124-133)
      uint64_t nxt;
      len = (len + 1) / 2;
      RC(V.alloc(len, &nxt));
      foldings.push_back(nxt);
    }
    RC(gm_fr_fold_chain(batched, b.challenges.data(), foldings.size() - first_level, foldings.data() + first_level));  // one wait per tree
  }
  P->nfold = foldings.size();
  if (P->nfold > P->cap_folds) return GM_EINVAL;
  if (P->nfold) RC(batch_commit(ck_bases, nck, foldings, P->fold_commitments));
  for (size_t k = 0; k < P->nfold; k++) RC(gm_transcript_append_g1(T.h, L("commitment"), 10, P->fold_commitments + 18 * k, 1, 0));
  uint64_t pts[12];  // beta^2, beta, -beta
  RC(gm_transcript_challenge_fr(T.h, L("evaluation-chal"), 15, pts + 4));
  {
    const Fr beta = Fr::from_limbs(pts + 4);
    beta.sqr().to_limbs(pts);
    beta.neg().to_limbs(pts + 8);
  }
  RC(gm_fr_eval_le_batch(base.data(), base.size(), pts, 3, &P->base_evaluations[0][0]));
  RC(gm_fr_eval_le_batch(foldings.data(), P->nfold, pts + 4, 2, P->fold_evaluations));
  for (size_t k = 0; k < 3 * base.size(); k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, &P->base_evaluations[0][0] + 4 * k, 1));
  for (size_t k = 0; k < 2 * P->nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal2[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal2));
  {
    std::vector<uint64_t> all = base;
    all.insert(all.end(), foldings.begin(), foldings.end());
    RC(batch_open(V, ck_bases, nck, all, pts, 3, open_chal2, P->evaluation_proof));
  }
  P->spans[10] = since(t0);
  P->spans[11] = since(t_all);
  return GM_OK;
}

// ---- preprocessing: the matrix-only part of the instance (gemini_amd/psnark.py::joint_matrices, _JointDevice) -------------
extern "C" int gm_psnark_preprocess(uint64_t a, uint64_t b, uint64_t c, size_t num_variables, gm_psnark_instance* I) {
  if (!I) return GM_EINVAL;
  const uint64_t mats[3] = {a, b, c};
  size_t nrows[3], ncols[3], nnzs[3];
  for (int k = 0; k < 3; k++) RC(gm_spm_shape(mats[k], &nrows[k], &ncols[k], &nnzs[k]));
  // as the reference and the Python mirror: keys are col * |rows of A| + row (src/misc.rs:269-366 takes num_constraints from the
  // instance; the three matrices of an R1CS have the same number of rows -- anything else is rejected here)
  if (nrows[1] != nrows[0] || nrows[2] != nrows[0]) return GM_EINVAL;
  const size_t num_constraints = nrows[0];
  for (int k = 0; k < 3; k++)
    if (ncols[k] > num_variables) return GM_EINVAL;
  // row / column indices are 32-bit index vectors (gm_idx_*), keys col * num_constraints + row are 64-bit words: an instance
  // past either limit is refused, not truncated into a wrong joint support
  if (num_constraints > 0xffffffffull || num_variables > 0xffffffffull) return GM_EINVAL;
  if (num_constraints && num_variables > UINT64_MAX / num_constraints) return GM_EINVAL;
  // keys col * num_constraints + row of every entry, per DISTINCT matrix (dummy_r1cs registers one matrix three times)
  struct Host {
    std::vector<uint64_t> rowptr, keys, vals;
    std::vector<uint32_t> cols;
    std::vector<size_t> order;  // entries sorted by key, the LAST of equal keys kept
  };
  Host H[3];
  int src_of[3] = {0, 1, 2};
  if (b == a) src_of[1] = 0;
  if (c == a) src_of[2] = 0;
  else if (c == b) src_of[2] = src_of[1];
  std::vector<uint64_t> all;
  for (int k = 0; k < 3; k++) {
    if (src_of[k] != k) continue;
    Host& h = H[k];
    h.rowptr.resize(nrows[k] + 1);
    h.cols.resize(nnzs[k]);
    h.vals.resize(4 * nnzs[k]);
    RC(gm_spm_download(mats[k], h.rowptr.data(), h.cols.data(), h.vals.data()));
    h.keys.resize(nnzs[k]);
    for (size_t r = 0; r < nrows[k]; r++)
      for (uint64_t e = h.rowptr[r]; e < h.rowptr[r + 1]; e++) h.keys[e] = (uint64_t)h.cols[e] * num_constraints + r;
    std::vector<size_t> ord(nnzs[k]);
    for (size_t e = 0; e < nnzs[k]; e++) ord[e] = e;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return h.keys[x] < h.keys[y]; });
    for (size_t t = 0; t < ord.size(); t++)
      if (t + 1 == ord.size() || h.keys[ord[t + 1]] != h.keys[ord[t]]) h.order.push_back(ord[t]);
    all.insert(all.end(), h.keys.begin(), h.keys.end());
  }
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  const size_t nnz = all.size();
  std::vector<uint32_t> row_index(nnz), col_index(nnz);
  for (size_t t = 0; t < nnz; t++) {
    row_index[t] = (uint32_t)(all[t] % num_constraints);
    col_index[t] = (uint32_t)(all[t] / num_constraints);
  }
  memset(I, 0, sizeof *I);
  I->a = a;
  I->b = b;
  I->c = c;
  I->nnz = nnz;
  int rc = GM_OK;
  auto fail = [&](int code) {
    (void)gm_psnark_preprocess_free(I);
    return code;
  };
  if ((rc = gm_idx_register(row_index.data(), nnz, &I->row_index))) return fail(rc);
  if ((rc = gm_idx_register(col_index.data(), nnz, &I->col_index))) return fail(rc);
  // row / col as field vectors: F::from(index) = 0 + 1 * index (the device's own builder, as gemini_amd/psnark.py does)
  {
    const Fr one = Fr::one(), zero = Fr::zero();
    uint64_t zeros = 0;
    if ((rc = gm_fr_vec_alloc(nnz, &zeros))) return fail(rc);
    rc = gm_fr_vec_fill(zeros, zero.l);
    uint64_t* outs[2] = {&I->row, &I->col};
    const uint64_t idx[2] = {I->row_index, I->col_index};
    for (int k = 0; k < 2 && !rc; k++) {
      rc = gm_fr_vec_alloc(nnz, outs[k]);
      if (!rc) rc = gm_fr_alg_hash(zeros, idx[k], one.l, *outs[k]);
    }
    (void)gm_fr_vec_free(zeros);
    if (rc) return fail(rc);
  }
  // the three value vectors over the joint support
  {
    uint64_t* outs[3] = {&I->val_a, &I->val_b, &I->val_c};
    std::vector<uint64_t> dense;
    for (int k = 0; k < 3; k++) {
      const Host& h = H[src_of[k]];
      dense.assign(4 * nnz, 0);
      size_t pos = 0;  // both sides ascend
      for (size_t e : h.order) {
        while (all[pos] != h.keys[e]) pos++;
        memcpy(&dense[4 * pos], &h.vals[4 * e], 32);
      }
      if ((rc = gm_fr_vec_alloc(nnz, outs[k]))) return fail(rc);
      if (nnz && (rc = gm_fr_vec_upload(*outs[k], 0, dense.data(), nnz))) return fail(rc);
    }
  }
  // extend_frequency(compute_frequency(set_len, index)): value v repeated 1 + #{j : index[j] = v} times
  {
    const size_t len_r = num_constraints <= 1 ? 1 : (size_t)1 << (64 - __builtin_clzll((unsigned long long)(num_constraints - 1)));
    const size_t set_len[2] = {len_r, num_variables};
    const std::vector<uint32_t>* index[2] = {&row_index, &col_index};
    uint64_t* outs[2] = {&I->ext_fre_row, &I->ext_fre_col};
    size_t* lens[2] = {&I->ext_fre_row_len, &I->ext_fre_col_len};
    for (int k = 0; k < 2; k++) {
      std::vector<uint32_t> freq(set_len[k], 1);
      for (uint32_t v : *index[k]) {
        if (v >= set_len[k]) return fail(GM_EINVAL);
        freq[v]++;
      }
      std::vector<uint32_t> ext;
      ext.reserve(set_len[k] + nnz);
      for (size_t v = 0; v < set_len[k]; v++) ext.insert(ext.end(), freq[v], (uint32_t)v);
      if ((rc = gm_idx_register(ext.data(), ext.size(), outs[k]))) return fail(rc);
      *lens[k] = ext.size();
    }
  }
  return GM_OK;
}

extern "C" int gm_psnark_preprocess_free(gm_psnark_instance* I) {
  if (!I) return GM_EINVAL;
  for (uint64_t* h : {&I->row_index, &I->col_index, &I->ext_fre_row, &I->ext_fre_col})
    if (*h) {
      (void)gm_idx_free(*h);
      *h = 0;
    }
  for (uint64_t* h : {&I->row, &I->col, &I->val_a, &I->val_b, &I->val_c})
    if (*h) {
      (void)gm_fr_vec_free(*h);
      *h = 0;
    }
  return GM_OK;
}

extern "C" int gm_psnark_index(const gm_psnark_instance* I, uint64_t ck_bases, uint64_t* out_jac) {
  if (!I || !out_jac) return GM_EINVAL;
  size_t nck = 0;
  RC(gm_ck_len(ck_bases, &nck));
  return batch_commit(ck_bases, nck, {I->row, I->col, I->val_a, I->val_b, I->val_c}, out_jac);
}

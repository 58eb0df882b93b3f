// snark::Proof::new_time (src/snark/time_prover.rs:19-117) with TensorcheckProof::new_time
// (src/subprotocols/tensorcheck/mod.rs:190-275), CommitterKey::{commit, batch_commit, batch_open_multi_points}
// (src/kzg/time.rs:81-159) and Sumcheck::new_time (sumcheck/proof.rs:125-130) as ONE entry point of the library.
//
// The reference's prover is compiled host code that drives its kernels (MSM, sumcheck rounds, vector passes); so is
// this: pure orchestration over the library's own C ABI (every O(n) step below is a gm_* call that a Rust / C++
// embedder could make itself -- gemini_amd/snark.py is the same sequence in Python and the tests hold the two byte for
// byte equal).  What a shim gains is one FFI call per proof: `Proof::new_time(&r1cs, &ck)` -> gm_snark_new_time.
#include "prover_common.hpp"

namespace {

using namespace gmprover;

}  // namespace

extern "C" int gm_snark_new_time(const uint64_t matrices[6], uint64_t z, uint64_t w, uint64_t ck_bases, int g1_encoding, size_t cap_rounds,
                                 gm_snark_proof* P) {
  if (!matrices || !P || !P->messages[0] || !P->messages[1] || !P->fold_commitments || !P->fold_evaluations) return GM_EINVAL;
  const auto t_all = Clock::now();
  Vecs V;
  size_t nz = 0, nw = 0, nck = 0;
  RC(vec_len(z, &nz));
  RC(vec_len(w, &nw));
  RC(gm_ck_len(ck_bases, &nck));
  // z_a, z_b, z_c (:32-34)
  // shapes: A, B, C have |z| columns and their transposes |z| rows -- abc_tensored below is exposed over all |z| entries
  for (int k = 0; k < 6; k++) {
    size_t rows = 0, cols = 0;
    RC(gm_spm_shape(matrices[k], &rows, &cols, nullptr));
    if ((k < 3 ? cols : rows) != nz) return GM_EINVAL;
  }
  RC(gm_footprint_admit(0, ck_bases, nz, 0, 0));  // room for the whole proof, or GM_ENOMEM with the numbers, before the first allocation
  uint64_t z_abc[3];
  for (int k = 0; k < 3; k++) {
    size_t rows = 0;
    RC(gm_spm_shape(matrices[k], &rows, nullptr, nullptr));
    RC(V.alloc(rows, &z_abc[k]));
    RC(gm_spm_mul(matrices[k], z, z_abc[k]));
  }
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";  // PROTOCOL_NAME, src/lib.rs:74
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));
  P->spans[0] = since(t_all);

  auto t0 = Clock::now();
  RC(gm_ck_msm(ck_bases, 0, 0, w, 0, nw < nck ? nw : nck, P->witness_commitment));  // ck.commit(&r1cs.w) :42
  P->spans[1] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  RC(gm_fr_eval_le(z_abc[2], alpha, 1, P->zc_alpha));  // :48
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1, ch2;
  RC(sumcheck_new_time(T.h, z_abc[0], z_abc[1], alpha, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));  // :52
  P->spans[2] = since(t0);
  for (int k = 0; k < 3; k++) V.release(z_abc[k]);  // vectors go at their last use: the peak of a proof is 9 vectors of n elements, not 16 (gm_snark_footprint)

  t0 = Clock::now();
  if (P->rounds[0] == 0) return GM_EINVAL;  // tensor() of no challenges: the reference asserts (src/misc.rs:134)
  const size_t nt = (size_t)1 << P->rounds[0];
  uint64_t b_ch, c_ch, a_ch;
  RC(V.alloc(nt, &b_ch));
  RC(gm_fr_tensor(ch1.data(), P->rounds[0], b_ch));  // :56
  RC(V.alloc(nt, &c_ch));
  RC(gm_fr_powers(alpha, nt, c_ch));  // :57
  RC(V.alloc(nt, &a_ch));
  RC(gm_fr_hadamard(b_ch, c_ch, a_ch));  // :58
  uint64_t eta[4];
  RC(gm_transcript_challenge_fr(T.h, L("eta"), 3, eta));
  uint64_t coeffs[12];
  Fr::one().to_limbs(coeffs);
  memcpy(coeffs + 4, eta, 32);
  Fr::from_limbs(eta).sqr().to_limbs(coeffs + 8);
  // abc_tensored[col] = sum_rows rA[i] A[i,col] + eta rB[i] B[i,col] + eta^2 rC[i] C[i,col]   :63-81
  uint64_t t_abc[3];
  const uint64_t rand_vecs[3] = {a_ch, b_ch, c_ch};
  for (int k = 0; k < 3; k++) {
    size_t rows = 0;
    RC(gm_spm_shape(matrices[3 + k], &rows, nullptr, nullptr));
    RC(V.alloc(rows > nz ? rows : nz, &t_abc[k]));
    RC(gm_spm_mul(matrices[3 + k], rand_vecs[k], t_abc[k]));
  }
  uint64_t abc;
  RC(V.alloc(nz, &abc));
  RC(gm_fr_lincomb(t_abc, coeffs, 3, abc));
  RC(gm_fr_vec_set_len(abc, nz));  // vec![0; z.len()]: no trimming (the trimmed tail is zero on the device)
  for (uint64_t v : {a_ch, b_ch, c_ch, t_abc[0], t_abc[1], t_abc[2]}) V.release(v);
  P->spans[3] = since(t0);

  t0 = Clock::now();
  uint64_t one[4];
  Fr::one().to_limbs(one);
  RC(sumcheck_new_time(T.h, abc, z, one, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :84-89
  P->spans[4] = since(t0);

  // ---- TensorcheckProof::new_time(transcript, ck, [w], [([abc_tensored, z], challenges)])   tensorcheck/mod.rs:190-275
  t0 = Clock::now();
  uint64_t batch_challenge[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, batch_challenge));
  uint64_t lc_coeffs[8];
  Fr::one().to_limbs(lc_coeffs);  // powers(batch_challenge, max_len)[0..2]
  memcpy(lc_coeffs + 4, batch_challenge, 32);
  const uint64_t body[2] = {abc, z};
  uint64_t batched;
  RC(V.alloc(nz, &batched));
  RC(gm_fr_lincomb(body, lc_coeffs, 2, batched));
  // foldings_polynomial (:124-133): successive folds with every challenge but the last
  std::vector<uint64_t> foldings;
  std::vector<size_t> fold_len;
  {
    size_t len = 0;
    RC(vec_len(batched, &len));
    for (size_t k = 0; k + 1 < P->rounds[1]; k++) {
      uint64_t nxt;
      len = (len + 1) / 2;
      RC(V.alloc(len, &nxt));
      foldings.push_back(nxt);
      fold_len.push_back(len);
    }
    RC(gm_fr_fold_chain(batched, ch2.data(), foldings.size(), foldings.data()));  // one wait for the whole tree
  }
  V.release(batched);
  V.release(abc);
  P->nfold = foldings.size();
  if (P->nfold > cap_rounds) return GM_EINVAL;
  if (P->nfold) {
    std::vector<size_t> ns(P->nfold);
    for (size_t k = 0; k < P->nfold; k++) ns[k] = fold_len[k] < nck ? fold_len[k] : nck;
    RC(gm_ck_msm_batch(ck_bases, foldings.data(), ns.data(), P->nfold, P->fold_commitments));  // batch_commit :98-107
  }
  for (size_t k = 0; k < P->nfold; k++) RC(gm_transcript_append_g1(T.h, L("commitment"), 10, P->fold_commitments + 18 * k, 1, 0));
  uint64_t pts[12];  // beta^2, beta, -beta
  RC(gm_transcript_challenge_fr(T.h, L("evaluation-chal"), 15, pts + 4));
  {
    const Fr beta = Fr::from_limbs(pts + 4);
    beta.sqr().to_limbs(pts);
    beta.neg().to_limbs(pts + 8);
  }
  RC(gm_fr_eval_le(w, pts, 3, P->base_evaluations));
  RC(gm_fr_eval_le_batch(foldings.data(), P->nfold, pts + 4, 2, P->fold_evaluations));  // one wait for all levels
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 4, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 8, 1));
  for (size_t k = 0; k < 2 * P->nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));
  // batch_open_multi_points (:149-159): commit((sum_i open_chal^i p_i) / ((x - beta^2)(x - beta)(x + beta)))
  {
    const size_t npoly = 1 + P->nfold;
    std::vector<uint64_t> polys(npoly), etas(4 * npoly);
    polys[0] = w;
    for (size_t k = 0; k < P->nfold; k++) polys[1 + k] = foldings[k];
    Fr acc = Fr::one();
    const Fr oc = Fr::from_limbs(open_chal);
    for (size_t k = 0; k < npoly; k++) {
      acc.to_limbs(etas.data() + 4 * k);
      acc = acc * oc;
    }
    size_t longest = nw;
    for (size_t l : fold_len) longest = l > longest ? l : longest;
    uint64_t combined, quotient;
    RC(V.alloc(longest, &combined));
    RC(gm_fr_lincomb(polys.data(), etas.data(), npoly, combined));
    size_t lc = 0;
    RC(vec_len(combined, &lc));
    RC(V.alloc(lc ? lc - 1 : 0, &quotient));
    uint64_t rem[12];
    RC(gm_fr_div_vanishing(combined, pts, 3, quotient, rem));
    V.release(combined);
    size_t lq = 0;
    RC(vec_len(quotient, &lq));
    RC(gm_ck_msm(ck_bases, 0, 0, quotient, 0, lq < nck ? lq : nck, P->evaluation_proof));
  }
  P->spans[5] = since(t0);
  P->spans[6] = since(t_all);
  return GM_OK;
}

// =====================================================================================================================
// snark::Proof::new_elastic (src/snark/elastic_prover.rs:174-266) with its `tensorcheck` (:105-168),
// CommitterKeyStream::{commit, commit_folding, open_multi_points, open_folding} (src/kzg/space.rs:95-285) and
// Sumcheck::new_elastic (sumcheck/proof.rs:145-154, elastic_prover.rs:44-57) as ONE entry point.
//
// The streams of the reference (`Reverse(..)` views, big-endian) are device-resident reversed vectors; the key stays in
// time order in HBM and the stream view `Reverse(powers_of_g)` + advance_by is the reversed / offset addressing of the MSM
// entry points.  `max_msm_buffer` bounds HOST buffering in the reference; here every flush shorter than
// `min_device_chunk` pairs is merged into one device MSM (gemini_amd/kzg.py::CommitterKeyStream, same rule).  The
// reference re-streams the folded polynomial tree three times to stay in O(log n) memory; with the streams resident
// the levels are folded once.  gemini_amd/snark.py::new_elastic is the same sequence in Python; the tests hold the two
// -- and the time prover, `assert_eq!(time_proof, space_proof)` src/snark/tests.rs:56 -- byte for byte equal.
// =====================================================================================================================
namespace {
using namespace gmprover;
}  // namespace

extern "C" int gm_snark_new_elastic(const uint64_t matrices_t[3], uint64_t z_stream, uint64_t w_stream, uint64_t za_stream, uint64_t zb_stream,
                                    uint64_t zc_stream, uint64_t ck_bases, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding,
                                    size_t cap_rounds, gm_snark_proof* P) {
  if (!matrices_t || !P || !P->messages[0] || !P->messages[1] || !P->fold_commitments || !P->fold_evaluations) return GM_EINVAL;
  const auto t_all = Clock::now();
  Vecs V;
  size_t nz = 0, nw = 0, nck = 0, nzc = 0;
  RC(vec_len(z_stream, &nz));
  RC(vec_len(w_stream, &nw));
  RC(vec_len(zc_stream, &nzc));
  RC(gm_ck_len(ck_bases, &nck));
  if (nw > nck || nz > nck) return GM_EINVAL;  // the streaming committer insists on a key as long as every stream (space.rs:169-175)
  for (int k = 0; k < 3; k++) {
    size_t rows = 0;
    RC(gm_spm_shape(matrices_t[k], &rows, nullptr, nullptr));
    if (rows != nz) return GM_EINVAL;
  }
  RC(gm_footprint_admit(0, ck_bases, nz, 0, min_device_chunk > 1 ? 1 : 2));  // (2: the literal schedule holds the reversed copies as well)
  const size_t flush = max_msm_buffer > min_device_chunk ? max_msm_buffer : min_device_chunk;
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));
  P->spans[0] = 0.0;  // the matrix products z_a, z_b, z_c belong to the stream construction (R1csStream), not to the prover

  auto t0 = Clock::now();
  RC(stream_msm(ck_bases, w_stream, nw, nw - 1, (size_t)1 << 20 > min_device_chunk ? (size_t)1 << 20 : min_device_chunk,
                P->witness_commitment));  // ck.commit(witness): msm_chunks of 2^20 (:209, space.rs:169-177)
  P->spans[1] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  {
    uint64_t zc_le;
    RC(V.alloc(nzc, &zc_le));
    RC(gm_fr_reverse(zc_stream, zc_le));
    RC(gm_fr_eval_le(zc_le, alpha, 1, P->zc_alpha));  // evaluate_be(z_c, alpha) :216
    V.release(zc_le);
  }
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1, ch2;
  // min_device_chunk > 1 (the default): everything is resident and max_msm_buffer advisory -- then the sumchecks take the RESIDENT
  // schedule as well: time provers on the little-endian vectors from the first round instead of space provers that re-derive
  // every message from the whole streams until SPACE_TIME_THRESHOLD rounds remain (the same field elements: sumcheck/tests.rs:42-87).
  // min_device_chunk = 1 is the literal elastic prover.
  const bool resident = min_device_chunk > 1;
  if (resident) {
    uint64_t za_le, zb_le;
    size_t na = 0, nb = 0;
    RC(vec_len(za_stream, &na));
    RC(vec_len(zb_stream, &nb));
    RC(V.alloc(na, &za_le));
    RC(gm_fr_reverse(za_stream, za_le));
    RC(V.alloc(nb, &zb_le));
    RC(gm_fr_reverse(zb_stream, zb_le));
    RC(sumcheck_new_time(T.h, za_le, zb_le, alpha, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));
    V.release(za_le);
    V.release(zb_le);
  } else {
    RC(sumcheck_new_elastic(T.h, za_stream, zb_stream, alpha, P->messages[0], ch1, cap_rounds, P->final_foldings[0], &P->rounds[0]));  // :222
  }
  P->spans[2] = since(t0);

  t0 = Clock::now();
  uint64_t eta[4];
  RC(gm_transcript_challenge_fr(T.h, L("eta"), 3, eta));
  const size_t nt = (size_t)1 << P->rounds[0];
  uint64_t a_ch, b_ch, c_ch;
  RC(V.alloc(nt, &b_ch));
  RC(gm_fr_tensor(ch1.data(), P->rounds[0], b_ch));  // MatrixTensor streams (:233-238) on the transposed matrices
  RC(V.alloc(nt, &c_ch));
  RC(gm_fr_powers(alpha, nt, c_ch));
  RC(V.alloc(nt, &a_ch));
  RC(gm_fr_hadamard(b_ch, c_ch, a_ch));
  uint64_t coeffs[12];
  Fr::one().to_limbs(coeffs);
  memcpy(coeffs + 4, eta, 32);
  (Fr::from_limbs(eta) * Fr::from_limbs(eta)).to_limbs(coeffs + 8);
  uint64_t t_abc[3];
  const uint64_t rand_vecs[3] = {a_ch, b_ch, c_ch};
  for (int k = 0; k < 3; k++) {
    RC(V.alloc(nz, &t_abc[k]));
    RC(gm_spm_mul(matrices_t[k], rand_vecs[k], t_abc[k]));
  }
  uint64_t lhs_le, lhs = 0, z_le;
  RC(V.alloc(nz, &lhs_le));
  RC(gm_fr_lincomb(t_abc, coeffs, 3, lhs_le));
  RC(gm_fr_vec_set_len(lhs_le, nz));
  for (uint64_t v : {a_ch, b_ch, c_ch, t_abc[0], t_abc[1], t_abc[2]}) V.release(v);
  RC(V.alloc(nz, &z_le));
  RC(gm_fr_reverse(z_stream, z_le));
  if (!resident) {
    RC(V.alloc(nz, &lhs));
    RC(gm_fr_reverse(lhs_le, lhs));
  }
  P->spans[3] = since(t0);

  t0 = Clock::now();
  uint64_t one[4];
  Fr::one().to_limbs(one);
  if (resident) {
    RC(sumcheck_new_time(T.h, lhs_le, z_le, one, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));
  } else {
    RC(sumcheck_new_elastic(T.h, lhs, z_stream, one, P->messages[1], ch2, cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :241
    V.release(lhs);
  }
  P->spans[4] = since(t0);

  // ---- tensorcheck (:105-168) over the folded polynomial tree of body = lhs + batch_challenge * z
  t0 = Clock::now();
  uint64_t batch_challenge[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, batch_challenge));
  uint64_t body_le;
  uint64_t lc_coeffs[8];
  Fr::one().to_limbs(lc_coeffs);
  memcpy(lc_coeffs + 4, batch_challenge, 32);
  const uint64_t body[2] = {lhs_le, z_le};
  RC(V.alloc(nz, &body_le));
  RC(gm_fr_lincomb(body, lc_coeffs, 2, body_le));
  std::vector<uint64_t> levels;
  std::vector<size_t> level_len;
  {
    size_t len = 0;
    RC(vec_len(body_le, &len));
    for (size_t k = 0; k + 1 < P->rounds[1]; k++) {  // strip_last
      uint64_t nxt;
      len = (len + 1) / 2;
      RC(V.alloc(len, &nxt));
      levels.push_back(nxt);
      level_len.push_back(len);
    }
    RC(gm_fr_fold_chain(body_le, ch2.data(), levels.size(), levels.data()));
  }
  for (uint64_t v : {body_le, lhs_le, z_le}) V.release(v);
  P->nfold = levels.size();
  if (P->nfold > cap_rounds) return GM_EINVAL;
  if (P->nfold) {  // commit_folding (space.rs:192-223): one ChunkedPippenger of max_msm_buffer / depth per level
    const size_t per = max_msm_buffer / P->nfold ? max_msm_buffer / P->nfold : 1;
    const size_t lvl_flush = per > min_device_chunk ? per : min_device_chunk;
    bool cut = false;
    for (size_t l : level_len) cut = cut || l > lvl_flush;
    if (!cut) {  // no level is cut: the commitments are sum_i level[i] * tau^i g whichever way the pairs are walked
      RC(gm_ck_msm_batch(ck_bases, levels.data(), level_len.data(), P->nfold, P->fold_commitments));
    } else {
      for (size_t k = 0; k < P->nfold; k++) {
        uint64_t s;
        RC(V.alloc(level_len[k], &s));
        RC(gm_fr_reverse(levels[k], s));
        RC(stream_msm(ck_bases, s, level_len[k], level_len[k] - 1, lvl_flush, P->fold_commitments + 18 * k));
        V.release(s);
      }
    }
  }
  for (size_t k = 0; k < P->nfold; k++) RC(gm_transcript_append_g1(T.h, L("commitment"), 10, P->fold_commitments + 18 * k, 1, 0));
  uint64_t pts[12];  // beta^2, beta, -beta
  RC(gm_transcript_challenge_fr(T.h, L("evaluation-chal"), 15, pts + 4));
  {
    const Fr beta = Fr::from_limbs(pts + 4);
    beta.sqr().to_limbs(pts);
    beta.neg().to_limbs(pts + 8);
  }
  RC(gm_fr_eval_le_batch(levels.data(), P->nfold, pts + 4, 2, P->fold_evaluations));  // evaluate_folding, tensorcheck/mod.rs:73-88
  uint64_t w_le;
  RC(V.alloc(nw, &w_le));
  RC(gm_fr_reverse(w_stream, w_le));
  RC(gm_fr_eval_le(w_le, pts, 3, P->base_evaluations));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 4, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 8, 1));
  for (size_t k = 0; k < 2 * P->nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));
  {
    // The reference adds two openings: open_multi_points(w) (space.rs:128-166) and open_folding (:229-285) = sum_i open_chal^(i + 1)
    // commit(quotient of level i), the HashMapPippenger of the latter merging the scalars of equal bases.  Both walk the same key
    // and division by the same Z is linear, so the merge extends over both: the quotient of  w + sum_i open_chal^(i + 1) level_i
    // -- one linear combination, ONE division (instead of one latency-bound scan per level), one stream MSM flushed every
    // max_msm_buffer pairs.  (It is the polynomial the time prover opens: time_prover.rs:98-107 -> kzg/time.rs:149-159.)
    std::vector<uint64_t> polys(1 + P->nfold), etas(4 * (1 + P->nfold));
    polys[0] = w_le;
    Fr acc = Fr::one();
    const Fr oc = Fr::from_limbs(open_chal);
    size_t longest = nw;
    for (size_t k = 0; k <= P->nfold; k++) {
      acc.to_limbs(etas.data() + 4 * k);
      acc = acc * oc;
      if (k) {
        polys[k] = levels[k - 1];
        longest = level_len[k - 1] > longest ? level_len[k - 1] : longest;
      }
    }
    size_t lb = 0;
    if (longest > 3) {
      uint64_t combined, q, bs, rem[12];
      RC(V.alloc(longest, &combined));
      RC(gm_fr_lincomb(polys.data(), etas.data(), polys.size(), combined));
      size_t lc = 0;
      RC(vec_len(combined, &lc));
      RC(V.alloc(lc ? lc - 1 : 0, &q));
      RC(gm_fr_div_vanishing(combined, pts, 3, q, rem));
      V.release(combined);
      RC(vec_len(q, &lb));
      if (lb) {
        RC(V.alloc(lb, &bs));
        RC(gm_fr_reverse(q, bs));
        V.release(q);
        RC(stream_msm(ck_bases, bs, lb, lb - 1, flush, P->evaluation_proof));
      }
    }
    if (lb == 0) RC(gm_g1_sum(nullptr, 0, P->evaluation_proof));
  }
  P->spans[5] = since(t0);
  P->spans[6] = since(t_all);
  return GM_OK;
}

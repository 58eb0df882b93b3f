// Helpers shared by the provers compiled into the library (snark.cpp, psnark.cpp): pure orchestration over the library's own C ABI.
#pragma once
#include <chrono>
#include <cstring>
#include <vector>

#include "../../include/gemini_hip.h"
#include "host_field.hpp"

namespace gmprover {

using gmh::Fr;
using Clock = std::chrono::steady_clock;

inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// device vectors owned by one proof: freed on every exit path
struct Vecs {
  std::vector<uint64_t> h;
  ~Vecs() {
    for (uint64_t v : h) (void)gm_fr_vec_free(v);
  }
  int alloc(size_t n, uint64_t* out) {
    int rc = gm_fr_vec_alloc(n, out);
    if (!rc) h.push_back(*out);
    return rc;
  }
  // free one vector before the end of the proof (temporaries of a phase: the preprocessing prover at 2^26 constraints
  // holds ~100 GB of live vectors as it is)
  void release(uint64_t v) {
    for (size_t i = 0; i < h.size(); i++)
      if (h[i] == v) {
        (void)gm_fr_vec_free(v);
        h[i] = h.back();
        h.pop_back();
        return;
      }
  }
};
struct TranscriptGuard {
  uint64_t h = 0;
  ~TranscriptGuard() {
    if (h) (void)gm_transcript_free(h);
  }
};

#define RC(x)            \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)

inline const uint8_t* L(const char* s) { return reinterpret_cast<const uint8_t*>(s); }

inline int vec_len(uint64_t v, size_t* n) { return gm_fr_vec_len(v, n); }

// Sumcheck::new_time (proof.rs:125-130): the prover reads f and g in place (the round loop runs inside this call and the
// callers do not touch the vectors meanwhile), its folds go to buffers of its own
inline int sumcheck_new_time(uint64_t transcript, uint64_t f, uint64_t g, const uint64_t twist[4], uint64_t* messages, std::vector<uint64_t>& challenges,
                             size_t cap_rounds, uint64_t final_foldings[8], size_t* rounds) {
  uint64_t prover = 0;
  RC(gm_sc_new_borrow(f, g, twist, &prover));
  challenges.assign(cap_rounds * 4, 0);
  int rc = gm_sumcheck_prove(transcript, prover, messages, challenges.data(), cap_rounds, final_foldings, rounds);
  (void)gm_sc_free(prover);
  if (!rc) challenges.resize(*rounds * 4);
  return rc;
}


// ---- shared by snark.cpp / psnark.cpp / psnark_elastic.cpp -------------------------------------------------------------
constexpr size_t SPACE_TIME_THRESHOLD = 22;  // src/lib.rs:76

// sum over stream positions k < len: stream[k] * power[top - k], flushed every max(chunk, min_chunk) pairs
// (msm_chunks / ChunkedPippenger composition, src/kzg/space.rs:22-55)
inline int stream_msm(uint64_t bases, uint64_t stream, size_t len, size_t top, size_t chunk, uint64_t out[18]) {
  if (len == 0) return gm_g1_sum(nullptr, 0, out);
  if (chunk == 0) chunk = 1;
  if (len <= chunk) return gm_ck_msm(bases, top, 1, stream, 0, len, out);
  std::vector<uint64_t> parts;
  for (size_t off = 0; off < len; off += chunk) {
    const size_t m = len - off < chunk ? len - off : chunk;
    parts.resize(parts.size() + 18);
    RC(gm_ck_msm(bases, top - off, 1, stream, off, m, parts.data() + parts.size() - 18));
  }
  return gm_g1_sum(parts.data(), parts.size() / 18, out);
}

// Sumcheck::prove over an ElasticProver: a SpaceProver that becomes a TimeProver when fewer than
// SPACE_TIME_THRESHOLD rounds remain (elastic_prover.rs:44-57); Prover::next_message folds first.
inline int sumcheck_new_elastic(uint64_t transcript, uint64_t f_stream, uint64_t g_stream, const uint64_t twist[4], uint64_t* messages,
                         std::vector<uint64_t>& challenges, size_t cap_rounds, uint64_t final_foldings[8], size_t* rounds) {
  uint64_t space = 0, time = 0;
  RC(gm_sp_new_borrow(f_stream, g_stream, twist, &space));
  struct Guard {
    uint64_t &s, &t;
    ~Guard() {
      if (t) (void)gm_sc_free(t);
      if (s) (void)gm_sp_free(s);
    }
  } guard{space, time};
  challenges.assign(cap_rounds * 4, 0);
  size_t k = 0;
  const uint64_t* vm = nullptr;
  for (;;) {
    if (vm && !time) {  // ElasticProver::fold
      size_t tot = 0, rnd = 0;
      RC(gm_sp_rounds(space, &tot, &rnd));
      if (tot - rnd < SPACE_TIME_THRESHOLD) {
        RC(gm_sp_to_time(space, &time));
        RC(gm_sc_fold(time, vm));
        (void)gm_sp_free(space);
        space = 0;
      } else {
        RC(gm_sp_fold(space, vm));
      }
      vm = nullptr;
    }
    uint64_t a[4], b[4];
    int has = 0;
    if (time) RC(gm_sc_round(time, vm, a, b, &has));
    else RC(gm_sp_round(space, vm, a, b, &has));
    if (!has) break;
    if (k >= cap_rounds) return GM_EINVAL;
    memcpy(messages + 8 * k, a, 32);
    memcpy(messages + 8 * k + 4, b, 32);
    RC(gm_transcript_append_fr(transcript, L("evaluations"), 11, messages + 8 * k, 2));
    RC(gm_transcript_challenge_fr(transcript, L("challenge"), 9, challenges.data() + 4 * k));
    vm = challenges.data() + 4 * k;
    k++;
  }
  int has = 0;
  if (time) RC(gm_sc_final(time, final_foldings, final_foldings + 4, &has));
  else RC(gm_sp_final(space, final_foldings, final_foldings + 4, &has));
  if (!has) return GM_ESTATE;
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings, 1));
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings + 4, 1));
  *rounds = k;
  challenges.resize(k * 4);
  return GM_OK;
}


inline Fr fr_pow(Fr base, size_t e) {
  Fr acc = Fr::one();
  while (e) {
    if (e & 1) acc = acc * base;
    base = base.sqr();
    e >>= 1;
  }
  return acc;
}

// ck.commit(v): msm_unchecked truncates to the shorter side (src/kzg/time.rs:82)
inline int commit(uint64_t ck, size_t nck, uint64_t v, uint64_t out[18]) {
  size_t n = 0;
  RC(vec_len(v, &n));
  return gm_ck_msm(ck, 0, 0, v, 0, n < nck ? n : nck, out);
}
inline int batch_commit(uint64_t ck, size_t nck, const std::vector<uint64_t>& vs, uint64_t* out) {
  std::vector<size_t> ns(vs.size());
  for (size_t k = 0; k < vs.size(); k++) {
    RC(vec_len(vs[k], &ns[k]));
    if (ns[k] > nck) ns[k] = nck;
  }
  return gm_ck_msm_batch(ck, vs.data(), ns.data(), vs.size(), out);
}

// batch_open_multi_points (src/kzg/time.rs:149-159): commit((sum_i chal^i p_i) / prod (x - point_j))
inline int batch_open(Vecs& V, uint64_t ck, size_t nck, const std::vector<uint64_t>& polys, const uint64_t* pts, size_t npts, const uint64_t chal[4],
               uint64_t out[18]) {
  std::vector<uint64_t> etas(4 * polys.size());
  Fr acc = Fr::one();
  const Fr c = Fr::from_limbs(chal);
  size_t longest = 0;
  for (size_t k = 0; k < polys.size(); k++) {
    acc.to_limbs(etas.data() + 4 * k);
    acc = acc * c;
    size_t l = 0;
    RC(vec_len(polys[k], &l));
    longest = l > longest ? l : longest;
  }
  uint64_t combined, quotient;
  RC(V.alloc(longest, &combined));
  RC(gm_fr_lincomb(polys.data(), etas.data(), polys.size(), combined));
  size_t lc = 0;
  RC(vec_len(combined, &lc));
  RC(V.alloc(lc ? lc - 1 : 0, &quotient));
  uint64_t rem[12];
  RC(gm_fr_div_vanishing(combined, pts, npts, quotient, rem));
  V.release(combined);
  const int rc = commit(ck, nck, quotient, out);
  V.release(quotient);
  return rc;
}

// plookup (plookup/time_prover.rs:89-112) -> lookup_set, lookup_subset, lookup_sorted
inline int plookup(Vecs& V, uint64_t subset, uint64_t set_, uint64_t index, size_t index_len, uint64_t ext_fre, size_t ext_len, const uint64_t y[4],
            const uint64_t z[4], const uint64_t zeta[4], uint64_t out[3]) {
  size_t nset = 0, nsub = 0;
  RC(vec_len(set_, &nset));
  RC(vec_len(subset, &nsub));
  uint64_t set_h = set_, subset_h = subset;
  if (!Fr::from_limbs(zeta).is_zero()) {
    RC(V.alloc(nset, &set_h));
    RC(gm_fr_alg_hash(set_, 0, zeta, set_h));
    const size_t n = nsub < index_len ? nsub : index_len;
    RC(V.alloc(n, &subset_h));
    RC(gm_fr_alg_hash(subset, index, zeta, subset_h));
    nsub = n;
  }
  RC(V.alloc(nset ? nset + 1 : 0, &out[0]));
  RC(gm_fr_plookup_set(set_h, y, z, out[0]));
  RC(V.alloc(nsub, &out[1]));
  RC(gm_fr_add_scalar(subset_h, y, out[1]));
  uint64_t srt;
  RC(V.alloc(ext_len, &srt));
  RC(gm_fr_gather(set_h, ext_fre, srt));
  RC(V.alloc(ext_len ? ext_len + 1 : 0, &out[2]));
  RC(gm_fr_plookup_set(srt, y, z, out[2]));
  V.release(srt);
  if (set_h != set_) V.release(set_h);
  if (subset_h != subset) V.release(subset_h);
  return GM_OK;
}


}  // namespace gmprover

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

// Sumcheck::new_time (proof.rs:125-130): prover over copies of f and g, round loop inside the library
inline int sumcheck_new_time(uint64_t transcript, uint64_t f, uint64_t g, const uint64_t twist[4], uint64_t* messages, std::vector<uint64_t>& challenges,
                             size_t cap_rounds, uint64_t final_foldings[8], size_t* rounds) {
  uint64_t prover = 0;
  RC(gm_sc_new_v(f, g, twist, &prover));
  challenges.assign(cap_rounds * 4, 0);
  int rc = gm_sumcheck_prove(transcript, prover, messages, challenges.data(), cap_rounds, final_foldings, rounds);
  (void)gm_sc_free(prover);
  if (!rc) challenges.resize(*rounds * 4);
  return rc;
}

}  // namespace gmprover

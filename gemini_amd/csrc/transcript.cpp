// Fiat-Shamir transcript of the Gemini prover, host side of libgemini_hip.so.
//
// Replaces merlin 3.0.0 `Transcript` (STROBE-128 over Keccak-f[1600]; crates.io dependency pinned in
// Cargo.lock:606-608, keccak 0.1.4 :564-566 -- not vendored in the reference) together with
// `GeminiTranscript` (src/transcript.rs:16-34) and the ark-serialize framing of the values the
// prover absorbs (Fr, RoundMsg, [F; 2], G1 commitments).  64-byte hashing per round: this stays on
// the host by design, but it is parity-critical, so it lives in the
// product library and is pinned against merlin's published test vector in tests/.
//
// Serialisation conventions are those of ark-serialize 0.4 for ark-test-curves' bls12_381 (the
// curve crate the reference's examples and tests use, examples/snark.rs:11-13):
//   Fr            32 bytes little-endian canonical
//   G1 (uncompr.) x (48 B LE) || y (48 B LE), flags in the top bits of the last byte:
//                 bit 7 = y is the lexicographically larger root, bit 6 = point at infinity
// These are recalled from the crates' sources (not checkable in this image); they are isolated in
// this file so that a correction is local.
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/gemini_hip.h"
#include "host_field.hpp"

namespace gm {
void set_error(const char* fmt, ...);
}

namespace {

const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

inline uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// FIPS-202 Keccak-p[1600, 24] on 25 little-endian lanes, lane index = x + 5y
void keccak_f1600(uint64_t A[25]) {
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t Cc[5], D[5], Bm[25];
    for (int x = 0; x < 5; x++) Cc[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
    for (int x = 0; x < 5; x++) D[x] = Cc[(x + 4) % 5] ^ rol(Cc[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) Bm[y + 5 * ((2 * x + 3 * y) % 5)] = rol(A[x + 5 * y], KECCAK_ROT[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) A[x + 5 * y] = Bm[x + 5 * y] ^ ((~Bm[(x + 1) % 5 + 5 * y]) & Bm[(x + 2) % 5 + 5 * y]);
    A[0] ^= KECCAK_RC[rnd];
  }
}

// STROBE-128/1600, the subset merlin uses (meta-AD, AD, PRF)
struct Strobe128 {
  static constexpr int R = 166;
  static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
  uint8_t st[200];
  int pos = 0, pos_begin = 0;
  uint8_t cur_flags = 0;

  void permute() {
    uint64_t lanes[25];
    for (int i = 0; i < 25; i++) {
      uint64_t v = 0;
      for (int b = 7; b >= 0; b--) v = (v << 8) | st[8 * i + b];
      lanes[i] = v;
    }
    keccak_f1600(lanes);
    for (int i = 0; i < 25; i++)
      for (int b = 0; b < 8; b++) st[8 * i + b] = (uint8_t)(lanes[i] >> (8 * b));
  }
  explicit Strobe128(const uint8_t* label, size_t len) {
    memset(st, 0, sizeof st);
    const uint8_t init[6] = {1, (uint8_t)(R + 2), 1, 0, 1, 96};
    memcpy(st, init, 6);
    memcpy(st + 6, "STROBEv1.0.2", 12);
    permute();
    meta_ad(label, len, false);
  }
  void run_f() {
    st[pos] ^= (uint8_t)pos_begin;
    st[pos + 1] ^= 0x04;
    st[R + 1] ^= 0x80;
    permute();
    pos = 0;
    pos_begin = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      st[pos++] ^= d[i];
      if (pos == R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      d[i] = st[pos];
      st[pos++] = 0;
      if (pos == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;  // continuation of the same operation
    uint8_t old_begin = (uint8_t)pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
    if (force_f && pos != 0) run_f();
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(d, n);
  }
  void ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_A, more);
    absorb(d, n);
  }
  void prf(uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, more);
    squeeze(d, n);
  }
};

// merlin::Transcript
struct Transcript {
  Strobe128 strobe;
  int g1_encoding = 0;  // 0: ark-ec default short-Weierstrass framing; 1: ark-bls12-381's zcash framing
  explicit Transcript(const uint8_t* label, size_t len) : strobe((const uint8_t*)"Merlin v1.0", 11) {
    append_message((const uint8_t*)"dom-sep", 7, label, len);
  }
  void append_message(const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
    uint8_t le[4] = {(uint8_t)mlen, (uint8_t)(mlen >> 8), (uint8_t)(mlen >> 16), (uint8_t)(mlen >> 24)};
    strobe.meta_ad(label, llen, false);
    strobe.meta_ad(le, 4, true);
    strobe.ad(msg, mlen, false);
  }
  void challenge_bytes(const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
    uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe.meta_ad(label, llen, false);
    strobe.meta_ad(le, 4, true);
    strobe.prf(out, n, false);
  }
};

std::mutex g_mu;
std::unordered_map<uint64_t, std::unique_ptr<Transcript>> g_transcripts;
uint64_t g_next = 1;

Transcript* find(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_transcripts.find(h);
  return it == g_transcripts.end() ? nullptr : it->second.get();
}

void fr_serialize(const uint64_t mont[4], uint8_t out[32]) {
  uint64_t c[4];
  gmh::Fr::from_limbs(mont).to_canonical(c);
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(c[i] >> (8 * b));
}

}  // namespace

#define T_CHECK(cond, code, ...)    \
  do {                              \
    if (!(cond)) {                  \
      ::gm::set_error(__VA_ARGS__); \
      return (code);                \
    }                               \
  } while (0)

extern "C" {

int gm_transcript_new(const uint8_t* label, size_t len, uint64_t* handle) {
  T_CHECK(handle && (label || len == 0), GM_EINVAL, "transcript_new: null pointer");
  auto t = std::make_unique<Transcript>(label, len);
  std::lock_guard<std::mutex> lk(g_mu);
  *handle = g_next++;
  g_transcripts[*handle] = std::move(t);
  return GM_OK;
}

int gm_transcript_free(uint64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  T_CHECK(g_transcripts.erase(handle) == 1, GM_EHANDLE, "transcript_free: unknown handle %llu", (unsigned long long)handle);
  return GM_OK;
}

int gm_transcript_append_message(uint64_t handle, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_append_message: unknown handle %llu", (unsigned long long)handle);
  t->append_message(label, llen, msg, mlen);
  return GM_OK;
}

int gm_transcript_challenge_bytes(uint64_t handle, const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_challenge_bytes: unknown handle %llu", (unsigned long long)handle);
  t->challenge_bytes(label, llen, out, n);
  return GM_OK;
}

// append_serializable for `count` field elements laid out consecutively: one Fr (count = 1),
// a RoundMsg (a || b, count = 2), an [F; 2] ...        src/transcript.rs:16-24
int gm_transcript_append_fr(uint64_t handle, const uint8_t* label, size_t llen, const uint64_t* mont, size_t count) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_append_fr: unknown handle %llu", (unsigned long long)handle);
  std::vector<uint8_t> buf(32 * count);
  for (size_t i = 0; i < count; i++) fr_serialize(mont + 4 * i, buf.data() + 32 * i);
  t->append_message(label, llen, buf.data(), buf.size());
  return GM_OK;
}

// append_serializable for `count` G1 elements (Commitment / EvaluationProof wrap one G1 each;
// a Vec<Commitment> is prefixed by its u64 length when `with_len` != 0)
int gm_transcript_append_g1(uint64_t handle, const uint8_t* label, size_t llen, const uint64_t* jac, size_t count, int with_len) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_append_g1: unknown handle %llu", (unsigned long long)handle);
  std::vector<uint8_t> buf;
  if (with_len)
    for (int b = 0; b < 8; b++) buf.push_back((uint8_t)((uint64_t)count >> (8 * b)));
  for (size_t i = 0; i < count; i++) {
    uint8_t enc[96];
    memset(enc, 0, sizeof enc);
    gmh::G1 p = gmh::G1::from_limbs(jac + 18 * i).normalized();
    if (t->g1_encoding == 1) {
      // ark-bls12-381 `serialize_with_mode` override, Compress::No: x || y big-endian, bit 7 of byte 0 clear
      // (uncompressed), bit 6 = infinity, no sort flag
      if (p.is_identity()) {
        enc[0] |= 1u << 6;
      } else {
        uint64_t x[6], y[6];
        p.x.to_canonical(x);
        p.y.to_canonical(y);
        for (int k = 0; k < 6; k++)
          for (int b = 0; b < 8; b++) {
            enc[47 - (8 * k + b)] = (uint8_t)(x[k] >> (8 * b));
            enc[95 - (8 * k + b)] = (uint8_t)(y[k] >> (8 * b));
          }
      }
    } else if (p.is_identity()) {
      enc[95] |= 1u << 6;
    } else {
      uint64_t x[6], y[6], ny[6];
      p.x.to_canonical(x);
      p.y.to_canonical(y);
      p.y.neg().to_canonical(ny);
      for (int k = 0; k < 6; k++)
        for (int b = 0; b < 8; b++) {
          enc[8 * k + b] = (uint8_t)(x[k] >> (8 * b));
          enc[48 + 8 * k + b] = (uint8_t)(y[k] >> (8 * b));
        }
      if (!gmh::geq<6>(ny, y)) enc[95] |= 1u << 7;  // y > -y
    }
    buf.insert(buf.end(), enc, enc + 96);
  }
  t->append_message(label, llen, buf.data(), buf.size());
  return GM_OK;
}

// Which ark-serialize framing `append_serializable` uses for G1: 0 = ark-ec's default (ark-test-curves, the
// reference's examples and tests), 1 = the zcash framing ark-bls12-381 substitutes (the reference's benches).
int gm_transcript_set_g1_encoding(uint64_t handle, int encoding) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_set_g1_encoding: unknown handle %llu", (unsigned long long)handle);
  T_CHECK(encoding == 0 || encoding == 1, GM_EINVAL, "transcript_set_g1_encoding: %d is not 0 (arkworks) or 1 (zcash)", encoding);
  t->g1_encoding = encoding;
  return GM_OK;
}

// GeminiTranscript::get_challenge::<Fr>: 64 challenge bytes -> Fr::from_random_bytes (first 32 bytes
// little-endian, bit 255 cleared, accept iff < r), retried until it succeeds.  src/transcript.rs:26-34
int gm_transcript_challenge_fr(uint64_t handle, const uint8_t* label, size_t llen, uint64_t out_mont[4]) {
  Transcript* t = find(handle);
  T_CHECK(t, GM_EHANDLE, "transcript_challenge_fr: unknown handle %llu", (unsigned long long)handle);
  for (;;) {
    uint8_t bytes[64];
    t->challenge_bytes(label, llen, bytes, 64);
    uint64_t c[4];
    for (int i = 0; i < 4; i++) {
      uint64_t v = 0;
      for (int b = 7; b >= 0; b--) v = (v << 8) | bytes[8 * i + b];
      c[i] = v;
    }
    c[3] &= 0x7fffffffffffffffULL;
    if (!gmh::geq<4>(c, gmh::FrP::MOD)) {
      gmh::Fr::from_canonical(c).to_limbs(out_mont);
      return GM_OK;
    }
  }
}

// Sumcheck::prove round loop (src/subprotocols/sumcheck/proof.rs:36-66) over a device prover:
// message -> absorb b"evaluations" -> challenge b"challenge" -> next_message(Some(challenge)) ...
// then the two b"final-folding" absorbs.  messages: rounds x 8 u64 (a || b), challenges: rounds x 4,
// final_foldings: 8 u64.  *rounds_out = number of messages produced.
int gm_sumcheck_prove(uint64_t transcript, uint64_t prover, uint64_t* messages, uint64_t* challenges, size_t cap_rounds,
                      uint64_t final_foldings[8], size_t* rounds_out) {
  T_CHECK(messages && challenges && final_foldings && rounds_out, GM_EINVAL, "sumcheck_prove: null pointer");
  size_t k = 0;
  const uint64_t* vm = nullptr;
  for (;;) {
    uint64_t a[4], b[4];
    int has = 0;
    int rc = gm_sc_round(prover, vm, a, b, &has);
    if (rc) return rc;
    if (!has) break;
    T_CHECK(k < cap_rounds, GM_EINVAL, "sumcheck_prove: more than %zu rounds", cap_rounds);
    memcpy(messages + 8 * k, a, 32);
    memcpy(messages + 8 * k + 4, b, 32);
    if ((rc = gm_transcript_append_fr(transcript, (const uint8_t*)"evaluations", 11, messages + 8 * k, 2))) return rc;
    if ((rc = gm_transcript_challenge_fr(transcript, (const uint8_t*)"challenge", 9, challenges + 4 * k))) return rc;
    vm = challenges + 4 * k;
    k++;
  }
  int has = 0;
  int rc = gm_sc_final(prover, final_foldings, final_foldings + 4, &has);
  if (rc) return rc;
  T_CHECK(has, GM_ESTATE, "sumcheck_prove: final foldings unavailable");
  if ((rc = gm_transcript_append_fr(transcript, (const uint8_t*)"final-folding", 13, final_foldings, 1))) return rc;
  if ((rc = gm_transcript_append_fr(transcript, (const uint8_t*)"final-folding", 13, final_foldings + 4, 1))) return rc;
  *rounds_out = k;
  return GM_OK;
}

// Sumcheck::prove_batch (src/subprotocols/sumcheck/proof.rs:69-122): k provers of possibly different
// lengths run in lock-step for max(rounds) + 1 rounds; coefficients c_j are drawn first
// (b"batch-sumcheck"); a prover that has run out contributes (f0 * g0, 0); the round message is
// sum_j c_j * m_j.  The reference maps provers over rayon (:85); here the rounds of all live provers
// are ONE kernel launch (k_sc_round_multi).  messages: cap_rounds x 8, challenges: cap_rounds x 4,
// final_foldings: k x 8 (lhs || rhs per prover).
int gm_sumcheck_prove_batch(uint64_t transcript, const uint64_t* provers, size_t k, uint64_t* messages, uint64_t* challenges,
                            size_t cap_rounds, uint64_t* final_foldings, size_t* rounds_out) {
  T_CHECK(provers && messages && challenges && final_foldings && rounds_out && k >= 1, GM_EINVAL, "sumcheck_prove_batch: bad arguments");
  size_t rounds = 0;
  for (size_t j = 0; j < k; j++) {
    size_t t = 0;
    int rc = gm_sc_rounds(provers[j], &t, nullptr);
    if (rc) return rc;
    if (t > rounds) rounds = t;
  }
  rounds += 1;  // "+1 to get the final foldings"
  T_CHECK(rounds <= cap_rounds, GM_EINVAL, "sumcheck_prove_batch: %zu rounds exceed capacity %zu", rounds, cap_rounds);
  std::vector<gmh::Fr> coeff(k);
  for (size_t j = 0; j < k; j++) {
    uint64_t c[4];
    int rc = gm_transcript_challenge_fr(transcript, (const uint8_t*)"batch-sumcheck", 14, c);
    if (rc) return rc;
    coeff[j] = gmh::Fr::from_limbs(c);
  }
  const uint64_t* vm = nullptr;
  // a prover that has run out contributes (f0 * g0, 0) in every later round: read its final foldings once
  std::vector<char> finished(k, 0), has(k, 0);
  std::vector<gmh::Fr> final_product(k);
  for (size_t r = 0; r < rounds; r++) {
    gmh::Fr ma = gmh::Fr::zero(), mb = gmh::Fr::zero();
    // the round of every live prover is enqueued before the first wait (the reference runs them on rayon threads, :85)
    {
      // ONE launch for the live provers of the round (gm_sc_round_begin_many)
      std::vector<uint64_t> live;
      std::vector<size_t> at;
      for (size_t j = 0; j < k; j++)
        if (!finished[j]) {
          live.push_back(provers[j]);
          at.push_back(j);
        }
      std::vector<int> hs(live.size(), 0);
      int rc = gm_sc_round_begin_many(live.data(), live.size(), vm, hs.data());
      if (rc) return rc;
      for (size_t t = 0; t < live.size(); t++) has[at[t]] = (char)hs[t];
    }
    for (size_t j = 0; j < k; j++) {
      gmh::Fr fa, fb;
      if (!finished[j] && has[j]) {
        uint64_t a[4], b[4];
        int rc = gm_sc_round_end(provers[j], a, b);
        if (rc) return rc;
        fa = gmh::Fr::from_limbs(a);
        fb = gmh::Fr::from_limbs(b);
      } else {
        if (!finished[j]) {
          uint64_t f0[4], g0[4];
          int hf = 0;
          int rc = gm_sc_final(provers[j], f0, g0, &hf);
          if (rc) return rc;
          T_CHECK(hf, GM_ESTATE, "If next_message is None, we expect final foldings to be available");
          final_product[j] = gmh::Fr::from_limbs(f0) * gmh::Fr::from_limbs(g0);
          finished[j] = 1;
        }
        fa = final_product[j];
        fb = gmh::Fr::zero();
      }
      ma = ma + fa * coeff[j];
      mb = mb + fb * coeff[j];
    }
    ma.to_limbs(messages + 8 * r);
    mb.to_limbs(messages + 8 * r + 4);
    int rc = gm_transcript_append_fr(transcript, (const uint8_t*)"evaluations", 11, messages + 8 * r, 2);
    if (rc) return rc;
    if ((rc = gm_transcript_challenge_fr(transcript, (const uint8_t*)"challenge", 9, challenges + 4 * r))) return rc;
    vm = challenges + 4 * r;
  }
  for (size_t j = 0; j < k; j++) {
    int has = 0;
    int rc = gm_sc_final(provers[j], final_foldings + 8 * j, final_foldings + 8 * j + 4, &has);
    if (rc) return rc;
    T_CHECK(has, GM_ESTATE, "sumcheck_prove_batch: final foldings unavailable for prover %zu", j);
    if ((rc = gm_transcript_append_fr(transcript, (const uint8_t*)"final-folding-lhs", 17, final_foldings + 8 * j, 1))) return rc;
    if ((rc = gm_transcript_append_fr(transcript, (const uint8_t*)"final-folding-rhs", 17, final_foldings + 8 * j + 4, 1))) return rc;
  }
  *rounds_out = rounds;
  return GM_OK;
}

}  // extern "C"

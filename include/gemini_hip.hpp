// gemini_hip.hpp -- header-only C++ host layer over the C ABI (gemini_hip.h), mirroring the
// reference's Rust interface for the hot path: same names, argument meaning and failure
// behaviour (the reference panics -> these throw gm::Error).  This is what a C++ embedder links;
// the Rust shim of INTEGRATION.md has the same shape.
//
//   gm::VariableBaseMSM::{msm, msm_unchecked, msm_bigint}   ark-ec 0.4.2 (in-tree: src/kzg/msm/variable_base.rs)
//   gm::ChunkedPippenger / gm::HashMapPippenger             src/kzg/msm/stream_pippenger.rs:143-271
//   gm::msm_chunks                                          src/kzg/space.rs:22-55
//   gm::CommitterKey::{commit, batch_commit}                src/kzg/time.rs:81-107
//   gm::TimeProver (trait Prover)                           src/subprotocols/sumcheck/prover.rs:30-45
//   gm::Transcript (GeminiTranscript over merlin)           src/transcript.rs:8-34
//   gm::Sumcheck::{prove, new_time}                         src/subprotocols/sumcheck/proof.rs:36-66,125-130
//   gm::R1cs, gm::SnarkProof::{new_time, new_elastic}       src/circuit.rs, src/snark/time_prover.rs:19-117, elastic_prover.rs:174-266
//   gm::dist::{init_rccl, init_shm, init_hook, ...}         the all-gather between the per-GPU processes (no reference counterpart:
//                                                           the reference is single-device; gemini_amd/csrc/dist.cpp)
//   gm::CommitterKey::cyclic_share                          CommitterKey::new (src/kzg/time.rs:49-72), every rank its powers i = rank (mod world)
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gemini_hip.h"

namespace gm {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc != GM_OK) throw Error(rc, gm_last_error());
}
inline void init(int device = 0) { check(gm_init(device)); }
// give the prefix tables of every committer key back (the library does so by itself when a device allocation fails twice)
inline void release_spare_tables() { check(gm_g1_release_spare_tables()); }
// The library's device-memory bookkeeping (gm_mem_stats) and the footprint contract: what a proof will allocate, before it starts
// (the reference's memory story is its constants, README.md:38-46; a prover that keeps its vectors resident owes the number)
struct MemStats {
  uint64_t device_total, device_free, held, held_peak, pool_cached, in_use, in_use_peak, tables, keys, spare_table_releases, msm_workspaces, reserved;
};
inline MemStats mem_stats() {
  uint64_t v[12];
  check(gm_mem_stats(v));
  return MemStats{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
}
inline void mem_reset_peak() { check(gm_mem_reset_peak()); }
struct Footprint {
  uint64_t vectors, workspaces_to_grow, needed, available;
};
inline Footprint snark_footprint(uint64_t ck_handle, size_t num_constraints, bool elastic = false) {
  uint64_t v[4];
  check(gm_snark_footprint(ck_handle, num_constraints, elastic ? 1 : 0, v));
  return Footprint{v[0], v[1], v[2], v[3]};
}
inline Footprint psnark_footprint(uint64_t ck_handle, size_t num_variables, size_t nnz, int elastic = 0) {  // 0 time, 1 elastic (resident), 2 literal
  uint64_t v[4];
  check(gm_psnark_footprint(ck_handle, num_variables, nnz, elastic, v));
  return Footprint{v[0], v[1], v[2], v[3]};
}

// one rank of gm_psnark_new_time_sharded (block = psnark_shard_block(longest vector, world)); z and the instance's blocks are the caller's
inline size_t psnark_shard_block(size_t longest, int world) { return gm_psnark_shard_block(longest, world); }
inline size_t psnark_shard_level(size_t family_len, size_t block, size_t tail_log, int world) { return gm_psnark_shard_level(family_len, block, tail_log, world); }
inline Footprint psnark_shard_footprint(uint64_t key_handle, size_t num_constraints, size_t num_variables, size_t nnz, size_t block, int world) {
  uint64_t v[4];
  check(gm_psnark_shard_footprint(key_handle, num_constraints, num_variables, nnz, block, world, 0, v));
  return Footprint{v[0], v[1], v[2], v[3]};
}

// One process per GPU: after gm::init(local_rank) pick the transport of the library's all-gathers.  Every native prover
// (SnarkProof::new_time ..., gm_snark_new_time_sharded) then runs on N GPUs when its key is a cyclic share / a shard key.
namespace dist {
inline std::array<uint8_t, 128> rccl_unique_id() {  // rank 0; the 128 bytes reach the peers out of band
  std::array<uint8_t, 128> id{};
  check(gm_dist_rccl_unique_id(id.data()));
  return id;
}
inline void init_rccl(int rank, int world, const std::array<uint8_t, 128>& id) { check(gm_dist_init_rccl(rank, world, id.data())); }
inline void init_rccl_node(int rank, int world, const std::string& name) { check(gm_dist_init_rccl_node(rank, world, name.c_str())); }
inline void init_shm(int rank, int world, const std::string& name, size_t slot_bytes = 0) { check(gm_dist_init_shm(rank, world, name.c_str(), slot_bytes)); }
inline void init_hook(int rank, int world, gm_allgather_fn fn, void* ctx) { check(gm_dist_init_hook(rank, world, fn, ctx)); }
inline void selftest() { check(gm_dist_selftest()); }
inline void finalize() { check(gm_dist_finalize()); }
// this rank failed OUTSIDE a collective: release the peers waiting for it (they get GM_ESTATE), poison this rank's transport (round 6)
inline void abort() { check(gm_dist_abort()); }
inline int rank() {
  int r = 0;
  check(gm_dist_info(&r, nullptr, nullptr));
  return r;
}
inline int world() {
  int w = 1;
  check(gm_dist_info(nullptr, &w, nullptr));
  return w;
}
template <class T>
inline std::vector<T> allgather(const T& mine) {  // trivially copyable host values, rank order
  std::vector<T> all((size_t)world());
  check(gm_dist_allgather_host(&mine, sizeof(T), all.data()));
  return all;
}
}  // namespace dist

using Fr = std::array<uint64_t, 4>;        // Montgomery limbs (ark-ff memory image)
using BigInt = std::array<uint64_t, 4>;    // canonical integer (Fr::into_bigint)
using G1Projective = std::array<uint64_t, 18>;  // Jacobian X, Y, Z
struct G1Affine {                          // Rust layout: x, y, infinity (104-byte stride)
  uint64_t x[6], y[6];
  uint64_t infinity;  // bool at byte 96, padded
};
static_assert(sizeof(G1Affine) == 104, "G1Affine must match the 104-byte Rust record");

inline G1Projective g1_zero() {
  G1Projective z;
  check(gm_g1_sum(nullptr, 0, z.data()));
  return z;
}
inline G1Projective g1_add(const G1Projective& a, const G1Projective& b) {
  uint64_t two[36];
  memcpy(two, a.data(), 144);
  memcpy(two + 18, b.data(), 144);
  G1Projective r;
  check(gm_g1_sum(two, 2, r.data()));
  return r;
}

struct VariableBaseMSM {
  static G1Projective msm_bigint(const std::vector<G1Affine>& bases, const std::vector<BigInt>& bigints) {
    size_t n = bases.size() < bigints.size() ? bases.size() : bigints.size();
    G1Projective out;
    check(gm_g1_msm(bases.data(), sizeof(G1Affine), n ? bigints[0].data() : nullptr, n, out.data()));
    return out;
  }
  // silently truncates to the shorter input (src/kzg/time.rs:82)
  static G1Projective msm_unchecked(const std::vector<G1Affine>& bases, const std::vector<Fr>& scalars) {
    size_t n = bases.size() < scalars.size() ? bases.size() : scalars.size();
    G1Projective out;
    if (n == 0) return g1_zero();
    uint64_t hb = 0, hv = 0;
    check(gm_g1_bases_register(bases.data(), sizeof(G1Affine), n, &hb));
    int rc = gm_fr_vec_alloc(n, &hv);
    if (!rc) rc = gm_fr_vec_upload(hv, 0, scalars[0].data(), n);
    if (!rc) rc = gm_g1_msm_v(hb, 0, 0, hv, 0, n, out.data());  // into_bigint happens on the device
    if (hv) gm_fr_vec_free(hv);
    gm_g1_bases_free(hb);
    check(rc);
    return out;
  }
  // Ok(result) or Err(min_len): std::pair<optional<result>, size_t>
  static std::pair<std::optional<G1Projective>, size_t> msm(const std::vector<G1Affine>& bases, const std::vector<Fr>& scalars) {
    if (bases.size() != scalars.size()) return {std::nullopt, bases.size() < scalars.size() ? bases.size() : scalars.size()};
    return {msm_unchecked(bases, scalars), 0};
  }
};

// src/kzg/msm/stream_pippenger.rs:209-271.  The buffer is the device slot of a gm_g1_msm_stream (chunk =
// max_msm_buffer pairs, two slots: the copy of one flush runs under the kernels of the previous one); pairs
// added one at a time are staged on the host and pushed in blocks, add_pairs pushes a whole block.
class ChunkedPippenger {
 public:
  explicit ChunkedPippenger(size_t max_msm_buffer) {
    check(gm_g1_msm_stream_new(max_msm_buffer < kMaxChunk ? max_msm_buffer : kMaxChunk, sizeof(G1Affine), 0, &h_));
    stage_ = max_msm_buffer < kStage ? max_msm_buffer : kStage;
    scalars_.reserve(stage_);
    bases_.reserve(stage_);
  }
  ChunkedPippenger(const ChunkedPippenger&) = delete;
  ChunkedPippenger& operator=(const ChunkedPippenger&) = delete;
  ChunkedPippenger(ChunkedPippenger&& o) noexcept : h_(o.h_), stage_(o.stage_), scalars_(std::move(o.scalars_)), bases_(std::move(o.bases_)) { o.h_ = 0; }
  ~ChunkedPippenger() {
    if (h_) gm_g1_msm_stream_free(h_);
  }
  static ChunkedPippenger with_size(size_t buf_size) { return ChunkedPippenger(buf_size); }
  void add(const G1Affine& base, const BigInt& scalar) {
    scalars_.push_back(scalar);
    bases_.push_back(base);
    if (scalars_.size() == stage_) push();
  }
  void add_pairs(const G1Affine* bases, const BigInt* scalars, size_t n) {
    push();
    if (n) check(gm_g1_msm_stream_add(h_, bases, scalars, n));
  }
  G1Projective finalize() {
    push();
    G1Projective r;
    check(gm_g1_msm_stream_finalize(h_, r.data(), nullptr));
    return r;
  }

 private:
  static constexpr size_t kMaxChunk = (size_t)1 << 26, kStage = (size_t)1 << 14;
  void push() {
    if (!scalars_.empty()) check(gm_g1_msm_stream_add(h_, bases_.data(), scalars_.data(), scalars_.size()));
    scalars_.clear();
    bases_.clear();
  }
  uint64_t h_ = 0;
  size_t stage_ = 0;
  std::vector<BigInt> scalars_;
  std::vector<G1Affine> bases_;
};

// src/kzg/space.rs:22-55: skip len(bases) - len(scalars) bases, then 2^20-pair MSMs, summed; the streams stay on the host
inline G1Projective msm_chunks(const std::vector<G1Affine>& bases_stream, const std::vector<Fr>& scalars_stream) {
  if (scalars_stream.size() > bases_stream.size()) throw Error(GM_EINVAL, "msm_chunks: bases not long enough");
  if (scalars_stream.empty()) return g1_zero();
  uint64_t h = 0;
  check(gm_g1_msm_stream_new((size_t)1 << 20, sizeof(G1Affine), 1, &h));
  int rc = gm_g1_msm_stream_add(h, bases_stream.data() + (bases_stream.size() - scalars_stream.size()), scalars_stream.data(), scalars_stream.size());
  G1Projective r;
  if (!rc) rc = gm_g1_msm_stream_finalize(h, r.data(), nullptr);
  gm_g1_msm_stream_free(h);
  check(rc);
  return r;
}

// src/kzg/msm/stream_pippenger.rs:143-206: scalars of equal bases are added in Fr before the MSM
class HashMapPippenger {
 public:
  explicit HashMapPippenger(size_t max_msm_buffer) : cap_(max_msm_buffer), result_(g1_zero()) {}
  void add(const G1Affine& base, const Fr& scalar) {
    Key k;
    memcpy(k.data(), &base, 104);
    auto it = buffer_.find(k);
    if (it == buffer_.end()) {
      buffer_.emplace(k, scalar);
    } else {
      it->second = fr_add(it->second, scalar);
    }
    if (buffer_.size() == cap_) flush();
  }
  G1Projective finalize() {
    if (!buffer_.empty()) flush();
    return result_;
  }

 private:
  using Key = std::array<uint64_t, 13>;
  static Fr fr_add(const Fr& a, const Fr& b) {  // modular addition of Montgomery residues
    static const uint64_t R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    Fr r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (unsigned __int128)a[i] + b[i];
      r[i] = (uint64_t)c;
      c >>= 64;
    }
    bool ge = c != 0;
    if (!ge) {
      ge = true;
      for (int i = 3; i >= 0; i--) {
        if (r[i] != R[i]) {
          ge = r[i] > R[i];
          break;
        }
      }
    }
    if (ge) {
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)r[i] - R[i] - (uint64_t)br;
        r[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
    }
    return r;
  }
  void flush() {
    std::vector<G1Affine> bases;
    std::vector<Fr> scalars;
    for (auto& kv : buffer_) {
      G1Affine b;
      memcpy(&b, kv.first.data(), 104);
      bases.push_back(b);
      scalars.push_back(kv.second);
    }
    result_ = g1_add(result_, VariableBaseMSM::msm_unchecked(bases, scalars));
    buffer_.clear();
  }
  size_t cap_;
  std::map<Key, Fr> buffer_;
  G1Projective result_;
};

// SRS resident in HBM + the commit path of src/kzg/time.rs:81-107
class CommitterKey {
 public:
  explicit CommitterKey(const std::vector<G1Affine>& powers_of_g) : n_(powers_of_g.size()) {
    check(gm_g1_bases_register(powers_of_g.data(), sizeof(G1Affine), n_, &h_));
    check(gm_g1_bases_precompute(h_, -1));  // a committer key stays resident: fixed-base tables when they fit (gm_set_auto_tables)
  }
  // this rank's ELEMENT-CYCLIC share of CommitterKey::new(max_degree, ..) with trapdoor tau (canonical) and generator g: powers
  // i = rank (mod world), generated on the device; commit / batch_commit and every prover handed this key then shard their MSMs
  // over the ranks (gm::dist::init_* first).  One rank: the whole key.
  static CommitterKey cyclic_share(const uint64_t g_affine[12], const BigInt& tau, size_t max_degree) {
    CommitterKey k;
    k.n_ = max_degree + 1;
    check(gm_g1_srs_register_cyclic(g_affine, tau.data(), k.n_, dist::rank(), dist::world(), &k.h_));
    return k;
  }
  CommitterKey(CommitterKey&& o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = 0; }
  ~CommitterKey() {
    if (h_) gm_g1_bases_free(h_);
  }
  CommitterKey(const CommitterKey&) = delete;
  CommitterKey& operator=(const CommitterKey&) = delete;
  uint64_t handle() const { return h_; }
  G1Projective commit(const std::vector<Fr>& polynomial) const {
    size_t n = polynomial.size() < n_ ? polynomial.size() : n_;
    if (n == 0) return g1_zero();
    uint64_t hv = 0;
    check(gm_fr_vec_alloc(n, &hv));
    G1Projective out;
    int rc = gm_fr_vec_upload(hv, 0, polynomial[0].data(), n);
    if (!rc) rc = gm_ck_msm(h_, 0, 0, hv, 0, n, out.data());  // a whole key: gm_g1_msm_v; a cyclic share: local MSM + all-gather
    gm_fr_vec_free(hv);
    check(rc);
    return out;
  }
  // src/kzg/time.rs:98-107: ONE pipelined call (gm_g1_msm_v_batch: two calls in flight, the host tail of one under the kernels of
  // the next, small ones on four lanes side by side) instead of a loop of commits
  std::vector<G1Projective> batch_commit(const std::vector<std::vector<Fr>>& polynomials) const {
    const size_t k = polynomials.size();
    std::vector<G1Projective> out(k, g1_zero());
    std::vector<uint64_t> hv(k, 0);
    std::vector<size_t> ns(k, 0);
    int rc = GM_OK;
    for (size_t j = 0; j < k && !rc; j++) {
      ns[j] = polynomials[j].size() < n_ ? polynomials[j].size() : n_;
      rc = gm_fr_vec_alloc(ns[j], &hv[j]);
      if (!rc && ns[j]) rc = gm_fr_vec_upload(hv[j], 0, polynomials[j][0].data(), ns[j]);
    }
    if (!rc && k) rc = gm_ck_msm_batch(h_, hv.data(), ns.data(), k, out[0].data());
    for (uint64_t h : hv)
      if (h) gm_fr_vec_free(h);
    check(rc);
    return out;
  }

 private:
  CommitterKey() = default;
  uint64_t h_ = 0;
  size_t n_ = 0;  // powers of the WHOLE key
};

struct RoundMsg {
  Fr a, b;
};

// trait Prover, src/subprotocols/sumcheck/prover.rs:30-45 / time_prover.rs:42-137
class TimeProver {
 public:
  TimeProver(const std::vector<Fr>& f, const std::vector<Fr>& g, const Fr& twist) {
    check(gm_sc_new(f.empty() ? nullptr : f[0].data(), f.size(), g.empty() ? nullptr : g[0].data(), g.size(), twist.data(), &h_));
  }
  ~TimeProver() {
    if (h_) gm_sc_free(h_);
  }
  TimeProver(const TimeProver&) = delete;
  TimeProver& operator=(const TimeProver&) = delete;
  std::optional<RoundMsg> next_message(const std::optional<Fr>& verifier_message) {
    RoundMsg m;
    int has = 0;
    check(gm_sc_round(h_, verifier_message ? verifier_message->data() : nullptr, m.a.data(), m.b.data(), &has));
    if (!has) return std::nullopt;
    return m;
  }
  void fold(const Fr& challenge) { check(gm_sc_fold(h_, challenge.data())); }
  size_t rounds() const {
    size_t t = 0;
    check(gm_sc_rounds(h_, &t, nullptr));
    return t;
  }
  size_t round() const {
    size_t r = 0;
    check(gm_sc_rounds(h_, nullptr, &r));
    return r;
  }
  std::optional<std::array<Fr, 2>> final_foldings() const {
    std::array<Fr, 2> ff;
    int has = 0;
    check(gm_sc_final(h_, ff[0].data(), ff[1].data(), &has));
    if (!has) return std::nullopt;
    return ff;
  }
  uint64_t handle() const { return h_; }

 private:
  uint64_t h_ = 0;
};

// merlin::Transcript + GeminiTranscript, src/transcript.rs
class Transcript {
 public:
  explicit Transcript(const std::string& label = "GEMINI-v0") { check(gm_transcript_new((const uint8_t*)label.data(), label.size(), &h_)); }
  ~Transcript() {
    if (h_) gm_transcript_free(h_);
  }
  Transcript(const Transcript&) = delete;
  Transcript& operator=(const Transcript&) = delete;
  void append_message(const std::string& label, const std::vector<uint8_t>& msg) {
    check(gm_transcript_append_message(h_, (const uint8_t*)label.data(), label.size(), msg.data(), msg.size()));
  }
  void append_serializable(const std::string& label, const Fr& x) { check(gm_transcript_append_fr(h_, (const uint8_t*)label.data(), label.size(), x.data(), 1)); }
  void append_serializable(const std::string& label, const RoundMsg& m) {
    uint64_t ab[8];
    memcpy(ab, m.a.data(), 32);
    memcpy(ab + 4, m.b.data(), 32);
    check(gm_transcript_append_fr(h_, (const uint8_t*)label.data(), label.size(), ab, 2));
  }
  void append_serializable(const std::string& label, const G1Projective& c) {
    check(gm_transcript_append_g1(h_, (const uint8_t*)label.data(), label.size(), c.data(), 1, 0));
  }
  Fr get_challenge(const std::string& label) {
    Fr r;
    check(gm_transcript_challenge_fr(h_, (const uint8_t*)label.data(), label.size(), r.data()));
    return r;
  }
  uint64_t handle() const { return h_; }

 private:
  uint64_t h_ = 0;
};

// src/subprotocols/sumcheck/proof.rs:19-66,125-130
struct Sumcheck {
  std::vector<RoundMsg> messages;
  std::vector<Fr> challenges;
  size_t rounds = 0;
  std::vector<std::array<Fr, 2>> final_foldings;

  static Sumcheck prove(Transcript& transcript, TimeProver& prover) {
    Sumcheck s;
    std::optional<Fr> verifier_message;
    while (auto message = prover.next_message(verifier_message)) {
      transcript.append_serializable("evaluations", *message);
      Fr challenge = transcript.get_challenge("challenge");
      verifier_message = challenge;
      s.messages.push_back(*message);
      s.challenges.push_back(challenge);
    }
    s.rounds = prover.rounds();
    auto ff = prover.final_foldings();
    if (!ff) throw Error(GM_ESTATE, "final foldings unavailable");
    s.final_foldings.push_back(*ff);
    transcript.append_serializable("final-folding", (*ff)[0]);
    transcript.append_serializable("final-folding", (*ff)[1]);
    return s;
  }
  static Sumcheck new_time(Transcript& transcript, const std::vector<Fr>& f, const std::vector<Fr>& g, const Fr& twist) {
    TimeProver prover(f, g, twist);
    return prove(transcript, prover);
  }
};

// `Matrix<F> = Vec<Vec<(F, usize)>>` (src/circuit.rs:43) resident in HBM as CSR, together with its transpose (the
// prover needs column sums, src/snark/time_prover.rs:63-81)
using Matrix = std::vector<std::vector<std::pair<Fr, size_t>>>;
class DeviceMatrix {
 public:
  DeviceMatrix(const Matrix& rows, size_t ncols, bool transpose) {
    const size_t out_rows = transpose ? ncols : rows.size(), out_cols = transpose ? rows.size() : ncols;
    std::vector<uint64_t> rowptr(out_rows + 1, 0);
    for (size_t i = 0; i < rows.size(); i++)
      for (auto& e : rows[i]) rowptr[(transpose ? e.second : i) + 1]++;
    for (size_t i = 0; i < out_rows; i++) rowptr[i + 1] += rowptr[i];
    const size_t nnz = rowptr[out_rows];
    std::vector<uint32_t> cols(nnz);
    std::vector<Fr> vals(nnz);
    std::vector<uint64_t> cur(rowptr.begin(), rowptr.end() - 1);
    for (size_t i = 0; i < rows.size(); i++)
      for (auto& e : rows[i]) {
        const size_t r = transpose ? e.second : i, c = transpose ? i : e.second;
        cols[cur[r]] = (uint32_t)c;
        vals[cur[r]++] = e.first;
      }
    check(gm_spm_register(rowptr.data(), cols.data(), nnz ? vals[0].data() : nullptr, out_rows, out_cols, nnz, &h_));
  }
  ~DeviceMatrix() {
    if (h_) gm_spm_free(h_);
  }
  DeviceMatrix(const DeviceMatrix&) = delete;
  DeviceMatrix& operator=(const DeviceMatrix&) = delete;
  uint64_t handle() const { return h_; }

 private:
  uint64_t h_ = 0;
};

// src/circuit.rs `R1cs { a, b, c, z, w, x }` with everything the prover touches on the device
class R1cs {
 public:
  R1cs(const Matrix& a, const Matrix& b, const Matrix& c, const std::vector<Fr>& z, const std::vector<Fr>& w)
      : a_(a, z.size(), false), b_(b, z.size(), false), c_(c, z.size(), false), at_(a, z.size(), true), bt_(b, z.size(), true),
        ct_(c, z.size(), true), nz_(z.size()) {
    check(gm_fr_vec_alloc(z.size(), &z_));
    check(gm_fr_vec_alloc(w.size(), &w_));
    if (!z.empty()) check(gm_fr_vec_upload(z_, 0, z[0].data(), z.size()));
    if (!w.empty()) check(gm_fr_vec_upload(w_, 0, w[0].data(), w.size()));
  }
  ~R1cs() {
    if (z_) gm_fr_vec_free(z_);
    if (w_) gm_fr_vec_free(w_);
  }
  R1cs(const R1cs&) = delete;
  R1cs& operator=(const R1cs&) = delete;

 private:
  friend struct SnarkProof;
  friend class PsnarkInstance;
  friend struct PsnarkProof;
  DeviceMatrix a_, b_, c_, at_, bt_, ct_;
  uint64_t z_ = 0, w_ = 0;
  size_t nz_;
};

// src/subprotocols/tensorcheck/mod.rs:110-121
struct TensorcheckProof {
  std::vector<G1Projective> folded_polynomials_commitments;
  std::vector<std::array<Fr, 2>> folded_polynomials_evaluations;
  G1Projective evaluation_proof;
  std::vector<std::array<Fr, 3>> base_polynomials_evaluations;
};

// src/snark/mod.rs:76-82 + Proof::new_time (src/snark/time_prover.rs:19-117): one call into the library
struct SnarkProof {
  G1Projective witness_commitment;
  Fr zc_alpha;
  std::vector<RoundMsg> first_sumcheck_msgs, second_sumcheck_msgs;
  std::array<Fr, 2> first_final_foldings, second_final_foldings;
  TensorcheckProof tensorcheck_proof;

  static SnarkProof new_time(const R1cs& r1cs, const CommitterKey& ck, int g1_encoding = 0) {
    Buffers b(r1cs.nz_);
    const uint64_t mats[6] = {r1cs.a_.handle(), r1cs.b_.handle(), r1cs.c_.handle(), r1cs.at_.handle(), r1cs.bt_.handle(), r1cs.ct_.handle()};
    check(gm_snark_new_time(mats, r1cs.z_, r1cs.w_, ck.handle(), g1_encoding, b.cap, &b.p));
    return b.unpack();
  }

  // Proof::new_elastic(r1cs_stream, ck_stream, max_msm_buffer) (src/snark/elastic_prover.rs:174-266): the streams of
  // R1csStream (`Reverse(..)` of z, w, A z, B z, C z: src/snark/tests.rs:38-52) are built here as reversed device vectors,
  // outside the prover like the reference's stream construction; the proof equals new_time's (src/snark/tests.rs:56)
  static SnarkProof new_elastic(const R1cs& r1cs, const CommitterKey& ck, size_t max_msm_buffer = (size_t)1 << 20,
                                size_t min_device_chunk = (size_t)1 << 26, int g1_encoding = 0) {
    struct Vec {
      uint64_t h = 0;
      explicit Vec(size_t n) { check(gm_fr_vec_alloc(n, &h)); }
      ~Vec() {
        if (h) gm_fr_vec_free(h);
      }
    };
    const size_t nz = r1cs.nz_;
    size_t nw = 0;
    check(gm_fr_vec_len(r1cs.w_, &nw));
    Vec z_be(nz), w_be(nw), tmp(nz), za(nz), zb(nz), zc(nz);
    check(gm_fr_reverse(r1cs.z_, z_be.h));
    check(gm_fr_reverse(r1cs.w_, w_be.h));
    const DeviceMatrix* m[3] = {&r1cs.a_, &r1cs.b_, &r1cs.c_};
    Vec* out[3] = {&za, &zb, &zc};
    for (int k = 0; k < 3; k++) {
      check(gm_spm_mul(m[k]->handle(), r1cs.z_, tmp.h));
      check(gm_fr_reverse(tmp.h, out[k]->h));
    }
    Buffers b(nz);
    const uint64_t mats_t[3] = {r1cs.at_.handle(), r1cs.bt_.handle(), r1cs.ct_.handle()};
    check(gm_snark_new_elastic(mats_t, z_be.h, w_be.h, za.h, zb.h, zc.h, ck.handle(), max_msm_buffer, min_device_chunk, g1_encoding, b.cap, &b.p));
    return b.unpack();
  }

 private:
  // the plain-C proof record of the library and its caller-owned arrays
  struct Buffers {
    size_t cap = 2;
    std::vector<uint64_t> m0, m1, fc, fe;
    gm_snark_proof p;
    explicit Buffers(size_t nz) {
      for (size_t n = nz; n > 1; n = (n + 1) / 2) cap++;
      m0.resize(cap * 8);
      m1.resize(cap * 8);
      fc.resize(cap * 18);
      fe.resize(cap * 8);
      memset(&p, 0, sizeof p);
      p.messages[0] = m0.data();
      p.messages[1] = m1.data();
      p.fold_commitments = fc.data();
      p.fold_evaluations = fe.data();
    }
    SnarkProof unpack() const {
      SnarkProof out;
      memcpy(out.witness_commitment.data(), p.witness_commitment, 144);
      memcpy(out.zc_alpha.data(), p.zc_alpha, 32);
      auto msgs = [](const uint64_t* m, size_t rounds) {
        std::vector<RoundMsg> v(rounds);
        for (size_t i = 0; i < rounds; i++) {
          memcpy(v[i].a.data(), m + 8 * i, 32);
          memcpy(v[i].b.data(), m + 8 * i + 4, 32);
        }
        return v;
      };
      out.first_sumcheck_msgs = msgs(m0.data(), p.rounds[0]);
      out.second_sumcheck_msgs = msgs(m1.data(), p.rounds[1]);
      memcpy(out.first_final_foldings.data(), p.final_foldings[0], 64);
      memcpy(out.second_final_foldings.data(), p.final_foldings[1], 64);
      auto& tc = out.tensorcheck_proof;
      tc.folded_polynomials_commitments.resize(p.nfold);
      tc.folded_polynomials_evaluations.resize(p.nfold);
      for (size_t i = 0; i < p.nfold; i++) {
        memcpy(tc.folded_polynomials_commitments[i].data(), fc.data() + 18 * i, 144);
        memcpy(tc.folded_polynomials_evaluations[i].data(), fe.data() + 8 * i, 64);
      }
      memcpy(tc.evaluation_proof.data(), p.evaluation_proof, 144);
      tc.base_polynomials_evaluations.resize(1);
      memcpy(tc.base_polynomials_evaluations[0].data(), p.base_evaluations, 96);
      return out;
    }
  };
};

// The preprocessing SNARK (src/psnark/mod.rs:29-51): the matrix-only part of the instance (`sum_matrices`, `joint_matrices`, the
// lookup frequencies; src/misc.rs:269-366) is built inside the library once per circuit and stays in HBM; `index` and `new_time`
// are one call each (src/psnark/time_prover.rs:49-64, 69-384).
class PsnarkInstance {
 public:
  explicit PsnarkInstance(const R1cs& r1cs) : r1cs_(r1cs) {
    memset(&rec_, 0, sizeof rec_);
    check(gm_psnark_preprocess(r1cs.a_.handle(), r1cs.b_.handle(), r1cs.c_.handle(), r1cs.nz_, &rec_));
    rec_.z = r1cs.z_;
    rec_.w = r1cs.w_;
  }
  ~PsnarkInstance() { gm_psnark_preprocess_free(&rec_); }
  PsnarkInstance(const PsnarkInstance&) = delete;
  PsnarkInstance& operator=(const PsnarkInstance&) = delete;
  size_t num_non_zero() const { return rec_.nnz; }
  // Proof::index: commitments to row, col, val_a, val_b, val_c
  std::array<G1Projective, 5> index(const CommitterKey& ck) const {
    std::array<G1Projective, 5> out;
    check(gm_psnark_index(&rec_, ck.handle(), out[0].data()));
    return out;
  }

 private:
  friend struct PsnarkProof;
  const R1cs& r1cs_;
  gm_psnark_instance rec_;
};

struct PsnarkProof {
  G1Projective witness_commitment;
  Fr zc_alpha;
  std::vector<RoundMsg> first_sumcheck_msgs, second_sumcheck_msgs, third_sumcheck_msgs;
  std::array<Fr, 2> first_final_foldings, second_final_foldings;
  std::array<std::array<Fr, 2>, 13> third_final_foldings;
  std::array<G1Projective, 3> r_star_commitments;
  G1Projective z_star_commitment;
  std::array<G1Projective, 3> sorted_commitments;  // r, alpha, z
  std::array<Fr, 9> products;                      // r (set, subset, sorted), alpha (..), z (..)
  std::array<G1Projective, 9> acc_v_commitments;
  std::array<Fr, 9> claimed_sumchecks;
  std::array<Fr, 10> ralpha_star_acc_mu_evals;
  G1Projective ralpha_star_acc_mu_proof;
  std::array<Fr, 2> rstars_vals;
  TensorcheckProof tensorcheck_proof;

  // `ck_g2_bytes`: the serialised G2 powers of the key as the transcript absorbs them (src/psnark/time_prover.rs:86-88)
  static PsnarkProof new_time(const CommitterKey& ck, const PsnarkInstance& inst, const std::array<G1Projective, 5>& index,
                              const std::vector<uint8_t>& ck_g2_bytes, int g1_encoding = 0) {
    gm_psnark_instance I = inst.rec_;
    I.index_commitments = index[0].data();
    I.ck_g2_bytes = ck_g2_bytes.data();
    I.ck_g2_len = ck_g2_bytes.size();
    size_t cap = 4;
    for (size_t n = 2 * (inst.r1cs_.nz_ > I.nnz ? inst.r1cs_.nz_ : I.nnz) + 4; n > 1; n = (n + 1) / 2) cap++;
    std::vector<uint64_t> m[3], fc(4 * cap * 18), fe(4 * cap * 8);
    gm_psnark_proof p;
    memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; k++) {
      m[k].resize(cap * 8);
      p.messages[k] = m[k].data();
    }
    p.cap_folds = 4 * cap;
    p.fold_commitments = fc.data();
    p.fold_evaluations = fe.data();
    check(gm_psnark_new_time(&I, ck.handle(), g1_encoding, cap, &p));
    PsnarkProof out;
    auto msgs = [](const std::vector<uint64_t>& mm, size_t rounds) {
      std::vector<RoundMsg> v(rounds);
      for (size_t i = 0; i < rounds; i++) {
        memcpy(v[i].a.data(), mm.data() + 8 * i, 32);
        memcpy(v[i].b.data(), mm.data() + 8 * i + 4, 32);
      }
      return v;
    };
    memcpy(out.witness_commitment.data(), p.witness_commitment, 144);
    memcpy(out.zc_alpha.data(), p.zc_alpha, 32);
    out.first_sumcheck_msgs = msgs(m[0], p.rounds[0]);
    out.second_sumcheck_msgs = msgs(m[1], p.rounds[1]);
    out.third_sumcheck_msgs = msgs(m[2], p.rounds[2]);
    memcpy(out.first_final_foldings.data(), p.final_foldings[0], 64);
    memcpy(out.second_final_foldings.data(), p.final_foldings[1], 64);
    memcpy(out.third_final_foldings.data(), p.third_final_foldings, sizeof p.third_final_foldings);
    memcpy(out.r_star_commitments.data(), p.r_star_commitments, sizeof p.r_star_commitments);
    memcpy(out.z_star_commitment.data(), p.z_star_commitment, 144);
    memcpy(out.sorted_commitments.data(), p.sorted_commitments, sizeof p.sorted_commitments);
    memcpy(out.products.data(), p.products, sizeof p.products);
    memcpy(out.acc_v_commitments.data(), p.acc_v_commitments, sizeof p.acc_v_commitments);
    memcpy(out.claimed_sumchecks.data(), p.claimed_sumchecks, sizeof p.claimed_sumchecks);
    memcpy(out.ralpha_star_acc_mu_evals.data(), p.ralpha_star_acc_mu_evals, sizeof p.ralpha_star_acc_mu_evals);
    memcpy(out.ralpha_star_acc_mu_proof.data(), p.ralpha_star_acc_mu_proof, 144);
    memcpy(out.rstars_vals.data(), p.rstars_vals, sizeof p.rstars_vals);
    TensorcheckProof& tc = out.tensorcheck_proof;
    tc.folded_polynomials_commitments.resize(p.nfold);
    tc.folded_polynomials_evaluations.resize(p.nfold);
    for (size_t i = 0; i < p.nfold; i++) {
      memcpy(tc.folded_polynomials_commitments[i].data(), fc.data() + 18 * i, 144);
      memcpy(tc.folded_polynomials_evaluations[i].data(), fe.data() + 8 * i, 64);
    }
    memcpy(tc.evaluation_proof.data(), p.evaluation_proof, 144);
    tc.base_polynomials_evaluations.resize(22);
    memcpy(tc.base_polynomials_evaluations[0].data(), p.base_evaluations, sizeof p.base_evaluations);
    return out;
  }
};

}  // namespace gm

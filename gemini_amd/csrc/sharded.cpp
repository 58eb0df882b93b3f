// The N-GPU provers compiled into the library, over the collective layer of dist.cpp (one process per GPU).
//
// The reference has no multi-device code.  What is sharded here is its loop structure:
//   * gm_ck_*                        CommitterKey::{commit, batch_commit} (src/kzg/time.rs:81-107) over a key sharded element-
//                                    cyclically (power i on rank i mod world): strided gather of the rank's scalars, local MSM,
//                                    ONE all-gather of k x 144 bytes, EC adds.  snark.cpp / psnark.cpp commit through these, so
//                                    gm_snark_new_time / _new_elastic / gm_psnark_new_time / _index run on N GPUs when the key
//                                    handle they are given is a cyclic share (gm_g1_bases_set_cyclic): MSMs sharded, field
//                                    arithmetic replicated.
//   * gm_sumcheck_prove_sharded      Sumcheck::prove (src/subprotocols/sumcheck/proof.rs:36-66) over contiguous blocks: per
//                                    round 64 bytes all-gathered and added mod r; short tails gathered and finished replicated.
//   * gm_snark_new_time_sharded      snark::Proof::new_time (src/snark/time_prover.rs:19-117) with EVERY vector block-sharded
//                                    (rank r holds elements [r m, (r + 1) m) of z_a, z_b, z_c, abc_tensored, the folding
//                                    levels) and the key in per-level block slices; general sparse matrices (row blocks with
//                                    global columns, product_matrix_vector src/misc.rs:100-110) or block-diagonal ones (local
//                                    columns).  gemini_amd/dist_prover.py is the same sequence in Python (the tests hold the
//                                    two, and the single-GPU prover, byte for byte equal).
#include <algorithm>
#include <utility>

#include "ctx.hpp"
#include "prover_common.hpp"

namespace {

using namespace gmprover;

size_t cyclic_count(size_t length, size_t rank, size_t world) { return length > rank ? (length - rank + world - 1) / world : 0; }

int dist_rank_world(int* rank, int* world) { return gm_dist_info(rank, world, nullptr); }

const uint64_t* identity_point() {
  static uint64_t id[18];
  static bool init = (gm_g1_sum(nullptr, 0, id), true);
  (void)init;
  return id;
}

// this rank's pairs of "vec[voffset + t] with power (offset + t) or (offset - t), t < n": the first t it owns, how many, the
// local index of that power in its share
struct CyclicCut {
  size_t t0, cnt, j0;
};
CyclicCut cyclic_cut(size_t offset, int reversed, size_t n, size_t rank, size_t world) {
  CyclicCut c;
  // forward: power offset + t = rank (mod world); reversed: power offset - t = rank (mod world)
  c.t0 = reversed ? (offset % world + world - rank) % world : (rank + world - offset % world) % world;
  c.cnt = cyclic_count(n, c.t0, world);
  const size_t power = reversed ? offset - c.t0 : offset + c.t0;
  c.j0 = c.cnt ? (power - rank) / world : 0;
  return c;
}

}  // namespace

extern "C" {

// ---- the committer key: plain, or this rank's cyclic share ------------------------------------------------------------
int gm_g1_bases_set_cyclic(uint64_t handle, size_t n_global, int rank, int world) {
  GM_CTX();
  gm::Bases* b = gm::find_bases(handle);
  GM_CHECK(b != nullptr, GM_EHANDLE, "bases_set_cyclic: unknown handle %llu", (unsigned long long)handle);
  if (world <= 1) {
    GM_CHECK(n_global == 0 || n_global == b->n, GM_EINVAL, "bases_set_cyclic: one rank holds the whole key (%zu != %zu)", n_global, b->n);
    b->cyclic_n = 0;
    b->cyc_rank = 0;
    b->cyc_world = 1;
    return GM_OK;
  }
  GM_CHECK(rank >= 0 && rank < world, GM_EINVAL, "bases_set_cyclic: rank %d of %d", rank, world);
  GM_CHECK(b->n == cyclic_count(n_global, (size_t)rank, (size_t)world), GM_EINVAL,
           "bases_set_cyclic: rank %d of %d holds %zu of %zu powers, this handle has %zu", rank, world,
           cyclic_count(n_global, (size_t)rank, (size_t)world), n_global, b->n);
  b->cyclic_n = n_global;
  b->cyc_rank = rank;
  b->cyc_world = world;
  return GM_OK;
}

// powers tau^i g, i = rank (mod world), i < n_global, generated on this rank's device: base tau^rank g, ratio tau^world
// (CommitterKey::new, src/kzg/time.rs:49-72, each rank its share)
int gm_g1_srs_register_cyclic(const uint64_t base_affine[12], const uint64_t tau[4], size_t n_global, int rank, int world, uint64_t* handle) {
  GM_CTX();
  GM_CHECK(base_affine && tau && handle && world >= 1 && rank >= 0 && rank < world, GM_EINVAL, "srs_register_cyclic: bad arguments");
  const Fr t = Fr::from_canonical(tau);
  uint64_t e[1] = {(uint64_t)rank}, w[1] = {(uint64_t)world};
  uint64_t t_rank[4], t_world[4];
  t.pow(e, 1).to_canonical(t_rank);
  t.pow(w, 1).to_canonical(t_world);
  uint64_t first = 0;
  RC(gm_g1_fixed_base_register(base_affine, t_rank, 1, &first));
  uint64_t base[12];
  int rc = gm_g1_bases_download(first, 0, 1, base);
  (void)gm_g1_bases_free(first);
  if (rc) return rc;
  RC(gm_g1_srs_register(base, t_world, cyclic_count(n_global, (size_t)rank, (size_t)world), handle));
  rc = gm_g1_bases_set_cyclic(*handle, n_global, rank, world);
  if (rc) (void)gm_g1_bases_free(*handle);
  return rc;
}

int gm_ck_len(uint64_t ck, size_t* n_global) {
  GM_CTX();
  gm::Bases* b = gm::find_bases(ck);
  GM_CHECK(b != nullptr && n_global != nullptr, GM_EHANDLE, "ck_len: unknown key handle %llu", (unsigned long long)ck);
  *n_global = b->cyclic_n ? b->cyclic_n : b->n;
  return GM_OK;
}

// k MSMs "vector j, elements [voffsets[j], +ns[j]) against powers offsets[j] + t (reversed: offsets[j] - t)" of the GLOBAL key.
// Plain key: the pipelined batch as it is.  Cyclic share: strided gathers, one pipelined batch of un-normalised local MSMs,
// one all-gather of k x 144 bytes, one normalisation per result -- identical bytes on every rank.
static int ck_msm_many(uint64_t ck, const size_t* offsets, int reversed, const uint64_t* vecs, const size_t* voffsets, const size_t* ns, size_t k,
                       uint64_t* out) {
  GM_CTX();
  gm::Bases* b = gm::find_bases(ck);
  GM_CHECK(b != nullptr, GM_EHANDLE, "ck_msm: unknown key handle %llu", (unsigned long long)ck);
  if (k == 0) return GM_OK;
  const size_t world = b->cyclic_n ? (size_t)b->cyc_world : 1, rank = (size_t)b->cyc_rank;
  if (world == 1) {
    bool plain0 = true;
    for (size_t j = 0; j < k; j++) plain0 = plain0 && voffsets[j] == 0;
    if (k == 1) return gm_g1_msm_v(ck, offsets[0], reversed, vecs[0], voffsets[0], ns[0], out);
    if (plain0) return gm_g1_msm_v_batch_at(ck, offsets, reversed, vecs, ns, k, 0, out);
    for (size_t j = 0; j < k; j++) RC(gm_g1_msm_v(ck, offsets[j], reversed, vecs[j], voffsets[j], ns[j], out + 18 * j));
    return GM_OK;
  }
  int drank = 0, dworld = 1;
  RC(dist_rank_world(&drank, &dworld));
  GM_CHECK((size_t)dworld == world && (size_t)drank == rank, GM_ESTATE, "ck_msm: the key is share %zu of %zu but gm_dist is rank %d of %d", rank, world,
           drank, dworld);
  Vecs V;
  std::vector<uint64_t> mine;
  std::vector<size_t> cnts, offs, idx;
  for (size_t j = 0; j < k; j++) {
    GM_CHECK(reversed ? offsets[j] < b->cyclic_n && ns[j] <= offsets[j] + 1 : offsets[j] + ns[j] <= b->cyclic_n, GM_EINVAL,
             "ck_msm: %zu pairs from power %zu (%s) outside a key of %zu powers", ns[j], offsets[j], reversed ? "down" : "up", b->cyclic_n);
    const CyclicCut c = cyclic_cut(offsets[j], reversed, ns[j], rank, world);
    if (c.cnt == 0) continue;
    uint64_t s = 0;
    RC(V.alloc(c.cnt, &s));
    RC(gm_fr_stride(vecs[j], voffsets[j] + c.t0, world, c.cnt, s));
    mine.push_back(s);
    cnts.push_back(c.cnt);
    offs.push_back(c.j0);
    idx.push_back(j);
  }
  std::vector<uint64_t> parts(18 * k), local(18 * std::max<size_t>(mine.size(), 1));
  for (size_t j = 0; j < k; j++) memcpy(parts.data() + 18 * j, identity_point(), 144);
  if (!mine.empty()) {
    RC(gm_g1_msm_v_batch_at(ck, offs.data(), reversed, mine.data(), cnts.data(), mine.size(), 1, local.data()));
    for (size_t t = 0; t < idx.size(); t++) memcpy(parts.data() + 18 * idx[t], local.data() + 18 * t, 144);
  }
  std::vector<uint64_t> all(18 * k * world);
  RC(gm_dist_allgather_host_class(parts.data(), 144 * k, all.data(), GM_DIST_CLASS_G1));
  std::vector<uint64_t> col(18 * world);
  for (size_t j = 0; j < k; j++) {
    for (size_t r = 0; r < world; r++) memcpy(col.data() + 18 * r, all.data() + 18 * (r * k + j), 144);
    RC(gm_g1_sum(col.data(), world, out + 18 * j));
  }
  return GM_OK;
}

int gm_ck_msm(uint64_t ck, size_t offset, int reversed, uint64_t vec, size_t voffset, size_t n, uint64_t out[18]) {
  return ck_msm_many(ck, &offset, reversed, &vec, &voffset, &n, 1, out);
}

int gm_ck_msm_batch(uint64_t ck, const uint64_t* vecs, const size_t* ns, size_t k, uint64_t* out) {
  std::vector<size_t> zeros(k, 0);
  return ck_msm_many(ck, zeros.data(), 0, vecs, zeros.data(), ns, k, out);
}

// ---- Sumcheck::prove over blocks ---------------------------------------------------------------------------------------
// This rank holds elements [lo, lo + len) of f and g (global length n_global, lo even).  Rounds are shard-local while the blocks
// stay longer than TAIL elements and pair-aligned: the rank's partial (a, b) -- 64 bytes -- is all-gathered and summed mod r.
// Then the blocks are gathered once and every rank finishes the protocol on the whole (short) vectors.  Consumes nothing: f and
// g are copied into the prover (Sumcheck::new_time copies too, proof.rs:125-130).
int gm_sumcheck_prove_sharded(uint64_t transcript, uint64_t f_block, uint64_t g_block, const uint64_t twist[4], size_t lo, size_t n_global,
                              uint64_t* messages, uint64_t* challenges, size_t cap_rounds, uint64_t final_foldings[8], size_t* rounds_out) {
  GM_CTX();
  GM_CHECK(messages && challenges && final_foldings && rounds_out && twist, GM_EINVAL, "sumcheck_prove_sharded: null pointer");
  GM_CHECK(lo % 2 == 0, GM_EINVAL, "sumcheck_prove_sharded: the block must start at an even index (pairs fold together)");
  int rank = 0, world = 1;
  RC(dist_rank_world(&rank, &world));
  {
    // the blocks must tile [0, n_global) in rank order with equal lengths: anything else sends the ranks down different branches of
    // the round loop -- a deadlocked all-gather or, worse, messages that silently differ from the single-GPU prover's
    size_t nf0 = 0, ng0 = 0;
    RC(vec_len(f_block, &nf0));
    RC(vec_len(g_block, &ng0));
    GM_CHECK(nf0 == ng0, GM_EINVAL, "sumcheck_prove_sharded: the blocks of f and g hold %zu and %zu elements", nf0, ng0);
    GM_CHECK(world == 1 || (nf0 * (size_t)world == n_global && lo == (size_t)rank * nf0), GM_EINVAL,
             "sumcheck_prove_sharded: rank %d of %d holds [%zu, %zu) of %zu elements; equal blocks in rank order are required", rank, world, lo, lo + nf0,
             n_global);
    GM_CHECK(world > 1 || (lo == 0 && nf0 == n_global), GM_EINVAL, "sumcheck_prove_sharded: one rank holds the whole vectors (%zu of %zu from %zu)", nf0, n_global, lo);
    if (world > 1) {  // and every rank must see the same thing
      uint64_t mine[2] = {(uint64_t)nf0, (uint64_t)n_global};
      std::vector<uint64_t> all(2 * (size_t)world);
      RC(gm_dist_allgather_host(mine, sizeof mine, all.data()));
      for (int r2 = 0; r2 < world; r2++)
        GM_CHECK(all[2 * (size_t)r2] == mine[0] && all[2 * (size_t)r2 + 1] == mine[1], GM_EINVAL,
                 "sumcheck_prove_sharded: rank %d holds blocks of %llu of %llu elements, this rank %zu of %zu", r2, (unsigned long long)all[2 * (size_t)r2],
                 (unsigned long long)all[2 * (size_t)r2 + 1], nf0, n_global);
    }
  }
  constexpr size_t TAIL = (size_t)1 << 10;
  uint64_t prover = 0;
  RC(gm_sc_new_borrow(f_block, g_block, twist, &prover));
  struct Guard {
    uint64_t& p;
    ~Guard() {
      if (p) (void)gm_sc_free(p);
    }
  } guard{prover};
  RC(gm_sc_set_shard(prover, lo / 2));
  bool replicated = world == 1;
  size_t cur_n = n_global, k = 0;
  const uint64_t* vm = nullptr;
  for (;;) {
    if (!replicated) {
      const size_t per = cur_n / (size_t)world;
      if (!(per > TAIL && per % 4 == 0)) {
        // apply the pending fold shard-locally, then gather: the replicated prover starts exactly at a message boundary
        if (vm) {
          RC(gm_sc_fold(prover, vm));
          cur_n = (cur_n + 1) / 2;
          vm = nullptr;
        }
        size_t nf = 0, ng = 0;
        uint64_t tw[4];
        RC(gm_sc_lens(prover, &nf, &ng, tw));
        GM_CHECK(nf == ng, GM_ESTATE, "sumcheck_prove_sharded: blocks of different lengths (%zu, %zu)", nf, ng);
        std::vector<uint64_t> loc(8 * nf), all(8 * nf * (size_t)world);
        RC(gm_sc_download(prover, loc.data(), loc.data() + 4 * nf));
        RC(gm_dist_allgather_host(loc.data(), 64 * nf, all.data()));
        std::vector<uint64_t> fs(4 * nf * (size_t)world), gs(4 * nf * (size_t)world);
        for (size_t r = 0; r < (size_t)world; r++) {
          memcpy(fs.data() + 4 * nf * r, all.data() + 8 * nf * r, 32 * nf);
          memcpy(gs.data() + 4 * nf * r, all.data() + 8 * nf * r + 4 * nf, 32 * nf);
        }
        (void)gm_sc_free(prover);
        prover = 0;
        RC(gm_sc_new(fs.data(), nf * (size_t)world, gs.data(), nf * (size_t)world, tw, &prover));
        replicated = true;
      }
    }
    uint64_t a[4], b[4];
    int has = 0;
    RC(gm_sc_round(prover, vm, a, b, &has));
    if (vm) cur_n = (cur_n + 1) / 2;
    if (!has) break;
    GM_CHECK(k < cap_rounds, GM_EINVAL, "sumcheck_prove_sharded: more than %zu rounds", cap_rounds);
    if (!replicated) {
      uint64_t mine[8];
      memcpy(mine, a, 32);
      memcpy(mine + 4, b, 32);
      std::vector<uint64_t> all(8 * (size_t)world);
      RC(gm_dist_allgather_host(mine, 64, all.data()));
      Fr sa = Fr::zero(), sb = Fr::zero();
      for (int r = 0; r < world; r++) {
        sa = sa + Fr::from_limbs(all.data() + 8 * r);
        sb = sb + Fr::from_limbs(all.data() + 8 * r + 4);
      }
      sa.to_limbs(a);
      sb.to_limbs(b);
    }
    memcpy(messages + 8 * k, a, 32);
    memcpy(messages + 8 * k + 4, b, 32);
    RC(gm_transcript_append_fr(transcript, L("evaluations"), 11, messages + 8 * k, 2));
    RC(gm_transcript_challenge_fr(transcript, L("challenge"), 9, challenges + 4 * k));
    vm = challenges + 4 * k;
    k++;
  }
  int has = 0;
  RC(gm_sc_final(prover, final_foldings, final_foldings + 4, &has));
  GM_CHECK(has, GM_ESTATE, "sumcheck_prove_sharded: final foldings unavailable");
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings, 1));
  RC(gm_transcript_append_fr(transcript, L("final-folding"), 13, final_foldings + 4, 1));
  *rounds_out = k;
  return GM_OK;
}

// ---- the block-sharded key -----------------------------------------------------------------------------------------------
// Level j of the folding tree of a polynomial of n coefficients is block-sharded with blocks of m / 2^j (m = n / world) while
// those hold >= 2^tail_log elements (j <= jmax); rank r's slice of the key for level j is powers [r m / 2^j, (r + 1) m / 2^j).
// One registered key holds the slices of levels 0 .. jmax back to back, then the first n / 2^(jmax + 1) powers (the gathered
// levels, the same on every rank): segments = jmax + 2.
int gm_snark_shard_key_new(const uint64_t base_affine[12], const uint64_t tau[4], size_t n, size_t tail_log, uint64_t* key, size_t offsets[64],
                           size_t counts[64], size_t* segments) {
  GM_CTX();
  GM_CHECK(base_affine && tau && key && offsets && counts && segments, GM_EINVAL, "snark_shard_key_new: null pointer");
  int rank = 0, world = 1;
  RC(dist_rank_world(&rank, &world));
  GM_CHECK(n >= 2 && (n & (n - 1)) == 0 && (world & (world - 1)) == 0 && n % (size_t)world == 0, GM_EINVAL,
           "snark_shard_key_new: block sharding needs powers of two (n = %zu, %d ranks)", n, world);
  const size_t m = n / (size_t)world, tail = (size_t)1 << tail_log;
  GM_CHECK(tail_log >= 3 && tail_log < 40 && m >= tail, GM_EINVAL, "snark_shard_key_new: blocks of %zu elements are shorter than the tail length 2^%zu", m,
           tail_log);
  size_t jmax = 0;
  while ((m >> (jmax + 1)) >= tail) jmax++;
  size_t starts[64], at = 0;
  for (size_t j = 0; j <= jmax; j++) {
    starts[j] = (size_t)rank * (m >> j);
    counts[j] = m >> j;
  }
  starts[jmax + 1] = 0;
  counts[jmax + 1] = std::max<size_t>(n >> (jmax + 1), 1);
  for (size_t j = 0; j <= jmax + 1; j++) {
    offsets[j] = at;
    at += counts[j];
  }
  *segments = jmax + 2;
  return gm_g1_srs_register_segments(base_affine, tau, starts, counts, jmax + 2, key);
}

// the polynomial of degree < 3 through (xs[i], ys[i]): coefficients c0, c1, c2
static void interp3(const Fr xs[3], const Fr ys[3], Fr c[3]) {
  c[0] = c[1] = c[2] = Fr::zero();
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const Fr den = (xs[i] - xs[j]) * (xs[i] - xs[k]);
    const Fr s = ys[i] * den.inv();
    c[0] = c[0] + s * xs[j] * xs[k];
    c[1] = c[1] - s * (xs[j] + xs[k]);
    c[2] = c[2] + s;
  }
}

static Fr fr_pow_u64(const Fr& x, uint64_t e) { return x.pow(&e, 1); }

// GM_SHARD_TRACE=1: the phases of the sharded tensor check on stderr (rank 0)
struct PhaseTrace {
  bool on;
  Clock::time_point t;
  explicit PhaseTrace(bool enable) : on(enable), t(Clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto now = Clock::now();
    fprintf(stderr, "[gm shard] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

static int snark_new_time_sharded_impl(const gm_snark_shard* S, int g1_encoding, size_t cap_rounds, gm_snark_proof* P);
// (a failure on this rank -- outside a collective as well: a bad input, an allocation -- tells the peers instead of leaving them in their next all-gather)
int gm_snark_new_time_sharded(const gm_snark_shard* S, int g1_encoding, size_t cap_rounds, gm_snark_proof* P) {
  const int rc = snark_new_time_sharded_impl(S, g1_encoding, cap_rounds, P);
  if (rc) (void)gm_dist_abort();
  return rc;
}
static int snark_new_time_sharded_impl(const gm_snark_shard* S, int g1_encoding, size_t cap_rounds, gm_snark_proof* P) {
  GM_CTX();
  GM_CHECK(S && P && P->messages[0] && P->messages[1] && P->fold_commitments && P->fold_evaluations, GM_EINVAL, "snark_new_time_sharded: null pointer");
  const auto t_all = Clock::now();
  int rank_i = 0, world_i = 1;
  RC(dist_rank_world(&rank_i, &world_i));
  const size_t r = (size_t)rank_i, g = (size_t)world_i, n = S->n;
  GM_CHECK(n >= 2 && (n & (n - 1)) == 0 && (g & (g - 1)) == 0 && n % g == 0, GM_EINVAL, "snark_new_time_sharded: powers of two needed (n = %zu, %zu ranks)", n, g);
  const size_t m = n / g, tail = (size_t)1 << (S->tail_log < 40 ? S->tail_log : 39);
  GM_CHECK(S->tail_log >= 3 && S->tail_log < 40 && m >= tail, GM_EINVAL, "snark_new_time_sharded: blocks of %zu elements are shorter than the tail length 2^%zu (3 .. 39)", m,
           (size_t)S->tail_log);
  size_t jmax = 0;
  while ((m >> (jmax + 1)) >= tail) jmax++;
  GM_CHECK(S->key_segments == jmax + 2, GM_EINVAL, "snark_new_time_sharded: the key has %zu segments, the layout needs %zu", S->key_segments, jmax + 2);
  const size_t PREFIX = jmax + 1;
  size_t logn = 0;
  while (((size_t)1 << logn) < n) logn++;
  size_t logm = 0;
  while (((size_t)1 << logm) < m) logm++;
  // shapes: row blocks of A, B, C and of their transposes, m rows each; columns GLOBAL (n: general matrices, z whole on every
  // rank) or LOCAL (m: block-diagonal instance, z is this rank's block)
  size_t ncols = 0;
  for (int k = 0; k < 6; k++) {
    size_t rows = 0, cols = 0;
    RC(gm_spm_shape(S->matrices[k], &rows, &cols, nullptr));
    GM_CHECK(rows == m && (cols == n || cols == m) && (k == 0 || cols == ncols), GM_EINVAL,
             "snark_new_time_sharded: matrix %d is %zu x %zu, expected %zu x %zu (global columns) or %zu x %zu (block-diagonal)", k, rows, cols, m, n, m, m);
    ncols = cols;
  }
  const bool global_cols = ncols == n && g > 1;
  size_t nz = 0, nw = 0;
  RC(vec_len(S->z, &nz));
  RC(vec_len(S->w_block, &nw));
  GM_CHECK(nz == ncols, GM_EINVAL, "snark_new_time_sharded: z has %zu elements, the matrices %zu columns", nz, ncols);
  GM_CHECK(nw <= m, GM_EINVAL, "snark_new_time_sharded: the block of w has %zu elements, more than a block (%zu)", nw, m);
  Vecs V;
  // this rank's block of z
  uint64_t z_blk = S->z;
  if (global_cols) {
    RC(V.alloc(m, &z_blk));
    RC(gm_fr_stride(S->z, r * m, 1, m, z_blk));
  }
  uint64_t z_abc[3];
  for (int k = 0; k < 3; k++) {
    RC(V.alloc(m, &z_abc[k]));
    RC(gm_spm_mul(S->matrices[k], S->z, z_abc[k]));  // :32-34
  }
  TranscriptGuard T;
  static const char protocol[] = "GEMINI-v0";
  RC(gm_transcript_new(L(protocol), sizeof protocol - 1, &T.h));
  if (g1_encoding) RC(gm_transcript_set_g1_encoding(T.h, g1_encoding));
  P->spans[0] = since(t_all);

  // values p(x) = sum_rr x^(rr * L) P_rr(x) of block-sharded polynomials: local block evaluations, one all-gather
  // allr[(rr * k + i) * npts + q]
  auto eval_blocks = [&](const std::vector<uint64_t>& blocks, const uint64_t* pts, size_t npts, const std::vector<size_t>& blen, std::vector<uint64_t>& allr,
                         std::vector<Fr>& vals) -> int {
    const size_t k = blocks.size();
    std::vector<uint64_t> local(4 * k * npts);
    RC(gm_fr_eval_le_batch(blocks.data(), k, pts, npts, local.data()));
    allr.assign(4 * k * npts * g, 0);
    RC(gm_dist_allgather_host(local.data(), 32 * k * npts, allr.data()));
    vals.assign(k * npts, Fr::zero());
    for (size_t i = 0; i < k; i++)
      for (size_t q = 0; q < npts; q++) {
        const Fr x = Fr::from_limbs(pts + 4 * q);
        const Fr step = fr_pow_u64(x, blen[i]);
        Fr acc = Fr::zero(), xp = Fr::one();
        for (size_t rr = 0; rr < g; rr++) {
          acc = acc + xp * Fr::from_limbs(allr.data() + 4 * ((rr * k + i) * npts + q));
          xp = xp * step;
        }
        vals[i * npts + q] = acc;
      }
    return GM_OK;
  };
  // MSMs of vecs[i] against the key slice of level levels[i], one pipelined batch, un-normalised
  auto key_commit = [&](const std::vector<size_t>& levels, const std::vector<uint64_t>& vecs, uint64_t* out) -> int {
    std::vector<size_t> offs(levels.size()), ns(levels.size());
    for (size_t i = 0; i < levels.size(); i++) {
      size_t len = 0;
      RC(vec_len(vecs[i], &len));
      offs[i] = S->key_offsets[levels[i]];
      ns[i] = std::min(len, S->key_counts[levels[i]]);
    }
    return gm_g1_msm_v_batch_at(S->key, offs.data(), 0, vecs.data(), ns.data(), vecs.size(), 1, out);
  };
  auto gather_sum = [&](const uint64_t* parts, size_t k, uint64_t* out) -> int {  // all-gather k partial points, add per column
    std::vector<uint64_t> all(18 * k * g), col(18 * g);
    RC(gm_dist_allgather_host_class(parts, 144 * k, all.data(), GM_DIST_CLASS_G1));
    for (size_t j = 0; j < k; j++) {
      for (size_t rr = 0; rr < g; rr++) memcpy(col.data() + 18 * rr, all.data() + 18 * (rr * k + j), 144);
      RC(gm_g1_sum(col.data(), g, out + 18 * j));
    }
    return GM_OK;
  };

  auto t0 = Clock::now();
  {
    uint64_t part[18];
    RC(key_commit({0}, {S->w_block}, part));  // ck.commit(&r1cs.w) :42
    RC(gather_sum(part, 1, P->witness_commitment));
  }
  P->spans[1] = since(t0);
  RC(gm_transcript_append_g1(T.h, L("witness"), 7, P->witness_commitment, 1, 0));
  uint64_t alpha[4];
  RC(gm_transcript_challenge_fr(T.h, L("alpha"), 5, alpha));
  {
    std::vector<uint64_t> allr;
    std::vector<Fr> vals;
    RC(eval_blocks({z_abc[2]}, alpha, 1, {m}, allr, vals));  // :48
    vals[0].to_limbs(P->zc_alpha);
  }
  RC(gm_transcript_append_fr(T.h, L("zc(alpha)"), 9, P->zc_alpha, 1));

  t0 = Clock::now();
  std::vector<uint64_t> ch1(4 * cap_rounds), ch2(4 * cap_rounds);
  RC(gm_sumcheck_prove_sharded(T.h, z_abc[0], z_abc[1], alpha, r * m, n, P->messages[0], ch1.data(), cap_rounds, P->final_foldings[0], &P->rounds[0]));  // :52
  P->spans[2] = since(t0);

  t0 = Clock::now();
  GM_CHECK(P->rounds[0] == logn, GM_ESTATE, "snark_new_time_sharded: %zu rounds for 2^%zu elements", P->rounds[0], logn);
  uint64_t eta[4];
  uint64_t coeffs[12];
  uint64_t abc;
  {
    // tensor(rho), powers(alpha), their product: whole (general matrices: every rank needs every entry, and an O(n) pass at HBM
    // speed is cheaper than moving n elements over xGMI) or this rank's block -- block r of tensor(rho) is a scalar times
    // tensor(rho[:log m]), of powers(alpha) it is alpha^(r m) powers(alpha, m); the scalars go into the coefficients of the linear
    // combination below, not into passes of their own                                                                  :56-58
    const size_t len = global_cols ? n : m;
    uint64_t a_ch, b_ch, c_ch;
    RC(V.alloc(len, &b_ch));
    RC(gm_fr_tensor(ch1.data(), global_cols ? logn : logm, b_ch));
    RC(V.alloc(len, &c_ch));
    RC(gm_fr_powers(alpha, len, c_ch));
    RC(V.alloc(len, &a_ch));
    RC(gm_fr_hadamard(b_ch, c_ch, a_ch));
    RC(gm_transcript_challenge_fr(T.h, L("eta"), 3, eta));
    Fr s_b = Fr::one(), s_c = Fr::one();
    if (!global_cols) {
      for (size_t j = logm; j < logn; j++)
        if ((r >> (j - logm)) & 1) s_b = s_b * Fr::from_limbs(ch1.data() + 4 * j);
      s_c = fr_pow_u64(Fr::from_limbs(alpha), r * m);
    }
    const Fr e = Fr::from_limbs(eta);
    (s_b * s_c).to_limbs(coeffs);
    (e * s_b).to_limbs(coeffs + 4);
    (e * e * s_c).to_limbs(coeffs + 8);
    uint64_t t_abc[3];
    const uint64_t rand_vecs[3] = {a_ch, b_ch, c_ch};
    for (int k = 0; k < 3; k++) {
      RC(V.alloc(m, &t_abc[k]));
      RC(gm_spm_mul(S->matrices[3 + k], rand_vecs[k], t_abc[k]));  // :63-81
    }
    RC(V.alloc(m, &abc));
    RC(gm_fr_lincomb(t_abc, coeffs, 3, abc));
    RC(gm_fr_vec_set_len(abc, m));
    for (uint64_t v : {t_abc[0], t_abc[1], t_abc[2], a_ch, b_ch, c_ch}) V.release(v);
  }
  P->spans[3] = since(t0);

  t0 = Clock::now();
  uint64_t one[4];
  Fr::one().to_limbs(one);
  RC(gm_sumcheck_prove_sharded(T.h, abc, z_blk, one, r * m, n, P->messages[1], ch2.data(), cap_rounds, P->final_foldings[1], &P->rounds[1]));  // :84-89
  P->spans[4] = since(t0);

  // ---- TensorcheckProof::new_time(transcript, ck, [w], [([abc_tensored, z], challenges)])   tensorcheck/mod.rs:190-275
  t0 = Clock::now();
  static const bool trace_env = getenv("GM_SHARD_TRACE") != nullptr;
  PhaseTrace TR(trace_env && r == 0);
  uint64_t batch_challenge[4];
  RC(gm_transcript_challenge_fr(T.h, L("batch_challenge"), 15, batch_challenge));
  uint64_t body;
  {
    uint64_t lc[8];
    Fr::one().to_limbs(lc);
    memcpy(lc + 4, batch_challenge, 32);
    const uint64_t two[2] = {abc, z_blk};
    RC(V.alloc(m, &body));
    RC(gm_fr_lincomb(two, lc, 2, body));
    RC(gm_fr_vec_set_len(body, m));
  }
  // the folding tree: levels 1 .. jmax stay block-sharded (allocated with room for the opening's carry), level jmax + 1 is
  // gathered, the rest is folded replicated                                                    foldings_polynomial :124-133
  std::vector<uint64_t> sharded, small;
  {
    uint64_t cur = body;
    size_t len = m;
    for (size_t j = 1; j + 0 < P->rounds[1]; j++) {  // every challenge but the last (strip_last)
      const size_t nl = (len + 1) / 2;
      uint64_t nxt;
      RC(V.alloc(nl + 3, &nxt));
      RC(gm_fr_vec_set_len(nxt, nl));
      RC(gm_fr_fold(cur, ch2.data() + 4 * (j - 1), nxt));
      len = nl;
      if (j <= jmax) {
        sharded.push_back(nxt);
      } else if (j == jmax + 1 && g > 1) {
        uint64_t full;
        RC(V.alloc(nl * g, &full));
        RC(gm_dist_allgather_vec(nxt, full));
        V.release(nxt);
        nxt = full;
        len = nl * g;
        small.push_back(nxt);
      } else {
        small.push_back(nxt);
      }
      cur = nxt;
    }
  }
  TR.mark("body + folds");
  P->nfold = sharded.size() + small.size();
  GM_CHECK(P->nfold <= cap_rounds, GM_EINVAL, "snark_new_time_sharded: %zu foldings exceed capacity %zu", P->nfold, cap_rounds);
  if (P->nfold) {
    // one pipelined batch: the sharded levels against their slices, the small ones against the replicated prefix
    std::vector<size_t> levels;
    std::vector<uint64_t> vecs;
    for (size_t i = 0; i < sharded.size(); i++) levels.push_back(1 + i);
    for (size_t i = 0; i < small.size(); i++) levels.push_back(PREFIX);
    vecs = sharded;
    vecs.insert(vecs.end(), small.begin(), small.end());
    std::vector<uint64_t> parts(18 * P->nfold);
    RC(key_commit(levels, vecs, parts.data()));
    if (!sharded.empty()) RC(gather_sum(parts.data(), sharded.size(), P->fold_commitments));
    for (size_t i = 0; i < small.size(); i++) RC(gm_g1_sum(parts.data() + 18 * (sharded.size() + i), 1, P->fold_commitments + 18 * (sharded.size() + i)));
  }
  TR.mark("fold commitments");
  for (size_t k = 0; k < P->nfold; k++) RC(gm_transcript_append_g1(T.h, L("commitment"), 10, P->fold_commitments + 18 * k, 1, 0));
  uint64_t pts[12];  // beta^2, beta, -beta
  RC(gm_transcript_challenge_fr(T.h, L("evaluation-chal"), 15, pts + 4));
  Fr xs[3];
  {
    const Fr beta = Fr::from_limbs(pts + 4);
    beta.sqr().to_limbs(pts);
    beta.neg().to_limbs(pts + 8);
    xs[0] = beta.sqr();
    xs[1] = beta;
    xs[2] = beta.neg();
  }
  // block evaluations at all three roots of Z (the transcript takes beta^2 for w only; the carries of the opening need it for
  // every level).  w gets its own copy with room for the carry -- the instance's vector is not ours to extend
  uint64_t w_blk;
  RC(V.alloc(nw + 3, &w_blk));
  RC(gm_fr_vec_set_len(w_blk, nw));
  if (nw) RC(gm_fr_stride(S->w_block, 0, 1, nw, w_blk));
  std::vector<uint64_t> blocks{w_blk};
  blocks.insert(blocks.end(), sharded.begin(), sharded.end());
  std::vector<size_t> blen(blocks.size());
  for (size_t i = 0; i < blocks.size(); i++) blen[i] = m >> i;
  std::vector<uint64_t> allr;
  std::vector<Fr> vals;
  RC(eval_blocks(blocks, pts, 3, blen, allr, vals));
  for (int q = 0; q < 3; q++) vals[q].to_limbs(P->base_evaluations + 4 * q);
  for (size_t i = 1; i < blocks.size(); i++) {
    vals[3 * i + 1].to_limbs(P->fold_evaluations + 8 * (i - 1));
    vals[3 * i + 2].to_limbs(P->fold_evaluations + 8 * (i - 1) + 4);
  }
  if (!small.empty()) RC(gm_fr_eval_le_batch(small.data(), small.size(), pts + 4, 2, P->fold_evaluations + 8 * sharded.size()));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 4, 1));
  RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->base_evaluations + 8, 1));
  for (size_t k = 0; k < 2 * P->nfold; k++) RC(gm_transcript_append_fr(T.h, L("eval"), 4, P->fold_evaluations + 4 * k, 1));
  uint64_t open_chal[4];
  RC(gm_transcript_challenge_fr(T.h, L("open-chal"), 9, open_chal));
  const Fr oc = Fr::from_limbs(open_chal);
  TR.mark("evaluations + transcript");
  // The opening (batch_open_multi_points, src/kzg/time.rs:149-159): commit(F div Z), F = sum_i eta_i p_i over w and every level.
  // Round 4 opened level by level against the per-level key slices -- sum_i |block_i| = 2 n / g pairs per rank.  F itself is ONE
  // polynomial of n coefficients: rank r takes ITS coefficient range [r m, (r + 1) m) of F and commits the quotient of that block
  // against the level-0 key slice it already holds -- n / g pairs.  Level i's coefficients of that range live on other ranks (level i
  // is sharded in blocks of m / 2^i: its range [r m, (r + 1) m) is spread over ranks r 2^i .. (r + 1) 2^i - 1), so the levels are
  // RE-BLOCKED first (gm_dist_reblock_vecs: one grouped send / recv over xGMI; rank 0 receives the most, (log2 g + 1) m elements,
  // ~75 MB at -i 24 on 8 GPUs: ~1 ms of one link against the ~5 ms of accumulation the m pairs cost -- and it can run under the
  // fold commitments).  The quotient of block r needs the carry from the blocks above: the polynomial c of degree < 3 that agrees
  // with S_r(x) = sum_{r' > r} x^((r' - r - 1) m) F_r'(x) at the three roots of Z (one all-gather of three evaluations per rank),
  // and leaves a remainder rem that agrees with G = F_r + x^m c at the roots: q_r = (G - rem) / Z exactly, and F div Z =
  // sum_r x^(r m) q_r.  One linear combination, one division, one MSM of m pairs.
  {
    const size_t nb = blocks.size();  // w and the block-sharded levels 1 .. jmax
    std::vector<uint64_t> pieces, piece_eta;
    auto add_piece = [&](uint64_t v, const Fr& eta) {
      pieces.push_back(v);
      piece_eta.resize(piece_eta.size() + 4);
      eta.to_limbs(piece_eta.data() + piece_eta.size() - 4);
    };
    std::vector<Fr> etas(nb + small.size());
    {
      Fr acc = Fr::one();
      for (auto& e : etas) {
        e = acc;
        acc = acc * oc;
      }
    }
    if (nw) add_piece(w_blk, etas[0]);
    if (g == 1) {
      for (size_t i = 1; i < nb; i++) add_piece(blocks[i], etas[i]);
    } else if (nb > 1) {
      std::vector<uint64_t> outs(nb - 1);
      for (size_t i = 1; i < nb; i++) RC(V.alloc(m, &outs[i - 1]));
      RC(gm_dist_reblock_vecs(blocks.data() + 1, nb - 1, m, outs.data()));
      for (size_t i = 1; i < nb; i++) {
        size_t len = 0;
        RC(vec_len(outs[i - 1], &len));
        if (len) add_piece(outs[i - 1], etas[i]);
      }
    }
    // the gathered levels are replicated: every rank takes ITS range of each (rank 0 all of them when they are shorter than a
    // block; with a long tail -- 2^tail_log > m / g -- the first gathered level spans several blocks)
    for (size_t i = 0; i < small.size(); i++) {
      size_t len = 0;
      RC(vec_len(small[i], &len));
      if (r * m >= len) continue;
      if (r == 0 && len <= m) {
        add_piece(small[i], etas[nb + i]);
        continue;
      }
      const size_t cnt = std::min(m, len - r * m);
      uint64_t part;
      RC(V.alloc(cnt, &part));
      RC(gm_fr_stride(small[i], r * m, 1, cnt, part));
      add_piece(part, etas[nb + i]);
    }
    TR.mark("re-block");
    uint64_t F;
    RC(V.alloc(m + 3, &F));
    {
      uint64_t zero[4] = {0, 0, 0, 0};
      RC(gm_fr_vec_fill(F, zero));
    }
    size_t lf = 0;
    if (!pieces.empty()) {
      RC(gm_fr_lincomb(pieces.data(), piece_eta.data(), pieces.size(), F));
      RC(vec_len(F, &lf));
    }
    GM_CHECK(lf <= m, GM_ESTATE, "snark_new_time_sharded: the block of the opened polynomial has %zu coefficients, more than a block (%zu)", lf, m);
    const bool carry_in = r + 1 < g;
    const size_t len_f = carry_in ? m + 3 : std::max<size_t>(lf, 3);
    RC(gm_fr_vec_set_len(F, len_f));  // (the tail beyond the combination is the zero fill)
    // F_r at the three roots, all ranks
    uint64_t mine_ev[12];
    std::vector<uint64_t> all_ev(12 * g);
    RC(gm_fr_eval_le(F, pts, 3, mine_ev));
    RC(gm_dist_allgather_host(mine_ev, 96, all_ev.data()));
    Fr c[3] = {Fr::zero(), Fr::zero(), Fr::zero()}, gv[3], rem[3];
    std::vector<size_t> pos;
    std::vector<uint64_t> val;
    auto seam = [&](size_t at, const Fr& v) {
      pos.push_back(at);
      val.resize(val.size() + 4);
      v.to_limbs(val.data() + val.size() - 4);
    };
    if (carry_in) {
      Fr ys[3];
      for (int q = 0; q < 3; q++) {
        const Fr step = fr_pow_u64(xs[q], m);
        Fr acc = Fr::zero(), xp = Fr::one();
        for (size_t rr = r + 1; rr < g; rr++) {
          acc = acc + xp * Fr::from_limbs(all_ev.data() + 4 * (rr * 3 + q));
          xp = xp * step;
        }
        ys[q] = acc;
      }
      interp3(xs, ys, c);
      for (int q = 0; q < 3; q++) seam(m + q, c[q]);
    }
    for (int q = 0; q < 3; q++) {
      const Fr cx = c[0] + xs[q] * (c[1] + xs[q] * c[2]);
      gv[q] = Fr::from_limbs(mine_ev + 4 * q) + fr_pow_u64(xs[q], m) * cx;
    }
    interp3(xs, gv, rem);
    for (int q = 0; q < 3; q++) seam(q, rem[q].neg());
    if (r == 0) {
      // the remainder of the whole division is known: the polynomial through the claimed evaluations sum_i eta_i p_i(x_q), which
      // the transcript has already absorbed -- a wrong re-blocking cannot go unnoticed
      for (int q = 0; q < 3; q++) {
        Fr want = etas[0] * vals[q];
        for (size_t i = 1; i < nb; i++) want = want + etas[i] * vals[3 * i + q];
        for (size_t i = 0; i < small.size(); i++) {
          if (q == 0) continue;
          want = want + etas[nb + i] * Fr::from_limbs(P->fold_evaluations + 8 * (sharded.size() + i) + 4 * (q - 1));
        }
        if (q != 0) GM_CHECK(want == gv[q], GM_ESTATE, "snark_new_time_sharded: the opened polynomial does not take the claimed value at root %d", q);
      }
    }
    RC(gm_fr_add_at(F, pos.data(), val.data(), pos.size()));
    uint64_t mine[18];
    memcpy(mine, identity_point(), 144);
    if (len_f > 3) {
      uint64_t quot, remz[12];
      RC(V.alloc(len_f - 1, &quot));  // (the division peels one linear factor at a time: room for the first quotient)
      RC(gm_fr_div_vanishing(F, pts, 3, quot, remz));
      for (int l = 0; l < 12; l++) GM_CHECK(remz[l] == 0, GM_ESTATE, "snark_new_time_sharded: the block of the opening is not divisible by Z (limb %d)", l);
      TR.mark("carries + division");
      size_t lq = 0;
      RC(vec_len(quot, &lq));
      if (lq) {
        lq = std::min(lq, S->key_counts[0]);
        const size_t off0 = S->key_offsets[0];
        RC(gm_g1_msm_v_batch_at(S->key, &off0, 0, &quot, &lq, 1, 1, mine));
      }
    }
    TR.mark("opening MSM");
    RC(gather_sum(mine, 1, P->evaluation_proof));
  }
  P->spans[5] = since(t0);
  P->spans[6] = since(t_all);
  return GM_OK;
}

// snark::Proof::new_elastic (src/snark/elastic_prover.rs:174-266, examples/snark.rs:54-66: BASELINE configs[3], `snark -i 28` on 8 GPUs) with
// every vector block-sharded.  On the device the elastic prover takes the RESIDENT schedule whenever min_device_chunk > 1 (gm_snark_new_elastic:
// time provers on the little-endian vectors from the first round -- the same field elements as the space provers over the reversed streams,
// sumcheck/tests.rs:42-87 -- and every stream MSM as ONE call: max_msm_buffer is advisory because everything is in HBM): that schedule over
// blocks IS the block-sharded time prover, on the same blocks, so this entry validates the elastic arguments and runs it.  The LITERAL
// schedule (min_device_chunk = 1: 2^20-pair flushes, space provers to SPACE_TIME_THRESHOLD) re-derives every message from whole streams
// and stays single-GPU (gm_snark_new_elastic).  The key of examples/snark.rs:59-63 (DummyStreamer: copies of the generator) in slices is
// gm_snark_shard_key_new with tau = 1.  Same proof bytes as gm_snark_new_elastic and gm_snark_new_time on the same instance and key
// (src/snark/tests.rs:56), on every rank.
int gm_snark_new_elastic_sharded(const gm_snark_shard* S, size_t max_msm_buffer, size_t min_device_chunk, int g1_encoding, size_t cap_rounds, gm_snark_proof* P) {
  GM_CTX();
  GM_CHECK(max_msm_buffer >= 1, GM_EINVAL, "snark_new_elastic_sharded: max_msm_buffer = 0");
  GM_CHECK(min_device_chunk > 1, GM_EINVAL,
           "snark_new_elastic_sharded: min_device_chunk = 1 selects the LITERAL elastic schedule (space provers over whole streams), which is single-GPU: "
           "gm_snark_new_elastic; the block-sharded entry runs the resident schedule");
  const int rc = gm_snark_new_time_sharded(S, g1_encoding, cap_rounds, P);
  if (!rc) P->spans[0] = 0.0;  // as gm_snark_new_elastic: the matrix products belong to the construction of the streams
  return rc;
}

}  // extern "C"

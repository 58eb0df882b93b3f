// What an embedder without Python gets from the C ABI alone: `world` PROCESSES (fork), one gm context each, the library's own
// shared-memory all-gather, and snark::Proof::new_time (src/snark/time_prover.rs:19-117) on dummy_r1cs(e, n)
// (src/circuit.rs:349-365) two ways --
//   block  gm_snark_new_time_sharded: every vector and the key block-sharded (general matrices with --global: row blocks, global
//          columns, z whole);
//   cyclic gm_snark_new_time handed an element-cyclic share of the key (gm_g1_srs_register_cyclic): MSMs sharded.
//   psnark gm_psnark_new_time_sharded (src/psnark/time_prover.rs:69-384): this rank's blocks of dummy_r1cs in closed form (the joint support
//          is the diagonal), each family at its level (gm_psnark_shard_level), gm_psnark_shard_key_new, gm_psnark_index_sharded; ANY world
//          size.  argv[9] = a file with the serialised G2 powers the transcript absorbs (the caller's: no G2 arithmetic in this ABI).
// Every rank prints a 64-bit digest of the whole proof; the caller (tests/test_gpu_cpp_host.py) compares it with the digest of
// the single-process run.  usage: test_sharded_ranks <world> <logn> <block|cyclic|psnark> <local|global> e einv g tau [g2 file]
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gemini_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_) {                                                                     \
      fprintf(stderr, "rank %d: %s -> %d: %s\n", g_rank, #x, rc_, gm_last_error()); \
      _exit(2);                                                                    \
    }                                                                              \
  } while (0)

static int g_rank = 0;

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

// ---- psnark, block-sharded ----------------------------------------------------------------------------------------------------
static int run_psnark_rank(int rank, int world, int logn, const char* shm, const uint64_t e_mont[4], const uint64_t einv_mont[4], const uint64_t g_aff[12],
                           const uint64_t tau[4], const std::vector<uint8_t>& g2) {
  g_rank = rank;
  CK(gm_init(0));
  CK(gm_dist_init_shm(rank, world, shm, 0));
  const size_t n = (size_t)1 << logn, tail_log = 4;
  const size_t ext_row = 2 * n, ext_col = 2 * n, longest = 2 * n + 2;  // (n a power of two: the tensor of the first sumcheck has n entries)
  const size_t block = gm_psnark_shard_block(longest, world);
  auto cut = [&](size_t family_len, size_t whole, size_t* lo, size_t* hi) {
    const size_t b = block >> gm_psnark_shard_level(family_len, block, tail_log, world);
    *lo = std::min((size_t)rank * b, whole);
    *hi = std::min(((size_t)rank + 1) * b, whole);
  };
  uint64_t one[4];
  {  // 1 in Montgomery form = e * (1 / e), by the library's own arithmetic
    uint64_t a = 0, b = 0, c = 0;
    CK(gm_fr_vec_alloc(1, &a));
    CK(gm_fr_vec_alloc(1, &b));
    CK(gm_fr_vec_alloc(1, &c));
    CK(gm_fr_vec_fill(a, e_mont));
    CK(gm_fr_vec_fill(b, einv_mont));
    CK(gm_fr_hadamard(a, b, c));
    CK(gm_fr_vec_download(c, 0, one, 1));
  }
  gm_psnark_shard S;
  memset(&S, 0, sizeof S);
  size_t lo = 0, hi = 0;
  cut(n, n, &lo, &hi);  // the row block of diag(1 / e), global columns
  if (hi > lo) {
    const size_t m = hi - lo;
    std::vector<uint64_t> rowptr(m + 1), vals(4 * m);
    std::vector<uint32_t> cols(m);
    for (size_t i = 0; i <= m; i++) rowptr[i] = i;
    for (size_t i = 0; i < m; i++) {
      cols[i] = (uint32_t)(lo + i);
      memcpy(&vals[4 * i], einv_mont, 32);
    }
    CK(gm_spm_register(rowptr.data(), cols.data(), vals.data(), m, n, m, &S.a));
    S.b = S.c = S.a;
  }
  CK(gm_fr_vec_alloc(n, &S.z));
  CK(gm_fr_vec_fill(S.z, e_mont));
  cut(n - 1, n - 1, &lo, &hi);
  S.w_len = n - 1;
  if (hi > lo) {
    CK(gm_fr_vec_alloc(hi - lo, &S.w_block));
    CK(gm_fr_vec_fill(S.w_block, e_mont));
  }
  cut(n + 1, n, &lo, &hi);  // the joint support = the diagonal, and everything indexed by it
  if (hi > lo) {
    const size_t m = hi - lo;
    std::vector<uint32_t> idx(m);
    for (size_t i = 0; i < m; i++) idx[i] = (uint32_t)(lo + i);
    CK(gm_idx_register(idx.data(), m, &S.row_index));
    CK(gm_idx_register(idx.data(), m, &S.col_index));
    uint64_t zeros = 0, zero[4] = {0, 0, 0, 0};
    CK(gm_fr_vec_alloc(m, &zeros));
    CK(gm_fr_vec_fill(zeros, zero));
    CK(gm_fr_vec_alloc(m, &S.row));
    CK(gm_fr_alg_hash(zeros, S.row_index, one, S.row));  // F::from(index) = 0 + 1 * index
    CK(gm_fr_vec_alloc(m, &S.col));
    CK(gm_fr_alg_hash(zeros, S.col_index, one, S.col));
    for (uint64_t* v : {&S.val_a, &S.val_b, &S.val_c}) {
      CK(gm_fr_vec_alloc(m, v));
      CK(gm_fr_vec_fill(*v, einv_mont));
    }
  }
  for (int k = 0; k < 2; k++) {  // extend_frequency(compute_frequency(n, 0 .. n - 1)): every index twice
    const size_t whole = k ? ext_col : ext_row;
    cut(whole + 2, whole, &lo, &hi);
    if (hi > lo) {
      std::vector<uint32_t> idx(hi - lo);
      for (size_t j = lo; j < hi; j++) idx[j - lo] = (uint32_t)(j / 2);
      CK(gm_idx_register(idx.data(), hi - lo, k ? &S.ext_fre_col : &S.ext_fre_row));
    }
  }
  S.ext_fre_row_len = ext_row;
  S.ext_fre_col_len = ext_col;
  S.num_constraints = S.num_variables = S.nnz = n;
  S.block = block;
  S.tail_log = tail_log;
  size_t offsets[64], counts[64], nseg = 0;
  S.key_len = 2 * n + 1;  // CommitterKey::new(num_constraints + num_variables, ..): max_degree + 1 powers (examples/psnark.rs:76)
  CK(gm_psnark_shard_key_new(g_aff, tau, S.key_len, block, tail_log, &S.key, offsets, counts, &nseg));
  S.key_offsets = offsets;
  S.key_counts = counts;
  S.key_segments = nseg;
  S.ck_g2_bytes = g2.data();
  S.ck_g2_len = g2.size();
  uint64_t index[5 * 18];
  CK(gm_psnark_index_sharded(&S, index));
  S.index_commitments = index;
  const size_t cap = (size_t)logn + 8, cap_folds = 4 * cap;
  std::vector<uint64_t> m0(8 * cap), m1(8 * cap), m2(8 * cap), fc(18 * cap_folds), fe(8 * cap_folds);
  gm_psnark_proof P;
  memset(&P, 0, sizeof P);
  P.messages[0] = m0.data();
  P.messages[1] = m1.data();
  P.messages[2] = m2.data();
  P.cap_folds = cap_folds;
  P.fold_commitments = fc.data();
  P.fold_evaluations = fe.data();
  CK(gm_psnark_new_time_sharded(&S, 0, cap, &P));
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv(h, index, sizeof index);
  h = fnv(h, P.witness_commitment, sizeof P.witness_commitment);
  h = fnv(h, P.zc_alpha, sizeof P.zc_alpha);
  for (int k = 0; k < 3; k++) {
    h = fnv(h, &P.rounds[k], sizeof(size_t));
    h = fnv(h, P.messages[k], 64 * P.rounds[k]);
  }
  h = fnv(h, P.final_foldings, sizeof P.final_foldings);
  h = fnv(h, P.third_final_foldings, sizeof P.third_final_foldings);
  h = fnv(h, P.r_star_commitments, sizeof P.r_star_commitments);
  h = fnv(h, P.z_star_commitment, sizeof P.z_star_commitment);
  h = fnv(h, P.sorted_commitments, sizeof P.sorted_commitments);
  h = fnv(h, P.products, sizeof P.products);
  h = fnv(h, P.acc_v_commitments, sizeof P.acc_v_commitments);
  h = fnv(h, P.claimed_sumchecks, sizeof P.claimed_sumchecks);
  h = fnv(h, P.ralpha_star_acc_mu_evals, sizeof P.ralpha_star_acc_mu_evals);
  h = fnv(h, P.ralpha_star_acc_mu_proof, sizeof P.ralpha_star_acc_mu_proof);
  h = fnv(h, P.rstars_vals, sizeof P.rstars_vals);
  h = fnv(h, &P.nfold, sizeof(size_t));
  h = fnv(h, P.fold_commitments, 144 * P.nfold);
  h = fnv(h, P.fold_evaluations, 64 * P.nfold);
  h = fnv(h, P.evaluation_proof, sizeof P.evaluation_proof);
  h = fnv(h, P.base_evaluations, sizeof P.base_evaluations);
  uint64_t all[64];
  CK(gm_dist_allgather_host(&h, 8, all));
  for (int r = 0; r < world; r++)
    if (all[r] != h) {
      fprintf(stderr, "rank %d: rank %d holds another proof\n", rank, r);
      return 3;
    }
  uint64_t calls = 0, bytes = 0;
  double secs = 0;
  CK(gm_dist_stats(&calls, &bytes, &secs, 0));
  if (rank == 0) printf("digest %016llx rounds %zu nfold %zu collectives %llu\n", (unsigned long long)h, P.rounds[2], P.nfold, (unsigned long long)calls);
  CK(gm_dist_finalize());
  return 0;
}

// Montgomery images from the library itself: x -> x * R via gm_fr_* would need more plumbing than this test wants, so the two field
// constants it needs (e and 1 / e) come in as arguments in Montgomery form from the caller
static int run_rank(int rank, int world, int logn, bool block, bool global_cols, const char* shm, const uint64_t e_mont[4], const uint64_t einv_mont[4],
                    const uint64_t g_aff[12], const uint64_t tau[4]) {
  g_rank = rank;
  CK(gm_init(0));  // every rank on the one GPU of the test box; one GPU per rank on a node
  CK(gm_dist_init_shm(rank, world, shm, 0));
  CK(gm_dist_selftest());
  const size_t n = (size_t)1 << logn, m = n / (size_t)world;
  const size_t cap = (size_t)logn + 3;
  std::vector<uint64_t> msg0(8 * cap), msg1(8 * cap), fc(18 * cap), fe(8 * cap);
  gm_snark_proof P;
  memset(&P, 0, sizeof P);
  P.messages[0] = msg0.data();
  P.messages[1] = msg1.data();
  P.fold_commitments = fc.data();
  P.fold_evaluations = fe.data();
  if (block) {
    const size_t ncols = global_cols ? n : m;
    std::vector<uint64_t> rowptr(m + 1), vals(4 * m);
    std::vector<uint32_t> cols(m);
    for (size_t i = 0; i <= m; i++) rowptr[i] = i;
    for (size_t i = 0; i < m; i++) {
      cols[i] = (uint32_t)(global_cols ? (size_t)rank * m + i : i);
      memcpy(&vals[4 * i], einv_mont, 32);
    }
    uint64_t d = 0, z = 0, w = 0;
    CK(gm_spm_register(rowptr.data(), cols.data(), vals.data(), m, ncols, m, &d));
    CK(gm_fr_vec_alloc(ncols, &z));
    CK(gm_fr_vec_fill(z, e_mont));
    CK(gm_fr_vec_alloc(rank == world - 1 ? m - 1 : m, &w));
    CK(gm_fr_vec_fill(w, e_mont));
    size_t offsets[64], counts[64], nseg = 0;
    uint64_t key = 0;
    CK(gm_snark_shard_key_new(g_aff, tau, n, 4, &key, offsets, counts, &nseg));
    gm_snark_shard S;
    for (int k = 0; k < 6; k++) S.matrices[k] = d;
    S.z = z;
    S.w_block = w;
    S.key = key;
    S.key_offsets = offsets;
    S.key_counts = counts;
    S.key_segments = nseg;
    S.n = n;
    S.tail_log = 4;
    CK(gm_snark_new_time_sharded(&S, 0, cap, &P));
  } else {
    std::vector<uint64_t> rowptr(n + 1), vals(4 * n);
    std::vector<uint32_t> cols(n);
    for (size_t i = 0; i <= n; i++) rowptr[i] = i;
    for (size_t i = 0; i < n; i++) {
      cols[i] = (uint32_t)i;
      memcpy(&vals[4 * i], einv_mont, 32);
    }
    uint64_t d = 0, z = 0, w = 0, ck = 0;
    CK(gm_spm_register(rowptr.data(), cols.data(), vals.data(), n, n, n, &d));
    CK(gm_fr_vec_alloc(n, &z));
    CK(gm_fr_vec_fill(z, e_mont));
    CK(gm_fr_vec_alloc(n - 1, &w));
    CK(gm_fr_vec_fill(w, e_mont));
    CK(gm_g1_srs_register_cyclic(g_aff, tau, 2 * n + 1, rank, world, &ck));
    const uint64_t mats[6] = {d, d, d, d, d, d};
    CK(gm_snark_new_time(mats, z, w, ck, 0, cap, &P));
  }
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv(h, P.witness_commitment, sizeof P.witness_commitment);
  h = fnv(h, P.zc_alpha, sizeof P.zc_alpha);
  for (int k = 0; k < 2; k++) {
    h = fnv(h, &P.rounds[k], sizeof(size_t));
    h = fnv(h, P.messages[k], 64 * P.rounds[k]);
    h = fnv(h, P.final_foldings[k], 64);
  }
  h = fnv(h, &P.nfold, sizeof(size_t));
  h = fnv(h, P.fold_commitments, 144 * P.nfold);
  h = fnv(h, P.fold_evaluations, 64 * P.nfold);
  h = fnv(h, P.evaluation_proof, sizeof P.evaluation_proof);
  h = fnv(h, P.base_evaluations, sizeof P.base_evaluations);
  uint64_t all[64];
  CK(gm_dist_allgather_host(&h, 8, all));
  for (int r = 0; r < world; r++)
    if (all[r] != h) {
      fprintf(stderr, "rank %d: rank %d holds another proof\n", rank, r);
      return 3;
    }
  uint64_t calls = 0, bytes = 0;
  double secs = 0;
  CK(gm_dist_stats(&calls, &bytes, &secs, 0));
  if (rank == 0) printf("digest %016llx rounds %zu nfold %zu collectives %llu\n", (unsigned long long)h, P.rounds[0], P.nfold, (unsigned long long)calls);
  CK(gm_dist_finalize());
  return 0;
}

static void parse_hex(const char* s, uint64_t* out, int limbs) {
  for (int i = 0; i < limbs; i++) {
    char buf[17];
    memcpy(buf, s + 16 * i, 16);
    buf[16] = 0;
    out[i] = strtoull(buf, nullptr, 16);
  }
}

int main(int argc, char** argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s world logn block|cyclic local|global e_mont(hex limbs) einv_mont g_affine(12 limbs) tau(4 limbs, canonical)\n", argv[0]);
    return 1;
  }
  const int world = atoi(argv[1]), logn = atoi(argv[2]);
  const bool block = !strcmp(argv[3], "block"), global_cols = !strcmp(argv[4], "global");
  uint64_t e[4], einv[4], g[12], tau[4];
  parse_hex(argv[5], e, 4);
  parse_hex(argv[6], einv, 4);
  parse_hex(argv[7], g, 12);
  parse_hex(argv[8], tau, 4);
  const std::string shm = "/gm_cpp_ranks_" + std::to_string((long)getpid());
  const bool psnark = !strcmp(argv[3], "psnark");
  std::vector<uint8_t> g2;
  if (psnark) {
    if (argc < 10) return 1;
    FILE* f = fopen(argv[9], "rb");
    if (!f) return 1;
    uint8_t buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) g2.insert(g2.end(), buf, buf + got);
    fclose(f);
  }
  auto rank_main = [&](int r) { return psnark ? run_psnark_rank(r, world, logn, shm.c_str(), e, einv, g, tau, g2) : run_rank(r, world, logn, block, global_cols, shm.c_str(), e, einv, g, tau); };
  std::vector<pid_t> kids;
  for (int r = 1; r < world; r++) {
    pid_t p = fork();  // before any GPU context exists in this process
    if (p == 0) _exit(rank_main(r));
    kids.push_back(p);
  }
  int rc = rank_main(0);
  for (pid_t p : kids) {
    int st = 0;
    waitpid(p, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = rc ? rc : 4;
  }
  return rc;
}

// Drives include/gemini_hip.hpp (the C++ host mirror of the reference's Rust API) on inputs written by
// tests/test_gpu_cpp_host.py and prints results as hex for the Python side to compare with the oracle.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "gemini_hip.hpp"

template <class T>
static std::vector<T> read_vec(std::ifstream& in) {
  uint64_t n;
  in.read((char*)&n, 8);
  std::vector<T> v(n);
  in.read((char*)v.data(), n * sizeof(T));
  return v;
}
template <size_t N>
static void print(const char* tag, const std::array<uint64_t, N>& a) {
  printf("%s", tag);
  for (auto x : a) printf(" %016llx", (unsigned long long)x);
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1], std::ios::binary);
  auto bases = read_vec<gm::G1Affine>(in);
  auto bigints = read_vec<gm::BigInt>(in);
  auto scalars = read_vec<gm::Fr>(in);
  auto f = read_vec<gm::Fr>(in);
  auto g = read_vec<gm::Fr>(in);
  auto tw = read_vec<gm::Fr>(in);
  try {
    gm::init(0);
    print("msm_bigint", gm::VariableBaseMSM::msm_bigint(bases, bigints));
    print("msm_unchecked", gm::VariableBaseMSM::msm_unchecked(bases, scalars));
    auto shorter = scalars;
    shorter.resize(scalars.size() - 5);
    auto res = gm::VariableBaseMSM::msm(bases, shorter);
    printf("msm_err %d %zu\n", res.first.has_value() ? 0 : 1, res.second);
    gm::ChunkedPippenger cp(7);
    for (size_t i = 0; i < bases.size(); i++) cp.add(bases[i], bigints[i]);
    print("chunked", cp.finalize());
    gm::ChunkedPippenger cp2 = gm::ChunkedPippenger::with_size(64);
    cp2.add(bases[0], bigints[0]);
    cp2.add_pairs(bases.data() + 1, bigints.data() + 1, bases.size() - 1);
    print("chunked_blocks", cp2.finalize());
    auto tail = std::vector<gm::Fr>(scalars.begin(), scalars.begin() + 100);
    print("msm_chunks", gm::msm_chunks(bases, tail));
    gm::HashMapPippenger hp(16);
    for (size_t i = 0; i < bases.size(); i++) hp.add(bases[i % 20], scalars[i]);
    print("hashmap", hp.finalize());
    gm::CommitterKey ck(bases);
    print("commit", ck.commit(scalars));
    {  // CommitterKey::batch_commit (src/kzg/time.rs:98-107): one pipelined call, mixed lengths incl. an empty polynomial
      std::vector<std::vector<gm::Fr>> polys = {scalars, tail, std::vector<gm::Fr>(), std::vector<gm::Fr>(scalars.begin(), scalars.begin() + 1)};
      auto batch = ck.batch_commit(polys);
      bool same = batch.size() == polys.size();
      for (size_t j = 0; same && j < polys.size(); j++) same = batch[j] == ck.commit(polys[j]);
      printf("batch_commit_equals_commits %d\n", same ? 1 : 0);
    }
    gm::Transcript t;
    auto sc = gm::Sumcheck::new_time(t, f, g, tw[0]);
    for (size_t k = 0; k < sc.messages.size(); k++) {
      print("msg_a", sc.messages[k].a);
      print("msg_b", sc.messages[k].b);
      print("chal", sc.challenges[k]);
    }
    print("ff0", sc.final_foldings[0][0]);
    print("ff1", sc.final_foldings[0][1]);
    print("after", t.get_challenge("after"));
    // snark::Proof::new_time on dummy_r1cs(e, n): A = B = C = diag(1 / e), z = [e; n], w = [e; n - 1]
    if (in.peek() != EOF) {
      auto ev = read_vec<gm::Fr>(in);  // e, 1 / e
      auto srs = read_vec<gm::G1Affine>(in);
      const size_t n = (srs.size() - 1) / 2;
      gm::Matrix diag(n);
      for (size_t i = 0; i < n; i++) diag[i].push_back({ev[1], i});
      std::vector<gm::Fr> z(n, ev[0]), w(n - 1, ev[0]);
      gm::R1cs r1cs(diag, diag, diag, z, w);
      gm::CommitterKey key(srs);
      // the footprint contract from C++: promised before, used after (gm_snark_footprint / gm_mem_stats)
      const gm::Footprint fp = gm::snark_footprint(key.handle(), n);
      const gm::MemStats m0 = gm::mem_stats();
      gm::mem_reset_peak();
      auto proof = gm::SnarkProof::new_time(r1cs, key);
      const gm::MemStats m1 = gm::mem_stats();
      if (fp.needed != fp.vectors + fp.workspaces_to_grow || m1.in_use_peak - m0.in_use > fp.needed || fp.available < fp.needed) {
        fprintf(stderr, "footprint: promised %llu, used %llu, available %llu\n", (unsigned long long)fp.needed, (unsigned long long)(m1.in_use_peak - m0.in_use),
                (unsigned long long)fp.available);
        return 3;
      }
      print("snark_witness", proof.witness_commitment);
      print("snark_zc_alpha", proof.zc_alpha);
      for (auto& m : proof.first_sumcheck_msgs) {
        print("snark_m1a", m.a);
        print("snark_m1b", m.b);
      }
      for (auto& m : proof.second_sumcheck_msgs) {
        print("snark_m2a", m.a);
        print("snark_m2b", m.b);
      }
      print("snark_ff1", proof.first_final_foldings[0]);
      print("snark_ff1", proof.first_final_foldings[1]);
      print("snark_ff2", proof.second_final_foldings[0]);
      print("snark_ff2", proof.second_final_foldings[1]);
      for (auto& c : proof.tensorcheck_proof.folded_polynomials_commitments) print("snark_fc", c);
      for (auto& e2 : proof.tensorcheck_proof.folded_polynomials_evaluations) {
        print("snark_fe", e2[0]);
        print("snark_fe", e2[1]);
      }
      print("snark_open", proof.tensorcheck_proof.evaluation_proof);
      for (auto& e : proof.tensorcheck_proof.base_polynomials_evaluations[0]) print("snark_be", e);
      // Proof::new_elastic on the same instance and key: assert_eq!(time_proof, space_proof) (src/snark/tests.rs:56);
      // max_msm_buffer cut literally (min_device_chunk = 1) and merged
      for (size_t floor : {(size_t)1, (size_t)1 << 26}) {
        auto el = gm::SnarkProof::new_elastic(r1cs, key, 4, floor);
        const bool same = el.witness_commitment == proof.witness_commitment && el.zc_alpha == proof.zc_alpha &&
                          el.first_final_foldings == proof.first_final_foldings && el.second_final_foldings == proof.second_final_foldings &&
                          el.tensorcheck_proof.folded_polynomials_commitments == proof.tensorcheck_proof.folded_polynomials_commitments &&
                          el.tensorcheck_proof.folded_polynomials_evaluations == proof.tensorcheck_proof.folded_polynomials_evaluations &&
                          el.tensorcheck_proof.evaluation_proof == proof.tensorcheck_proof.evaluation_proof &&
                          el.tensorcheck_proof.base_polynomials_evaluations == proof.tensorcheck_proof.base_polynomials_evaluations &&
                          el.first_sumcheck_msgs.size() == proof.first_sumcheck_msgs.size() && el.second_sumcheck_msgs.size() == proof.second_sumcheck_msgs.size();
        bool msgs_same = same;
        for (size_t i = 0; msgs_same && i < el.first_sumcheck_msgs.size(); i++)
          msgs_same = el.first_sumcheck_msgs[i].a == proof.first_sumcheck_msgs[i].a && el.first_sumcheck_msgs[i].b == proof.first_sumcheck_msgs[i].b;
        for (size_t i = 0; msgs_same && i < el.second_sumcheck_msgs.size(); i++)
          msgs_same = el.second_sumcheck_msgs[i].a == proof.second_sumcheck_msgs[i].a && el.second_sumcheck_msgs[i].b == proof.second_sumcheck_msgs[i].b;
        printf("snark_elastic_equals_time %d\n", (int)msgs_same);
      }
    }
    // psnark::Proof::{index, new_time} on a general sparse instance: matrices in the Rust layout (rows of (value, column)), the key
    // and its serialised G2 powers from the Python side
    if (in.peek() != EOF) {
      auto dims = read_vec<uint64_t>(in);  // n, then per matrix: nnz entries follow as (row, col) pairs + values
      const size_t n = dims[0];
      gm::Matrix M[3];
      for (int k = 0; k < 3; k++) {
        auto rc = read_vec<uint64_t>(in);  // row, col interleaved
        auto vals = read_vec<gm::Fr>(in);
        M[k].assign(n, {});
        for (size_t e = 0; e < vals.size(); e++) M[k][rc[2 * e]].push_back({vals[e], (size_t)rc[2 * e + 1]});
      }
      auto z = read_vec<gm::Fr>(in);
      auto w = read_vec<gm::Fr>(in);
      auto srs = read_vec<gm::G1Affine>(in);
      auto g2 = read_vec<uint8_t>(in);
      gm::R1cs r1cs(M[0], M[1], M[2], z, w);
      gm::CommitterKey key(srs);
      gm::PsnarkInstance inst(r1cs);
      printf("psnark_nnz %zu\n", inst.num_non_zero());
      auto index = inst.index(key);
      for (auto& c : index) print("psnark_index", c);
      auto proof = gm::PsnarkProof::new_time(key, inst, index, g2);
      print("psnark_witness", proof.witness_commitment);
      print("psnark_zc_alpha", proof.zc_alpha);
      for (auto& c : proof.r_star_commitments) print("psnark_rstar", c);
      print("psnark_zstar", proof.z_star_commitment);
      for (auto& c : proof.sorted_commitments) print("psnark_sorted", c);
      for (auto& v : proof.products) print("psnark_product", v);
      for (auto& c : proof.acc_v_commitments) print("psnark_accv", c);
      for (auto& v : proof.rstars_vals) print("psnark_rstars_val", v);
      print("psnark_mu_proof", proof.ralpha_star_acc_mu_proof);
      print("psnark_open", proof.tensorcheck_proof.evaluation_proof);
      printf("psnark_rounds %zu %zu %zu %zu\n", proof.first_sumcheck_msgs.size(), proof.second_sumcheck_msgs.size(), proof.third_sumcheck_msgs.size(),
             proof.tensorcheck_proof.folded_polynomials_commitments.size());
    }
    // error behaviour: hadamard-style length mismatch surfaces as gm::Error, not a crash
    try {
      gm::check(gm_g1_bases_free(0xdeadbeef));
      printf("error_path none\n");
    } catch (const gm::Error& e) {
      printf("error_path %d\n", e.code);
    }
  } catch (const gm::Error& e) {
    printf("FAILED %d %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}

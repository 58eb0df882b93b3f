"""GPU: the C++ host mirror (include/gemini_hip.hpp) compiled with g++ against libgemini_hip.so and
checked against the oracle -- the same assertions as the Python-mirror tests, through compiled code."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.util import jac_to_affine_ints, rand_bases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wvec(fh, arr):
    arr = np.ascontiguousarray(arr)
    fh.write(struct.pack("<Q", arr.shape[0]))
    fh.write(arr.tobytes())


def test_cpp_host_layer(oracle, pyref, tmp_path):
    exe = str(tmp_path / "test_host_api")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"),
                           "-L", os.path.join(ROOT, "gemini_amd"), "-lgemini_hip", "-Wl,-rpath," + os.path.join(ROOT, "gemini_amd"), "-o", exe])
    n = 200
    bases = rand_bases(oracle, 301, n)
    bases[9] = 0
    rust = np.zeros((n, 13), dtype=np.uint64)
    rust[:, :12] = bases
    rust[9, 12] = 1
    sc = oracle.random_fr(302, n)
    mont = oracle.fr_to_mont(sc)
    nf = 300
    f = oracle.fr_to_mont(oracle.random_fr(303, nf))
    g = oracle.fr_to_mont(oracle.random_fr(304, nf))
    tw = oracle.fr_to_mont(oracle.random_fr(305, 1))
    inp = str(tmp_path / "in.bin")
    # snark::Proof::new_time through the C++ mirror: dummy_r1cs(e, 16) and an SRS of 2 n + 1 powers in the Rust layout
    from oracle import snark_ref as sr

    ns = 16
    e_i = oracle.limbs_to_ints(oracle.random_fr(306, 1))[0]
    tau_i = oracle.limbs_to_ints(oracle.random_fr(307, 1))[0]
    ev = oracle.fr_to_mont(oracle.ints_to_limbs([e_i, pow(e_i, -1, pyref.R_MOD)], 4))
    srs = sr.srs(tau_i, 2 * ns + 1)
    srs_rust = np.zeros((2 * ns + 1, 13), dtype=np.uint64)
    srs_rust[:, :12] = srs
    # psnark::Proof::{index, new_time} through the C++ mirror on a general sparse instance (gm::PsnarkInstance: the joint matrices
    # are built inside the library); expected: the Python mirror's own proof of the same instance on the same key
    import gemini_amd as gm
    from gemini_amd.circuit import R1cs, SparseMatrix
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof as PsnarkProof
    from oracle import psnark_ref as pr
    from tests.util import random_r1cs_instance

    gm.capi.init()
    np_ = 32
    inst, ptau = random_r1cs_instance(pyref, sr, np_, 7400)
    M = lambda v: gm.fr.fr_from_int(v)  # noqa: E731
    dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]  # noqa: E731
    mont_of = lambda ints: oracle.fr_to_mont(oracle.ints_to_limbs(ints, 4))  # noqa: E731
    mats = [SparseMatrix.from_rows(dev(inst[k]), np_) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), np_, transpose=True) for k in "abc"]
    pr1cs = R1cs(*mats, gm.FrVec.from_host(mont_of(inst["z"])), gm.FrVec.from_host(mont_of(inst["w"])), gm.FrVec.from_host(mont_of(inst["x"])))
    jm = pr.sum_matrices(inst["a"], inst["b"], inst["c"], np_)
    nnz = len(pr.joint_matrices(jm, inst["a"], inst["b"], inst["c"])[0])
    pck = CommitterKey.new(nnz + 2 * np_, 3, oracle.ints_to_limbs([ptau], 4)[0])
    pindex = PsnarkProof.index(pck, pr1cs)
    pexp = PsnarkProof.new_time(pck, pr1cs, pindex)
    psrs = pck.powers_of_g.download()
    psrs_rust = np.zeros((len(psrs), 13), dtype=np.uint64)
    psrs_rust[:, :12] = psrs
    g2_bytes = np.frombuffer(pck.powers_of_g2_bytes(), dtype=np.uint8)
    with open(inp, "wb") as fh:
        for a in (rust, sc, mont, f, g, tw, ev, srs_rust):
            _wvec(fh, a)
        _wvec(fh, np.array([np_], dtype=np.uint64))
        for k in "abc":
            rc = np.array([[r, col] for r, row in enumerate(inst[k]) for _, col in row], dtype=np.uint64).reshape(-1)
            _wvec(fh, rc)
            _wvec(fh, mont_of([v for row in inst[k] for v, _ in row]))
        _wvec(fh, mont_of(inst["z"]))
        _wvec(fh, mont_of(inst["w"]))
        _wvec(fh, psrs_rust)
        _wvec(fh, g2_bytes)
    out = subprocess.check_output([exe, inp], text=True)
    vals = {}
    for line in out.splitlines():
        parts = line.split()
        vals.setdefault(parts[0], []).append(parts[1:])
    assert "FAILED" not in vals, out
    J = lambda key, k=0: np.array([int(x, 16) for x in vals[key][k]], dtype=np.uint64)
    exp = oracle.msm_pippenger(bases, sc)
    for key in ("msm_bigint", "msm_unchecked", "chunked", "chunked_blocks", "commit"):
        assert jac_to_affine_ints(oracle, J(key)) == jac_to_affine_ints(oracle, exp), key
    # msm_chunks aligns the streams: the last 100 bases with the 100 scalars (src/kzg/space.rs:36-40)
    assert jac_to_affine_ints(oracle, J("msm_chunks")) == jac_to_affine_ints(oracle, oracle.msm_pippenger(bases[n - 100:], sc[:100]))
    assert vals["msm_err"][0] == ["1", str(n - 5)]
    assert vals["batch_commit_equals_commits"][0] == ["1"]
    dup = bases[np.arange(n) % 20]
    assert jac_to_affine_ints(oracle, J("hashmap")) == jac_to_affine_ints(oracle, oracle.hashmap_pippenger(dup, mont, 16))
    # sumcheck + transcript vs the restatements
    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    msgs, chs, ff = pyref.sumcheck_prove(tr, pyref.TimeProver(I(f), I(g), I(tw)[0]))
    assert len(vals["msg_a"]) == len(msgs) == 9
    for k in range(len(msgs)):
        assert (I(J("msg_a", k))[0], I(J("msg_b", k))[0]) == msgs[k]
        assert I(J("chal", k))[0] == chs[k]
    assert (I(J("ff0"))[0], I(J("ff1"))[0]) == ff
    assert I(J("after"))[0] == tr.get_challenge(b"after")
    assert vals["error_path"][0] == ["-3"]  # GM_EHANDLE surfaced as an exception, no crash
    # the proof of gm::SnarkProof::new_time equals the restatement's, element by element
    exp = sr.snark_new_time(sr.dummy_r1cs(e_i, ns), srs)
    A = lambda key, k=0: jac_to_affine_ints(oracle, J(key, k))
    F = lambda key, k=0: I(J(key, k))[0]
    assert A("snark_witness") == exp["witness_commitment"] and F("snark_zc_alpha") == exp["zc_alpha"]
    for tag, name in (("1", "first_sumcheck_msgs"), ("2", "second_sumcheck_msgs")):
        msgs, ff = exp[name]
        assert [(F(f"snark_m{tag}a", k), F(f"snark_m{tag}b", k)) for k in range(len(vals[f"snark_m{tag}a"]))] == msgs
        assert (F(f"snark_ff{tag}", 0), F(f"snark_ff{tag}", 1)) == ff
    tc = exp["tensorcheck_proof"]
    assert [A("snark_fc", k) for k in range(len(vals["snark_fc"]))] == tc["folded_polynomials_commitments"]
    assert [[F("snark_fe", 2 * k), F("snark_fe", 2 * k + 1)] for k in range(len(vals["snark_fe"]) // 2)] == tc["folded_polynomials_evaluations"]
    assert A("snark_open") == tc["evaluation_proof"]
    assert [[F("snark_be", k) for k in range(3)]] == tc["base_polynomials_evaluations"]
    # gm::SnarkProof::new_elastic (gm_snark_new_elastic), flushes cut literally and merged: the time prover's proof
    assert vals["snark_elastic_equals_time"] == [["1"], ["1"]]
    # gm::PsnarkInstance / gm::PsnarkProof: the Python mirror's index and proof of the same instance
    same = lambda key, k, want: jac_to_affine_ints(oracle, J(key, k)) == jac_to_affine_ints(oracle, want)  # noqa: E731
    assert vals["psnark_nnz"][0] == [str(nnz)]
    assert all(same("psnark_index", k, pindex[k]) for k in range(5))
    assert same("psnark_witness", 0, pexp.witness_commitment) and (J("psnark_zc_alpha") == pexp.zc_alpha).all()
    assert all(same("psnark_rstar", k, pexp.r_star_commitments[k]) for k in range(3)) and same("psnark_zstar", 0, pexp.z_star_commitment)
    for k, c in enumerate((pexp.sorted_r_commitment, pexp.sorted_alpha_commitment, pexp.sorted_z_commitment)):
        assert same("psnark_sorted", k, c)
    for k, v in ((0, pexp.set_r_ep), (1, pexp.subset_r_ep), (3, pexp.set_alpha_ep), (4, pexp.subset_alpha_ep), (6, pexp.set_z_ep), (7, pexp.subset_z_ep)):
        assert (J("psnark_product", k) == v).all(), k
    assert all(same("psnark_accv", k, pexp.ep_msgs.acc_v_commitments[k]) for k in range(9))
    assert all((J("psnark_rstars_val", k) == pexp.rstars_vals[k]).all() for k in range(2))
    assert same("psnark_mu_proof", 0, pexp.ralpha_star_acc_mu_proof) and same("psnark_open", 0, pexp.tensorcheck_proof.evaluation_proof)
    assert [int(x) for x in vals["psnark_rounds"][0]] == [len(pexp.first_sumcheck_msgs[0]), len(pexp.second_sumcheck_msgs[0]), len(pexp.third_sumcheck_msgs[0]),
                                                          len(pexp.tensorcheck_proof.folded_polynomials_commitments)]
    pr1cs.free()
    pck.powers_of_g.free()


def test_sharded_provers_from_plain_c_abi_processes(oracle, pyref, tmp_path):
    """tests/cpp/test_sharded_ranks.cpp: forked PROCESSES with nothing but the C ABI -- gm_dist_init_shm, gm_snark_shard_key_new,
    gm_snark_new_time_sharded (block-diagonal and general-matrix form) and gm_snark_new_time over gm_g1_srs_register_cyclic shares
    -- must all produce the digest of the one-process run (what a Rust embedder of include/gemini_hip.h gets; no Python, no torch
    in the ranks)."""
    exe = str(tmp_path / "test_sharded_ranks")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_sharded_ranks.cpp"),
                           "-L", os.path.join(ROOT, "gemini_amd"), "-lgemini_hip", "-Wl,-rpath," + os.path.join(ROOT, "gemini_amd"), "-o", exe])
    R = pyref.R_MOD
    e_i = oracle.limbs_to_ints(oracle.random_fr(9001, 1))[0]
    tau_i = oracle.limbs_to_ints(oracle.random_fr(9002, 1))[0]
    hexl = lambda limbs: "".join(f"{int(v):016x}" for v in limbs)  # noqa: E731
    ev = oracle.fr_to_mont(oracle.ints_to_limbs([e_i, pow(e_i, -1, R)], 4))
    args = [hexl(ev[0]), hexl(ev[1]), hexl(oracle.g1_generator()), hexl(oracle.ints_to_limbs([tau_i], 4)[0])]

    def run(world, mode, cols="local", logn=10):
        out = subprocess.run([exe, str(world), str(logn), mode, cols] + args, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.split()[1]

    one = run(1, "cyclic")
    assert run(1, "block") == one
    for world in (8,):  # (1 / 2 / 4 / 8 of the same prover through Python launchers: tests/test_gpu_dist_native.py)
        assert run(world, "block") == one, world
    assert run(4, "block", "global") == one
    for world in (3, 5):
        assert run(world, "cyclic") == one, world
    # psnark with every vector in blocks at the level of its family (gm_psnark_new_time_sharded): blocks of dummy_r1cs built in C from closed forms,
    # any world size
    import gemini_amd as gm
    from gemini_amd import g2 as G2

    g2_file = tmp_path / "g2.bin"
    g2_file.write_bytes(G2.serialize_vec_uncompressed([G2.mul(G2.generator(), pow(tau_i, i, R)) for i in range(6)]))

    def run_p(world, logn=9):
        out = subprocess.run([exe, str(world), str(logn), "psnark", "global"] + args + [str(g2_file)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.split()[1]

    p_one = run_p(1)  # (one process through the same entry; that world 1 equals gm_psnark_new_time byte for byte: tests/test_gpu_dist_native.py)
    assert run_p(5) == p_one
    # and the digest is the one of the Python mirror's single-GPU native prover on the same instance and key
    import gemini_amd as gm
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof

    gm.capi.init()
    n = 1 << 10
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau_i], 4)[0])
    r1cs = dummy_r1cs(e_i, n)
    p = Proof.new_time(r1cs, ck, native=True)

    def fnv(h, b):
        for x in b:
            h = ((h ^ x) * 0x100000001B3) & (2**64 - 1)
        return h

    A = lambda a: np.ascontiguousarray(a, dtype=np.uint64).tobytes()  # noqa: E731
    h = fnv(0xCBF29CE484222325, A(p.witness_commitment))
    h = fnv(h, A(p.zc_alpha))
    for msgs, ff in (p.first_sumcheck_msgs, p.second_sumcheck_msgs):
        h = fnv(h, len(msgs).to_bytes(8, "little"))
        h = fnv(h, b"".join(A(a) + A(b) for a, b in msgs))
        h = fnv(h, A(ff[0][0]) + A(ff[0][1]))
    tc = p.tensorcheck_proof
    h = fnv(h, len(tc.folded_polynomials_commitments).to_bytes(8, "little"))
    h = fnv(h, b"".join(A(c) for c in tc.folded_polynomials_commitments))
    h = fnv(h, b"".join(A(e2) for e2 in tc.folded_polynomials_evaluations))
    h = fnv(h, A(tc.evaluation_proof))
    h = fnv(h, A(tc.base_polynomials_evaluations[0]))
    assert f"{h:016x}" == one
    r1cs.free()
    ck.powers_of_g.free()


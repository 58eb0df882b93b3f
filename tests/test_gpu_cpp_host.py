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
    with open(inp, "wb") as fh:
        for a in (rust, sc, mont, f, g, tw, ev, srs_rust):
            _wvec(fh, a)
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

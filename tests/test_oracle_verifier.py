"""CPU: the verifier restatement (oracle/verifier_ref.py, oracle/pairing.py) is pinned by what a verifier must do --
the pairing is bilinear and non-degenerate, honest proofs are accepted, any single altered element is rejected.
The GPU tests (tests/test_gpu_snark.py) then feed it the proofs of the HIP path."""
import copy

import pytest

from oracle import pairing as E
from oracle import pyref as P
from oracle import snark_ref as sr
from oracle import verifier_ref as V

R = P.R_MOD
G1 = (P.G1_X, P.G1_Y)


def test_pairing_is_bilinear_and_non_degenerate():
    a, b = 0x1234567, 0xfedcba987
    e = E.pairing(E.G2_GEN, G1)
    assert e != E.ONE
    assert E.f12_pow(e, R) == E.ONE  # lands in the order-r subgroup of F_q^12
    assert E.pairing(E.g2_mul(E.G2_GEN, b), P.g1_mul(G1, a)) == E.f12_pow(e, a * b)
    assert E.pairing(E.G2_GEN, P.g1_mul(G1, a * b % R)) == E.f12_pow(e, a * b)
    # additive in the first argument
    p, q = P.g1_mul(G1, 77), P.g1_mul(G1, 1000003)
    assert E.f12_mul(E.pairing(E.G2_GEN, p), E.pairing(E.G2_GEN, q)) == E.pairing(E.G2_GEN, P.g1_add(p, q))
    assert E.pairing_product_is_one([(P.g1_mul(G1, a), E.g2_mul(E.G2_GEN, b)), (P.g1_neg(P.g1_mul(G1, a * b % R)), E.G2_GEN)])
    assert not E.pairing_product_is_one([(P.g1_mul(G1, a), E.g2_mul(E.G2_GEN, b)), (P.g1_neg(P.g1_mul(G1, a * b % R + 1)), E.G2_GEN)])
    assert E.pairing_product_is_one([(None, E.G2_GEN), (G1, None)])


def test_kzg_opening_against_the_pairing_check():
    """src/kzg/time.rs:193-211 shape: commit, open at three points, verify; a wrong evaluation is rejected"""
    tau = 0x1F2E3D4C5B6A79880123456789ABCDEF % R
    srs = sr.srs(tau, 40)
    vk = V.VerifierKey.from_trapdoor(tau, 5)
    rng = P.SplitMix64(5)
    polys = [[rng.fr() for _ in range(n)] for n in (33, 17, 9)]
    pts = [rng.fr() for _ in range(3)]
    chal = rng.fr()
    comms = [sr.commit(srs, p) for p in polys]
    evals = [[P.evaluate_le(p, x) for x in pts] for p in polys]
    proof = sr.batch_open_multi_points(srs, polys, pts, chal)
    V.verify_multi_points(vk, comms, pts, evals, proof, chal)
    evals[1][2] = (evals[1][2] + 1) % R
    with pytest.raises(V.VerificationError):
        V.verify_multi_points(vk, comms, pts, evals, proof, chal)


@pytest.fixture(scope="module")
def honest():
    n = 8
    e, tau = 0x123456789ABCDEF0FEDCBA9876543210 % R, 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0 % R
    r1cs = sr.dummy_r1cs(e, n)
    proof = sr.snark_new_time(r1cs, sr.srs(tau, 2 * n + 1))
    return r1cs, proof, V.VerifierKey.from_trapdoor(tau, 5)


def test_verifier_accepts_the_restated_prover(honest):
    r1cs, proof, vk = honest
    V.snark_verify(proof, r1cs, vk)
    # a general (non-diagonal) instance
    from tests.util import random_r1cs_instance

    inst, tau = random_r1cs_instance(P, sr, 16, 99, nx=2)
    V.snark_verify(sr.snark_new_time(inst, sr.srs(tau, 2 * 16 + 1)), inst, V.VerifierKey.from_trapdoor(tau, 5))


def _tamper(proof, path):
    p = copy.deepcopy(proof)
    node = p
    for k in path[:-1]:
        node = node[k]
    v = node[path[-1]]
    if isinstance(v, tuple) and len(v) == 2 and v[0] > (1 << 300):  # a G1 point: replace by its double
        nv = P.g1_add(v, v)
    else:
        nv = (v + 1) % R
    if isinstance(node, tuple):
        raise AssertionError("path must end in a list or dict slot")
    node[path[-1]] = nv
    return p


@pytest.mark.parametrize("path", [
    ("zc_alpha",),
    ("witness_commitment",),
    ("tensorcheck_proof", "folded_polynomials_commitments", 0),
    ("tensorcheck_proof", "folded_polynomials_evaluations", 1, 0),
    ("tensorcheck_proof", "base_polynomials_evaluations", 0, 0),
    ("tensorcheck_proof", "evaluation_proof"),
])
def test_verifier_rejects_an_altered_element(honest, path):
    r1cs, proof, vk = honest
    with pytest.raises(V.VerificationError):
        V.snark_verify(_tamper(proof, path), r1cs, vk)


def test_verifier_rejects_altered_sumcheck_messages_and_a_wrong_key(honest):
    r1cs, proof, vk = honest
    for which in ("first_sumcheck_msgs", "second_sumcheck_msgs"):
        msgs, ff = proof[which]
        bad = copy.deepcopy(proof)
        m = list(msgs)
        m[1] = (m[1][0], (m[1][1] + 1) % R)
        bad[which] = (m, ff)
        with pytest.raises(V.VerificationError):
            V.snark_verify(bad, r1cs, vk)
        bad = copy.deepcopy(proof)
        bad[which] = (list(msgs), ((ff[0] + 1) % R, ff[1]))
        with pytest.raises(V.VerificationError):
            V.snark_verify(bad, r1cs, vk)
    # the G2 half of the key matters: powers of another trapdoor reject the honest proof
    other = V.VerifierKey.from_trapdoor(12345, 5)
    with pytest.raises(V.VerificationError):
        V.snark_verify(proof, r1cs, V.VerifierKey(vk.powers_of_g, other.powers_of_g2))
    # and so does the instance: another public input
    r2 = dict(r1cs)
    r2["x"] = [(r1cs["x"][0] + 1) % R]
    with pytest.raises(V.VerificationError):
        V.snark_verify(proof, r2, vk)


@pytest.fixture(scope="module")
def honest_psnark():
    from oracle import psnark_ref as pr

    from tests.util import random_r1cs_instance

    n = 8
    inst, tau = random_r1cs_instance(P, sr, n, 2024)
    a, b, c = inst["a"], inst["b"], inst["c"]
    srs = sr.srs(tau, 12 * n + 1)
    index = pr.index(srs, inst)
    proof = pr.psnark_new_time(srs, pr.powers_of_g2(tau, 3), inst, index)
    jm = pr.sum_matrices(a, b, c, n)
    nnz = len(pr.joint_matrices(jm, a, b, c)[0])
    return inst, proof, V.VerifierKey.from_trapdoor(tau, 3), index, nnz


def test_psnark_verifier_accepts_the_restated_prover_and_rejects_alterations(honest_psnark):
    """src/psnark/tests.rs:130-145 (test_psnark_correctness: `time_proof.verify(&r1cs, &vk, &index, num_non_zero).is_ok()`)"""
    inst, proof, vk, index, nnz = honest_psnark
    V.psnark_verify(proof, inst, vk, index, nnz)
    for path in (("zc_alpha",), ("rstars_vals", 1), ("ralpha_star_acc_mu_evals", 3), ("ep_msgs", "claimed_sumchecks", 4), ("set_z_ep",),
                 ("sorted_alpha_commitment",), ("ep_msgs", "acc_v_commitments", 8), ("ralpha_star_acc_mu_proof",),
                 ("tensorcheck_proof", "base_polynomials_evaluations", 17, 2), ("tensorcheck_proof", "folded_polynomials_evaluations", 0, 1),
                 ("tensorcheck_proof", "evaluation_proof")):
        with pytest.raises(V.VerificationError):
            V.psnark_verify(_tamper(proof, path), inst, vk, index, nnz)
    msgs, ff = proof["third_sumcheck_msgs"]
    bad = copy.deepcopy(proof)
    ff2 = [tuple(f) for f in ff]
    ff2[12] = (ff2[12][0], (ff2[12][1] + 1) % R)
    bad["third_sumcheck_msgs"] = (list(msgs), ff2)
    with pytest.raises(V.VerificationError):
        V.psnark_verify(bad, inst, vk, index, nnz)
    # the index commitments are part of the statement
    with pytest.raises(V.VerificationError):
        V.psnark_verify(proof, inst, vk, [index[1], index[0]] + list(index[2:]), nnz)
    with pytest.raises(V.VerificationError):
        V.psnark_verify(proof, inst, vk, index, nnz + 1)


def test_reference_example_key_is_one_power_short():
    """examples/psnark.rs:76 (`CommitterKey::new(num_constraints + num_variables, 5, rng)`, i.e. 2n + 1 powers for
    dummy_r1cs(n)) against the prover's longest polynomials: the accumulated products of the three sorted vectors have
    set_len + nnz + 2 = 2n + 2 coefficients, `msm_unchecked` drops the top one (src/kzg/time.rs:82) and the proof is
    rejected; with one more power it is accepted.  The example only times the prover and never verifies; the test key of
    src/psnark/tests.rs:137 (num_non_zero + num_variables + num_constraints) is long enough."""
    from oracle import psnark_ref as pr

    n = 16
    e, tau = 987654321987654321, 1234567890123456789012345
    inst = sr.dummy_r1cs(e, n)
    vk = V.VerifierKey.from_trapdoor(tau, 5)
    verdict = {}
    for max_degree in (2 * n, 2 * n + 1):
        srs = sr.srs(tau, max_degree + 1)
        index = pr.index(srs, inst)
        proof = pr.psnark_new_time(srs, pr.powers_of_g2(tau, 5), inst, index)
        try:
            V.psnark_verify(proof, inst, vk, index, n)
            verdict[max_degree] = True
        except V.VerificationError:
            verdict[max_degree] = False
    assert verdict == {2 * n: False, 2 * n + 1: True}


def test_c_backed_matrix_evaluations_equal_the_generic_ones(honest):
    """the hook that lets the verifier take 2^24-constraint dummy proofs (C vector passes) computes what the Python loops do"""
    n = 64
    e, tau = 0xABCDEF0123456789 % R, 0x13579BDF02468ACE13579BDF % R
    inst = sr.dummy_r1cs(e, n)
    proof = sr.snark_new_time(inst, sr.srs(tau, 2 * n + 1))
    vk = V.VerifierKey.from_trapdoor(tau, 5)
    V.snark_verify(proof, inst, vk)
    V.snark_verify(proof, {"a": range(n), "x": [e]}, vk, m_of=V.dummy_matrix_evaluations(e, n))
    rng = P.SplitMix64(11)
    ch1, alpha, beta, etas = [rng.fr() for _ in range(6)], rng.fr(), rng.fr(), P.powers(rng.fr(), 3)
    t, ap, bp = P.tensor(ch1), P.powers(alpha, n), P.powers(beta, n)
    a_bp = sr.matvec(inst["a"], bp)
    want = P.ip([P.ip(a_bp, [x * y % R for x, y in zip(t, ap)]), sum(x * y for x, y in zip(a_bp, t)) % R, P.ip(a_bp, ap)], etas)
    assert V.dummy_matrix_evaluations(e, n)(beta, ch1, alpha, etas) == want
    # ... and the O(log n) closed form of the same hook (the verifier of a 2^28-constraint proof)
    assert V.dummy_matrix_evaluations_closed_form(e, n)(beta, ch1, alpha, etas) == want
    assert V.dummy_matrix_evaluations_closed_form(e, n)((-beta) % R, ch1, alpha, etas) == V.dummy_matrix_evaluations(e, n)((-beta) % R, ch1, alpha, etas)
    V.snark_verify(proof, {"a": range(n), "x": [e]}, vk, m_of=V.dummy_matrix_evaluations_closed_form(e, n))
    # the generator-copies key of examples/snark.rs:59-63 is the key of the trapdoor tau = 1
    p1 = sr.snark_new_time(inst, sr.srs(1, 2 * n + 1))
    V.snark_verify(p1, {"a": range(n), "x": [e]}, V.VerifierKey.from_trapdoor(1, 3), m_of=V.dummy_matrix_evaluations_closed_form(e, n))
    # a proof for another instance value is rejected through the hook as well
    with pytest.raises(V.VerificationError):
        V.snark_verify(proof, {"a": range(n), "x": [e]}, vk, m_of=V.dummy_matrix_evaluations((e + 1) % R, n))

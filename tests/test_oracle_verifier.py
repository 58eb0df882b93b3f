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
    n = 16
    rng = P.SplitMix64(99)
    z = [rng.fr() for _ in range(n)]
    mk = lambda: [[(rng.fr(), int(rng.next() % n)) for _ in range(1 + int(rng.next() % 3))] for _ in range(n)]  # noqa: E731
    a, b = mk(), mk()
    za, zb = sr.matvec(a, z), sr.matvec(b, z)
    c = [[(za[i] * zb[i] % R * pow(z[i], -1, R) % R, i)] for i in range(n)]
    inst = {"a": a, "b": b, "c": c, "z": z, "w": z[2:], "x": z[:2]}
    tau = rng.fr()
    V.snark_verify(sr.snark_new_time(inst, sr.srs(tau, 2 * n + 1)), inst, V.VerifierKey.from_trapdoor(tau, 5))


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

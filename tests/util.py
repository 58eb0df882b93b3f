"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def rand_bases(orc, seed: int, n: int) -> np.ndarray:
    """n affine points k_i * G (Montgomery, (n, 12)) from the C oracle's fixed-base routine."""
    ks = orc.random_fr(seed, n)
    return orc.g1_fixed_base_mul(orc.g1_generator(), ks)


def jac_to_affine_ints(orc, jac):
    return orc.affine_to_ints(orc.g1_to_affine(jac))


def assert_same_point(orc, got_jac, exp_jac):
    """bit-exact after normalisation: both sides reduced to canonical affine integers."""
    assert jac_to_affine_ints(orc, got_jac) == jac_to_affine_ints(orc, exp_jac)


def is_normalised(orc, jac) -> bool:
    jac = np.asarray(jac, dtype=np.uint64).reshape(3, 6)
    one = orc.fq_to_mont(orc.ints_to_limbs([1], 6))[0]
    if not jac[2].any():
        return bool((jac[0] == one).all() and (jac[1] == one).all())
    return bool((jac[2] == one).all())


def snark_proof_to_ints(gm, orc, proof) -> dict:
    """a device `snark::Proof` in the layout oracle/snark_ref.py and oracle/verifier_ref.py use"""
    I = gm.fr.fr_to_int
    J = lambda p: jac_to_affine_ints(orc, p)  # noqa: E731
    msgs = lambda m: ([(I(a), I(b)) for a, b in m[0]], (I(m[1][0][0]), I(m[1][0][1])))  # noqa: E731
    tc = proof.tensorcheck_proof
    return {"witness_commitment": J(proof.witness_commitment), "zc_alpha": I(proof.zc_alpha),
            "first_sumcheck_msgs": msgs(proof.first_sumcheck_msgs), "second_sumcheck_msgs": msgs(proof.second_sumcheck_msgs),
            "tensorcheck_proof": {"folded_polynomials_commitments": [J(c) for c in tc.folded_polynomials_commitments],
                                  "folded_polynomials_evaluations": [[I(x) for x in e2] for e2 in tc.folded_polynomials_evaluations],
                                  "evaluation_proof": J(tc.evaluation_proof),
                                  "base_polynomials_evaluations": [[I(x) for x in e3] for e3 in tc.base_polynomials_evaluations]}}

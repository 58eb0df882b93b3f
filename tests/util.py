"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def rand_bases(orc, seed: int, n: int) -> np.ndarray:
    """n affine points k_i * G (Montgomery, (n, 12)) from the C oracle's fixed-base routine."""
    ks = orc.random_fr(seed, n)
    return orc.g1_fixed_base_mul(orc.g1_generator(), ks)


def dot_ints(a: np.ndarray, k: np.ndarray) -> int:
    """sum_i a_i * k_i as an exact integer, a and k (n, 4) little-endian u64 limbs.  No oracle involved: 16-bit limbs as
    float64 columns, one 16 x 16 BLAS product per block of 2^20 rows (partial sums < 2^32 * 2^20 = 2^52, exact), recombined
    with Python integers.  With bases k_i * G the whole MSM result must be (dot mod r) * G -- a check of ALL n pairs that
    takes about a second at 2^24."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    k = np.ascontiguousarray(k, dtype=np.uint64).reshape(-1, 4)
    assert a.shape == k.shape
    total = 0
    for lo in range(0, len(a), 1 << 20):
        A = a[lo: lo + (1 << 20)].view(np.uint16).reshape(-1, 16).astype(np.float64)
        K = k[lo: lo + (1 << 20)].view(np.uint16).reshape(-1, 16).astype(np.float64)
        M = A.T @ K
        assert M.max() < 2.0 ** 53
        for j in range(16):
            for l in range(16):
                total += int(M[j, l]) << (16 * (j + l))
    return total


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


def random_r1cs_instance(pyref, sr, n: int, seed: int, nx: int = 1):
    """A satisfied random R1CS over n variables / n constraints in the layout of oracle/snark_ref.py: A and B have 1-3
    entries per row in DISTINCT columns (rows of ark-relations matrices never repeat a column; the joint-matrix encoding of
    the preprocessing SNARK relies on it -- a proof for an instance with a repeated column is rejected by the verifier),
    C is the diagonal that makes Az . Bz = Cz.  Returns (instance, a further random field element)."""
    rng = pyref.SplitMix64(seed)
    R = pyref.R_MOD
    z = [rng.fr() for _ in range(n)]

    def mk():
        rows = []
        for _ in range(n):
            cols = []
            while len(cols) < 1 + int(rng.next() % 3):
                c = int(rng.next() % n)
                if c not in cols:
                    cols.append(c)
            rows.append([(rng.fr(), c) for c in cols])
        return rows

    a, b = mk(), mk()
    za, zb = sr.matvec(a, z), sr.matvec(b, z)
    c = [[(za[i] * zb[i] % R * pow(z[i], -1, R) % R, i)] for i in range(n)]
    return {"a": a, "b": b, "c": c, "z": z, "w": z[nx:], "x": z[:nx]}, rng.fr()


def psnark_proof_to_ints(gm, orc, proof) -> dict:
    """a device `psnark::Proof` in the layout oracle/psnark_ref.py and oracle/verifier_ref.py use"""
    I = gm.fr.fr_to_int
    J = lambda p: jac_to_affine_ints(orc, p)  # noqa: E731

    def msgs(m):
        finals = [(I(a), I(b)) for a, b in m[1]]
        return ([(I(a), I(b)) for a, b in m[0]], finals[0] if len(finals) == 1 else finals)

    tc = proof.tensorcheck_proof
    out = {k: J(getattr(proof, k)) for k in ("witness_commitment", "z_star_commitment", "sorted_r_commitment", "sorted_alpha_commitment",
                                             "sorted_z_commitment", "ralpha_star_acc_mu_proof")}
    out.update({k: I(getattr(proof, k)) for k in ("zc_alpha", "set_r_ep", "subset_r_ep", "set_alpha_ep", "subset_alpha_ep", "set_z_ep", "subset_z_ep")})
    out.update({k: msgs(getattr(proof, k)) for k in ("first_sumcheck_msgs", "second_sumcheck_msgs", "third_sumcheck_msgs")})
    out["r_star_commitments"] = [J(c) for c in proof.r_star_commitments]
    out["ep_msgs"] = {"acc_v_commitments": [J(c) for c in proof.ep_msgs.acc_v_commitments], "claimed_sumchecks": [I(e) for e in proof.ep_msgs.claimed_sumchecks]}
    out["ralpha_star_acc_mu_evals"] = [I(e) for e in proof.ralpha_star_acc_mu_evals]
    out["rstars_vals"] = [I(e) for e in proof.rstars_vals]
    out["tensorcheck_proof"] = {"folded_polynomials_commitments": [J(c) for c in tc.folded_polynomials_commitments],
                                "folded_polynomials_evaluations": [[I(x) for x in e2] for e2 in tc.folded_polynomials_evaluations],
                                "evaluation_proof": J(tc.evaluation_proof),
                                "base_polynomials_evaluations": [[I(x) for x in e3] for e3 in tc.base_polynomials_evaluations]}
    return out

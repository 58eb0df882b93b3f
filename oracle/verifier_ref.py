"""CPU restatement of the reference's VERIFIER for the non-preprocessing SNARK (TEST INFRASTRUCTURE ONLY).

Follows src/snark/verifier.rs:19-119, src/subprotocols/sumcheck/subclaim.rs:23-43,77-96,
src/subprotocols/tensorcheck/mod.rs:96-107,286-392 and src/kzg/mod.rs:155-244 on Python integers, with the pairing of
oracle/pairing.py.  The prover is out of this file's sight: it consumes a proof in the layout oracle/snark_ref.py
returns (integers and affine integer points) and accepts or rejects it -- the acceptance predicate of the reference,
independent of how the proof was computed.  The parity tests feed it the proofs of the HIP path: an accepted proof
means every commitment opens to the claimed evaluations under the key (G1 and G2 halves), both sumchecks reduce to
their claims and the tensor relation holds, whatever restatement the prover side was compared with.
"""
from __future__ import annotations

from . import pairing as E
from . import pyref as P
from . import snark_ref as sr

R = P.R_MOD


class VerificationError(Exception):
    pass


def _ip_unsafe(a, b):
    """src/misc.rs:222-224: zip-truncating inner product"""
    return sum(x * y for x, y in zip(a, b)) % R


def _hadamard_unsafe(a, b):
    """src/misc.rs:210-212"""
    return [x * y % R for x, y in zip(a, b)]


# ---- src/subprotocols/sumcheck/subclaim.rs ---------------------------------------------------------------------
def subclaim_new(tr: P.GeminiTranscript, messages, final_foldings, asserted_sum: int):
    """:23-43 with reduce :77-96 -> (challenges, final_foldings)"""
    reduced = asserted_sum % R
    challenges = []
    for a, b in messages:
        tr.append_round_msg(b"evaluations", a, b)
        r = tr.get_challenge(b"challenge")
        challenges.append(r)
        c = (reduced - a) % R
        reduced = (a + r * b + c * r * r) % R  # a + b x + c x^2 at r
    tr.append_fr(b"final-folding", final_foldings[0])
    tr.append_fr(b"final-folding", final_foldings[1])
    if final_foldings[0] * final_foldings[1] % R != reduced:
        raise VerificationError("sumcheck: final foldings do not multiply to the reduced claim")
    return challenges, final_foldings


# ---- src/kzg/mod.rs --------------------------------------------------------------------------------------------
class VerifierKey:
    """:139-147 / src/kzg/time.rs:29-40: the first max_eval_points powers of g and max_eval_points + 1 powers of g2"""

    def __init__(self, powers_of_g, powers_of_g2):
        self.powers_of_g = list(powers_of_g)    # affine integer points
        self.powers_of_g2 = list(powers_of_g2)  # affine F_q^2 points

    @classmethod
    def from_trapdoor(cls, tau: int, max_eval_points: int, g=None, g2=None) -> "VerifierKey":
        g = (P.G1_X, P.G1_Y) if g is None else g
        g2 = E.G2_GEN if g2 is None else g2
        return cls([P.g1_mul(g, pow(tau, i, R)) for i in range(max_eval_points)],
                   [E.g2_mul(g2, pow(tau, i, R)) for i in range(max_eval_points + 1)])


def _poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R
    return out


def _g1_msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = P.g1_add(acc, P.g1_mul(p, s % R))
    return acc


def _g2_msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = E.g2_add(acc, E.g2_mul(p, s % R))
    return acc


def verify_multi_points(vk: VerifierKey, commitments, eval_points, evaluations, proof, open_chal: int) -> None:
    """:181-244: e(sum eta^i C_i - [I(tau)] g, g2) == e(proof, [Z(tau)] g2), I = the eta-combination of the
    interpolations of the claimed evaluations, Z = the vanishing polynomial of the points"""
    zeros = [1]
    for pt in eval_points:
        zeros = _poly_mul(zeros, [(-pt) % R, 1])
    zeros_g2 = _g2_msm(vk.powers_of_g2, zeros)
    sca_inverse = []
    for j, xj in enumerate(eval_points):
        sca = 1
        for k, xk in enumerate(eval_points):
            if j != k:
                sca = sca * (xj - xk) % R
        sca_inverse.append(pow(sca, -1, R))
    lang = []
    for j in range(len(eval_points)):
        lp = [1]
        for k, xk in enumerate(eval_points):
            if j != k:
                lp = _poly_mul(lp, [(-xk) % R, 1])
        lang.append(lp)
    etas = P.powers(open_chal, len(evaluations))
    interpolated = []
    for evals in evaluations:
        res = [0] * len(eval_points)
        for j, yj in enumerate(evals[: len(eval_points)]):
            f = sca_inverse[j] * yj % R
            for d, c in enumerate(lang[j]):
                res[d] = (res[d] + c * f) % R
        interpolated.append(res)
    i_poly = P.linear_combination(interpolated, etas)
    i_comm = _g1_msm(vk.powers_of_g, i_poly)
    if len(commitments) != len(etas):  # G::msm(..).unwrap() fails on a length mismatch
        raise VerificationError("verify_multi_points: one evaluation row per commitment")
    f_comm = _g1_msm(commitments, etas)
    lhs = P.g1_add(f_comm, P.g1_neg(i_comm))
    # e(lhs, g2) == e(proof, zeros)  <=>  e(lhs, g2) * e(-proof, zeros) == 1
    if not E.pairing_product_is_one([(lhs, vk.powers_of_g2[0]), (P.g1_neg(proof), zeros_g2)]):
        raise VerificationError("verify_multi_points: pairing check failed")


# ---- src/subprotocols/tensorcheck/mod.rs -----------------------------------------------------------------------
def evaluate_sq_fp(pos: int, neg: int, rho: int, two_inv: int, two_beta_inv: int) -> int:
    """:96-107  f'(beta^2) = (f(beta) + f(-beta)) / 2 + rho (f(beta) - f(-beta)) / (2 beta)"""
    return ((pos + neg) * two_inv + (pos - neg) * rho % R * two_beta_inv) % R


def tensorcheck_verify(tc, tr, vk, asserted_res_vec, base_commitments, direct_base_evals, fold_randomness, eval_chal, batch_challenge):
    """:286-392"""
    minus = (-eval_chal) % R
    chal2 = eval_chal * eval_chal % R
    two_inv = pow(2, -1, R)
    two_beta_inv = pow(2 * eval_chal % R, -1, R)
    evaluations = [list(e) for e in tc["base_polynomials_evaluations"]]
    fold_evals = tc["folded_polynomials_evaluations"]
    offset = 0
    for instance, randomness in enumerate(fold_randomness):
        rounds = len(randomness) - 1
        base = direct_base_evals[instance]
        fe = fold_evals[offset: offset + rounds]
        if len(fe) != rounds:
            raise VerificationError("tensorcheck: missing folded evaluations")
        asserted = asserted_res_vec[instance]
        offset += rounds
        evaluations.append([evaluate_sq_fp(base[0], base[1], randomness[0], two_inv, two_beta_inv), fe[0][0], fe[0][1]])
        for i in range(1, rounds):
            evaluations.append([evaluate_sq_fp(fe[i - 1][0], fe[i - 1][1], randomness[i], two_inv, two_beta_inv), fe[i][0], fe[i][1]])
        subclaim = evaluate_sq_fp(fe[rounds - 1][0], fe[rounds - 1][1], randomness[rounds], two_inv, two_beta_inv)
        if subclaim != P.ip(asserted, P.powers(batch_challenge, len(asserted))):
            raise VerificationError("tensorcheck: the last folding does not meet the asserted value")
    all_commitments = list(base_commitments) + list(tc["folded_polynomials_commitments"])
    for e3 in tc["base_polynomials_evaluations"]:
        for e in e3:
            tr.append_fr(b"eval", e)
    for e2 in fold_evals:
        for e in e2:
            tr.append_fr(b"eval", e)
    open_chal = tr.get_challenge(b"open-chal")
    verify_multi_points(vk, all_commitments, [chal2, eval_chal, minus], evaluations, tc["evaluation_proof"], open_chal)


# ---- src/snark/verifier.rs -------------------------------------------------------------------------------------
def snark_verify(proof, r1cs, vk: VerifierKey) -> None:
    """:19-119; raises VerificationError on rejection.  `r1cs` as in snark_ref (rows of (value, column) pairs)."""
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    tr.append_message(b"witness", P.g1_serialize_uncompressed(proof["witness_commitment"]))
    alpha = tr.get_challenge(b"alpha")
    tr.append_fr(b"zc(alpha)", proof["zc_alpha"])
    m1, ff1 = proof["first_sumcheck_msgs"]
    ch1, ff1 = subclaim_new(tr, m1, ff1, proof["zc_alpha"])
    eta = tr.get_challenge(b"eta")
    etas = P.powers(eta, 3)
    num_constraints = len(r1cs["a"])
    tensor_challenges = P.tensor(ch1)
    alpha_powers = P.powers(alpha, num_constraints)
    hadamard_randomness = _hadamard_unsafe(tensor_challenges, alpha_powers)
    asserted_sum_2 = P.ip([ff1[0], ff1[1], proof["zc_alpha"]], etas)
    m2, ff2 = proof["second_sumcheck_msgs"]
    ch2, ff2 = subclaim_new(tr, m2, ff2, asserted_sum_2)
    tc = proof["tensorcheck_proof"]
    gamma = tr.get_challenge(b"batch_challenge")
    for c in tc["folded_polynomials_commitments"]:
        tr.append_message(b"commitment", P.g1_serialize_uncompressed(c))
    beta = tr.get_challenge(b"evaluation-chal")
    beta_powers = P.powers(beta, num_constraints)
    minus_beta_powers = P.powers((-beta) % R, num_constraints)

    def m_of(pw):
        return P.ip([P.ip(sr.matvec(r1cs["a"], pw), hadamard_randomness), _ip_unsafe(sr.matvec(r1cs["b"], pw), tensor_challenges),
                     P.ip(sr.matvec(r1cs["c"], pw), alpha_powers)], etas)

    m_pos, m_neg = m_of(beta_powers), m_of(minus_beta_powers)
    x = r1cs["x"]
    beta_power = beta_powers[len(x)]
    base_evals = tc["base_polynomials_evaluations"][0]
    z_pos = (P.evaluate_le(x, beta) + beta_power * base_evals[1]) % R
    if len(x) & 1:
        beta_power = (-beta_power) % R
    z_neg = (P.evaluate_le(x, (-beta) % R) + beta_power * base_evals[2]) % R
    direct = [[(m_pos + gamma * z_pos) % R, (m_neg + gamma * z_neg) % R]]
    tensorcheck_verify(tc, tr, vk, [list(ff2)], [proof["witness_commitment"]], direct, [ch2], beta, gamma)

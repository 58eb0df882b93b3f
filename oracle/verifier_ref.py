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


def verify(vk: VerifierKey, commitment, alpha: int, evaluation: int, proof) -> None:
    """:155-175: e(C - [evaluation] g, g2) == e(proof, [tau - alpha] g2)"""
    ep = _g2_msm(vk.powers_of_g2, [(-alpha) % R, 1])
    lhs = P.g1_add(commitment, P.g1_neg(P.g1_mul(vk.powers_of_g[0], evaluation % R)))
    if not E.pairing_product_is_one([(lhs, vk.powers_of_g2[0]), (P.g1_neg(proof), ep)]):
        raise VerificationError("verify: pairing check failed")


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
def dummy_matrix_evaluations(e: int, n: int):
    """The O(n) part of the verifier (src/snark/verifier.rs:63-88) for dummy_r1cs(e, n) -- A = B = C = diag(1 / e)
    (src/circuit.rs:349-365) -- with every vector pass a call into the C restatement (oracle/gemini_oracle.c: powers,
    tensor, hadamard, ip) instead of a Python loop: the verifier of a 2^24-constraint proof in seconds.  Returns the
    `m_of` hook of snark_verify; the generic (Python) hook computes the same values (tests/test_oracle_verifier.py)."""
    import numpy as np

    from . import oracle as orc

    M = lambda v: orc.fr_to_mont(orc.ints_to_limbs([v % R], 4))[0]  # noqa: E731
    I = lambda x: orc.limbs_to_ints(orc.fr_from_mont(np.asarray(x, dtype=np.uint64).reshape(1, 4)))[0]  # noqa: E731
    inv_e = pow(e, -1, R)
    cache = {}

    def m_of(point: int, ch1, alpha: int, etas):
        if "tensor" not in cache:
            t = orc.tensor(np.stack([M(c) for c in ch1]))
            ap = orc.powers(M(alpha), n)
            cache["tensor"], cache["alpha"], cache["had"] = t, ap, orc.hadamard(t[:n], ap)
        bp = orc.powers(M(point), n)  # A * powers = powers / e for the diagonal matrices
        parts = [I(orc.ip(bp, cache["had"])), I(orc.ip(bp, cache["tensor"][:n])), I(orc.ip(bp, cache["alpha"]))]
        return inv_e * P.ip(parts, etas) % R

    return m_of


def dummy_matrix_evaluations_closed_form(e: int, n: int):
    """The same three inner products for dummy_r1cs(e, n), n a power of two, in O(log n): with A = B = C = diag(1 / e),
    A * powers(x) = powers(x) / e, and  <powers(x), tensor(rho)> = prod_j (1 + rho_j x^(2^j))  (evaluate_tensor_poly,
    src/misc.rs:373-382),  <powers(x), tensor(rho) o powers(alpha)> = the same product at alpha x,  <powers(x), powers(alpha)> =
    geometric(alpha x, n) (src/misc.rs:387-389).  Held equal to the O(n) hook above in tests/test_oracle_verifier.py; it is what
    lets the verifier run on a 2^28-constraint proof."""
    inv_e = pow(e, -1, R)

    def m_of(point: int, ch1, alpha: int, etas):
        assert 1 << len(ch1) == n
        ax = alpha * point % R
        geo = n % R if ax == 1 else (pow(ax, n, R) - 1) * pow(ax - 1, -1, R) % R
        parts = [_tensor_poly(ch1, ax), _tensor_poly(ch1, point), geo]
        return inv_e * P.ip(parts, etas) % R

    return m_of


def _tensor_poly(elements, x: int) -> int:
    res, s = 1, x % R
    for el in elements:
        res = res * (1 + el * s) % R
        s = s * s % R
    return res


def snark_verify(proof, r1cs, vk: VerifierKey, m_of=None) -> None:
    """:19-119; raises VerificationError on rejection.  `r1cs` as in snark_ref (rows of (value, column) pairs); `m_of`
    (optional) replaces the O(n) evaluation of the matrices at the powers of +-beta (dummy_matrix_evaluations)."""
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    tr.append_message(b"witness", P.g1_serialize_uncompressed(proof["witness_commitment"]))
    alpha = tr.get_challenge(b"alpha")
    tr.append_fr(b"zc(alpha)", proof["zc_alpha"])
    m1, ff1 = proof["first_sumcheck_msgs"]
    ch1, ff1 = subclaim_new(tr, m1, ff1, proof["zc_alpha"])
    eta = tr.get_challenge(b"eta")
    etas = P.powers(eta, 3)
    num_constraints = len(r1cs["a"])
    asserted_sum_2 = P.ip([ff1[0], ff1[1], proof["zc_alpha"]], etas)
    m2, ff2 = proof["second_sumcheck_msgs"]
    ch2, ff2 = subclaim_new(tr, m2, ff2, asserted_sum_2)
    tc = proof["tensorcheck_proof"]
    gamma = tr.get_challenge(b"batch_challenge")
    for c in tc["folded_polynomials_commitments"]:
        tr.append_message(b"commitment", P.g1_serialize_uncompressed(c))
    beta = tr.get_challenge(b"evaluation-chal")
    if m_of is None:
        tensor_challenges = P.tensor(ch1)
        alpha_powers = P.powers(alpha, num_constraints)
        hadamard_randomness = _hadamard_unsafe(tensor_challenges, alpha_powers)

        def generic(pw):
            return P.ip([P.ip(sr.matvec(r1cs["a"], pw), hadamard_randomness), _ip_unsafe(sr.matvec(r1cs["b"], pw), tensor_challenges),
                         P.ip(sr.matvec(r1cs["c"], pw), alpha_powers)], etas)

        m_pos, m_neg = generic(P.powers(beta, num_constraints)), generic(P.powers((-beta) % R, num_constraints))
    else:
        m_pos, m_neg = m_of(beta, ch1, alpha, etas), m_of((-beta) % R, ch1, alpha, etas)
    x = r1cs["x"]
    beta_power = pow(beta, len(x), R)
    base_evals = tc["base_polynomials_evaluations"][0]
    z_pos = (P.evaluate_le(x, beta) + beta_power * base_evals[1]) % R
    if len(x) & 1:
        beta_power = (-beta_power) % R
    z_neg = (P.evaluate_le(x, (-beta) % R) + beta_power * base_evals[2]) % R
    direct = [[(m_pos + gamma * z_pos) % R, (m_neg + gamma * z_neg) % R]]
    tensorcheck_verify(tc, tr, vk, [list(ff2)], [proof["witness_commitment"]], direct, [ch2], beta, gamma)


# ---- src/psnark/verifier.rs ------------------------------------------------------------------------------------
def evaluate_tensor_poly(elements, x: int) -> int:
    """src/misc.rs:373-382"""
    res, s = 1, x % R
    for el in elements:
        res = res * (1 + el * s) % R
        s = s * s % R
    return res


def evaluate_geometric_poly(rx: int, n: int) -> int:
    """src/misc.rs:387-389: 1 + rx + ... + rx^(n-1)"""
    return (pow(rx, n, R) - 1) * pow(rx - 1, -1, R) % R


def evaluate_index_poly(x: int, n: int) -> int:
    """src/misc.rs:394-399: 0 + x + 2 x^2 + ... + (n-1) x^(n-1)"""
    assert x % R != 1
    x1 = (1 - x) % R
    x_n = pow(x, n - 1, R)
    return (x * (1 - x_n) % R * pow(x1 * x1 % R, -1, R) - (n - 1) * x_n % R * x % R * pow(x1, -1, R)) % R


def _plookup_subset_eval(subset_eval, index_eval, x, y, zeta, n):
    """:37-63: shift(f + zeta * index + y * geometric)(x) = x (..) + 1"""
    return (x * (subset_eval + zeta * index_eval + y * evaluate_geometric_poly(x, n)) + 1) % R


def _plookup_set_eval(set_eval, x, y, z, n):
    """:68-84: shift((1 + z) y geometric(n + 1) + (x + z) f)(x)"""
    return (x * ((1 + z) * y % R * evaluate_geometric_poly(x, n + 1) + (x + z) * set_eval) + 1) % R


def subclaim_new_batch(tr: P.GeminiTranscript, messages, final_foldings, asserted_sums):
    """src/subprotocols/sumcheck/subclaim.rs:45-75"""
    coefficients = [tr.get_challenge(b"batch-sumcheck") for _ in asserted_sums]
    reduced = P.ip(coefficients, [s % R for s in asserted_sums])
    challenges = []
    for a, b in messages:
        tr.append_round_msg(b"evaluations", a, b)
        r = tr.get_challenge(b"challenge")
        challenges.append(r)
        c = (reduced - a) % R
        reduced = (a + r * b + c * r * r) % R
    expected = 0
    for (f0, f1), coeff in zip(final_foldings, coefficients):
        tr.append_fr(b"final-folding-lhs", f0)
        tr.append_fr(b"final-folding-rhs", f1)
        expected = (expected + f0 * f1 % R * coeff) % R
    if expected != reduced:
        raise VerificationError("batched sumcheck: final foldings do not meet the reduced claim")
    return challenges, final_foldings


def psnark_verify(proof, r1cs, vk: VerifierKey, index, num_non_zero: int) -> None:
    """:86-565; raises VerificationError on rejection.  `proof` in the layout of psnark_ref.psnark_new_time, `index` the
    five index commitments (affine integer points)."""
    from .psnark_ref import g2_serialize_uncompressed

    G1 = P.g1_serialize_uncompressed
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    tr.append_message(b"witness", G1(proof["witness_commitment"]))
    tr.append_message(b"ck", len(vk.powers_of_g2).to_bytes(8, "little") + b"".join(g2_serialize_uncompressed(p) for p in vk.powers_of_g2))
    tr.append_message(b"instance", len(index).to_bytes(8, "little") + b"".join(G1(p) for p in index))
    alpha = tr.get_challenge(b"alpha")
    zc_alpha = proof["zc_alpha"]
    tr.append_fr(b"zc(alpha)", zc_alpha)
    m1, ff1 = proof["first_sumcheck_msgs"]
    ch1, ff1 = subclaim_new(tr, m1, ff1, zc_alpha)
    num_variables = len(r1cs["z"])
    for label, c in zip((b"ra*", b"rb*", b"rc*"), proof["r_star_commitments"]):
        tr.append_message(label, G1(c))
    tr.append_message(b"z*", G1(proof["z_star_commitment"]))
    eta = tr.get_challenge(b"chal")
    challenges = P.powers(eta, 3)
    asserted_sum_2 = (ff1[0] + ff1[1] * challenges[1] + zc_alpha * challenges[2]) % R
    m2, ff2 = proof["second_sumcheck_msgs"]
    ch2, ff2 = subclaim_new(tr, m2, ff2, asserted_sum_2)
    zeta = tr.get_challenge(b"zeta")
    for label, key in ((b"sorted_alpha_commitment", "sorted_alpha_commitment"), (b"sorted_r_commitment", "sorted_r_commitment"),
                       (b"sorted_z_commitment", "sorted_z_commitment")):
        tr.append_message(label, G1(proof[key]))
    y = tr.get_challenge(b"gamma")
    z = tr.get_challenge(b"chi")
    # (the labels repeat "set_r_ep" / "subset_r_ep" for the alpha products: :169-176)
    for label, key in ((b"set_r_ep", "set_alpha_ep"), (b"subset_r_ep", "subset_alpha_ep"), (b"set_r_ep", "set_r_ep"),
                       (b"subset_r_ep", "subset_r_ep"), (b"set_z_ep", "set_z_ep"), (b"subset_z_ep", "subset_z_ep")):
        tr.append_fr(label, proof[key])
    ep = proof["ep_msgs"]
    for c in ep["acc_v_commitments"]:
        tr.append_message(b"acc_v", G1(c))
    mu = tr.get_challenge(b"ep-chal")
    open_chal = tr.get_challenge(b"open-chal")
    commitments = [proof["r_star_commitments"][0]] + list(ep["acc_v_commitments"])
    mu_evals = proof["ralpha_star_acc_mu_evals"]
    verify_multi_points(vk, commitments, [mu], [[e] for e in mu_evals], proof["ralpha_star_acc_mu_proof"], open_chal)
    for e in mu_evals:
        tr.append_fr(b"ralpha_star_acc_mu", e)
    tr.append_message(b"ralpha_star_mu_proof", G1(proof["ralpha_star_acc_mu_proof"]))
    rstars = proof["rstars_vals"]
    asserted_sum_3 = list(ep["claimed_sumchecks"]) + list(rstars)
    asserted_sum_3.append((ff2[1] - rstars[0] - rstars[1] * eta) * pow(eta * eta % R, -1, R) % R)
    asserted_sum_3.append(mu_evals[0])
    m3, ff3 = proof["third_sumcheck_msgs"]
    ch3, ff3 = subclaim_new_batch(tr, m3, ff3, asserted_sum_3)
    batch_consistency = tr.get_challenge(b"batch_challenge")
    tc = proof["tensorcheck_proof"]
    for c in tc["folded_polynomials_commitments"]:
        tr.append_message(b"commitment", G1(c))
    beta = tr.get_challenge(b"evaluation-chal")
    nbeta = (-beta) % R
    res1 = [ff3[i][0] for i in range(9)] + [ff3[12][0]]
    res2 = [ff3[i][1] for i in range(9)] + [ff3[i][1] for i in range(9, 13)]
    res3 = [ff2[0]]
    res4 = [ff3[9][0], ff3[10][0], ff3[11][0]]
    be = tc["base_polynomials_evaluations"]
    bc = batch_consistency

    # first body: the nine accumulated products, then r*
    d1 = [0, 0]
    tmp = 1
    for i in list(range(13, 22)) + [2]:
        d1[0] = (d1[0] + tmp * be[i][1]) % R
        d1[1] = (d1[1] + tmp * be[i][2]) % R
        tmp = tmp * bc % R
    # second body: the nine shifted monic lookup vectors, then val_a, val_b, val_c, alpha*
    set_len = 1 << len(ch1)
    x = r1cs["x"]
    beta_power = pow(beta, len(x), R)
    z_pos = (P.evaluate_le(x, beta) + beta_power * be[0][1]) % R
    z_neg = (P.evaluate_le(x, nbeta) + (beta_power if len(x) % 2 == 0 else -beta_power) * be[0][2]) % R
    terms = []
    for pt, col, zval in ((beta, 1, z_pos), (nbeta, 2, z_neg)):
        terms.append([
            # lookup r*
            _plookup_set_eval((evaluate_tensor_poly(ch1, pt) + zeta * evaluate_index_poly(pt, set_len)) % R, pt, y, z, set_len),
            _plookup_subset_eval(be[2][col], be[5][col], pt, y, zeta, num_non_zero),
            _plookup_set_eval(be[10][col], pt, y, z, set_len + num_non_zero),
            # lookup alpha*
            _plookup_set_eval((evaluate_geometric_poly(alpha * pt % R, set_len) + zeta * evaluate_index_poly(pt, set_len)) % R, pt, y, z, set_len),
            _plookup_subset_eval(be[3][col], be[5][col], pt, y, zeta, num_non_zero),
            _plookup_set_eval(be[11][col], pt, y, z, set_len + num_non_zero),
            # lookup z*
            _plookup_set_eval((zval + zeta * evaluate_index_poly(pt, num_variables)) % R, pt, y, z, num_variables),
            _plookup_subset_eval(be[4][col], be[6][col], pt, y, zeta, num_non_zero),
            _plookup_set_eval(be[12][col], pt, y, z, num_variables + num_non_zero),
            # val_a, val_b, val_c, alpha*
            be[7][col], be[8][col], be[9][col], be[3][col],
        ])
    d2 = [P.ip(t, P.powers(bc, len(t))) for t in terms]
    d3 = [be[4][1], be[4][2]]
    d4 = [P.ip([be[1][c], be[2][c], be[3][c]], P.powers(bc, 3)) for c in (1, 2)]
    base_commitments = [proof["witness_commitment"]] + list(proof["r_star_commitments"]) + [proof["z_star_commitment"]] + list(index) + \
        [proof["sorted_r_commitment"], proof["sorted_alpha_commitment"], proof["sorted_z_commitment"]] + list(ep["acc_v_commitments"])
    mu_powers2 = P.powers2(mu, len(ch3))
    head = ch3[: len(ch2)]
    tensorcheck_verify(tc, tr, vk, [res1, res2, res3, res4], base_commitments, [d1, d2, d3, d4],
                       [P.hadamard(ch3, mu_powers2), list(ch3), list(ch2), P.hadamard(ch2, head)], beta, batch_consistency)

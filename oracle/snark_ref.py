"""CPU restatement of the non-preprocessing SNARK time prover (TEST INFRASTRUCTURE ONLY).

Follows src/snark/time_prover.rs:19-117, src/subprotocols/tensorcheck/mod.rs:124-133,190-275,
src/kzg/time.rs:81-159 and src/circuit.rs:349-365 on Python integers (oracle/pyref.py) with the
commitments computed by the C oracle's Pippenger (oracle/gemini_oracle.c).  Used by the parity
tests to check every element of the proof the HIP path produces, including the Fiat-Shamir
challenges in between.  Parity with a Rust run of the reference is unpinned at the byte level
(see pyref.py header); what this pins is HIP path == independent restatement, transcript included.
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc
from . import pyref as P

R = P.R_MOD


def srs(tau: int, n: int, g=None) -> np.ndarray:
    """powers_of_g[i] = tau^i * g  (src/kzg/time.rs:51-59); g = the standard generator unless an affine integer
    pair is given (the draw a reference run recorded, tools/refvectors)"""
    pw = orc.ints_to_limbs([pow(tau, i, R) for i in range(n)], 4)
    base = orc.g1_generator() if g is None else np.concatenate([orc.fq_to_mont(orc.ints_to_limbs([g[0]], 6))[0], orc.fq_to_mont(orc.ints_to_limbs([g[1]], 6))[0]])
    return orc.g1_fixed_base_mul(base, pw)


def commit(powers_of_g: np.ndarray, poly) -> tuple | None:
    """src/kzg/time.rs:81-83 -> affine integer point (or None)"""
    n = min(len(powers_of_g), len(poly))
    if n == 0:
        return None
    jac = orc.msm_pippenger(powers_of_g[:n], orc.ints_to_limbs([int(c) % R for c in poly[:n]], 4))
    return orc.affine_to_ints(orc.g1_to_affine(jac))


def open_multi_points(powers_of_g, poly, points):
    """src/kzg/time.rs:134-145"""
    q, _ = P.poly_divmod(poly, P.vanishing_polynomial(points))
    return commit(powers_of_g, q)


def batch_open_multi_points(powers_of_g, polys, points, eval_chal):
    """src/kzg/time.rs:149-159"""
    etas = P.powers(eval_chal, len(polys))
    return open_multi_points(powers_of_g, P.linear_combination(polys, etas), points)


def foldings_polynomial(poly, challenges):
    """tensorcheck/mod.rs:124-133"""
    out = []
    cur = list(poly)
    for ch in challenges[:-1]:
        cur = P.fold_polynomial(cur, ch)
        out.append(cur)
    return out


def tensorcheck_new_time(tr: P.GeminiTranscript, powers_of_g, base_polys, body_polys):
    """tensorcheck/mod.rs:190-275"""
    max_len = max(len(p) for p, _ in body_polys)
    batch_challenge = tr.get_challenge(b"batch_challenge")
    batch_challenges = P.powers(batch_challenge, max_len)
    foldings = []
    for polys, challenges in body_polys:
        batched = P.linear_combination(polys, batch_challenges)
        foldings.extend(foldings_polynomial(batched, challenges))
    commitments = [commit(powers_of_g, f) for f in foldings]
    for c in commitments:
        tr.append_message(b"commitment", P.g1_serialize_uncompressed(c))
    eval_chal = tr.get_challenge(b"evaluation-chal")
    minus = (-eval_chal) % R
    chal2 = eval_chal * eval_chal % R
    base_evals = [[P.evaluate_le(p, chal2), P.evaluate_le(p, eval_chal), P.evaluate_le(p, minus)] for p in base_polys]
    fold_evals = [[P.evaluate_le(p, eval_chal), P.evaluate_le(p, minus)] for p in foldings]
    for e3 in base_evals:
        for e in e3:
            tr.append_fr(b"eval", e)
    for e2 in fold_evals:
        for e in e2:
            tr.append_fr(b"eval", e)
    open_chal = tr.get_challenge(b"open-chal")
    proof = batch_open_multi_points(powers_of_g, list(base_polys) + foldings, [chal2, eval_chal, minus], open_chal)
    return {"folded_polynomials_commitments": commitments, "folded_polynomials_evaluations": fold_evals,
            "evaluation_proof": proof, "base_polynomials_evaluations": base_evals}


def matvec(rows, z):
    """src/misc.rs:100-110"""
    return [sum(v * z[c] for v, c in row) % R for row in rows]


def dummy_r1cs(e: int, n: int):
    """src/circuit.rs:349-365"""
    inv_e = pow(e, -1, R)
    diag = [[(inv_e, i)] for i in range(n)]
    return {"a": diag, "b": diag, "c": diag, "z": [e] * n, "w": [e] * (n - 1), "x": [e]}


def snark_new_time(r1cs, powers_of_g):
    """src/snark/time_prover.rs:19-117"""
    z = r1cs["z"]
    z_a, z_b, z_c = matvec(r1cs["a"], z), matvec(r1cs["b"], z), matvec(r1cs["c"], z)
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    witness_commitment = commit(powers_of_g, r1cs["w"])
    tr.append_message(b"witness", P.g1_serialize_uncompressed(witness_commitment))
    alpha = tr.get_challenge(b"alpha")
    zc_alpha = P.evaluate_le(z_c, alpha)
    tr.append_fr(b"zc(alpha)", zc_alpha)
    m1, ch1, ff1 = P.sumcheck_prove(tr, P.TimeProver(z_a, z_b, alpha))
    b_ch = P.tensor(ch1)
    c_ch = P.powers(alpha, len(b_ch))
    a_ch = P.hadamard(b_ch, c_ch)
    eta = tr.get_challenge(b"eta")
    eta2 = eta * eta % R
    abc = [0] * len(z)
    for i, row in enumerate(r1cs["a"]):
        for v, c in row:
            abc[c] = (abc[c] + a_ch[i] * v) % R
    for i, row in enumerate(r1cs["b"]):
        for v, c in row:
            abc[c] = (abc[c] + eta * b_ch[i] * v) % R
    for i, row in enumerate(r1cs["c"]):
        for v, c in row:
            abc[c] = (abc[c] + eta2 * c_ch[i] * v) % R
    m2, ch2, ff2 = P.sumcheck_prove(tr, P.TimeProver(abc, z, 1))
    tc = tensorcheck_new_time(tr, powers_of_g, [r1cs["w"]], [([abc, z], ch2)])
    return {"witness_commitment": witness_commitment, "zc_alpha": zc_alpha, "first_sumcheck_msgs": (m1, ff1),
            "second_sumcheck_msgs": (m2, ff2), "tensorcheck_proof": tc,
            "challenges": {"alpha": alpha, "first": ch1, "eta": eta, "second": ch2}}

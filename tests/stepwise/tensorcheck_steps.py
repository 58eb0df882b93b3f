"""TEST INFRASTRUCTURE: TensorcheckProof::new_time (src/subprotocols/tensorcheck/mod.rs:190-275) step by step over the Python mirror of the
device primitives -- part of the step-wise cross-check of the native provers (tests/stepwise/__init__.py)."""
from __future__ import annotations

import numpy as np

from gemini_amd.fr import FrVec, evaluate_le, evaluate_le_batch, fr_from_int, fr_to_int, linear_combination, powers, R_MOD
from gemini_amd.tensorcheck import TensorcheckProof, foldings_polynomial


def tensorcheck_new_time(transcript, ck, base_polynomials, body_polynomials) -> TensorcheckProof:
    """:190-275.  body_polynomials: [(polynomials, challenges)]"""
    max_len = max((len(p) for p, _ in body_polynomials), default=0)
    batch_challenge = transcript.get_challenge(b"batch_challenge")
    batch_challenges = powers(batch_challenge, max_len)
    assert max_len != 0 and all(len(p) != 0 for p, _ in body_polynomials)
    bc_host = batch_challenges.to_host()
    batch_challenges.free()
    foldings = []
    batched_list = []
    for polys, challenges in body_polynomials:
        batched = linear_combination(polys, bc_host)
        batched_list.append(batched)
        foldings.extend(foldings_polynomial(batched, challenges))
    commitments = ck.batch_commit(foldings)
    for c in commitments:
        transcript.append_g1(b"commitment", c)
    eval_chal = transcript.get_challenge(b"evaluation-chal")
    ec = fr_to_int(eval_chal)
    minus_eval_chal = fr_from_int((-ec) % R_MOD)
    eval_chal2 = fr_from_int(ec * ec % R_MOD)
    pts3 = np.stack([eval_chal2, eval_chal, minus_eval_chal])
    # :228-247, one wait per group instead of one per polynomial (22 base polynomials + ~90 foldings in the preprocessing prover)
    base_evals = list(evaluate_le_batch(list(base_polynomials), pts3)) if all(isinstance(p, FrVec) for p in base_polynomials) \
        else [evaluate_le(p, pts3) for p in base_polynomials]
    fold_evals = list(evaluate_le_batch(foldings, pts3[1:]))
    for e3 in base_evals:
        for e in e3:
            transcript.append_fr(b"eval", e)
    for e2 in fold_evals:
        for e in e2:
            transcript.append_fr(b"eval", e)
    open_chal = transcript.get_challenge(b"open-chal")
    all_polys = list(base_polynomials) + foldings
    evaluation_proof = ck.batch_open_multi_points(all_polys, pts3, open_chal)
    for v in foldings + batched_list:
        v.free()
    return TensorcheckProof(commitments, fold_evals, evaluation_proof, base_evals)

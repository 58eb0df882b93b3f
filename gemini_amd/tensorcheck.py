"""TensorcheckProof (src/subprotocols/tensorcheck/mod.rs:39-46) as data, and foldings_polynomial (:124-133).  The prover (:190-275) runs inside the
native provers (gemini_amd/csrc/{snark,psnark}.cpp); its step-wise statement is tests/stepwise/tensorcheck_steps.py."""
from __future__ import annotations

from .fr import FrVec, fold_polynomial


def foldings_polynomial(polynomial: FrVec, challenges_mont) -> list:
    """:124-133  successive folds with all challenges but the last (strip_last)"""
    out = []
    cur = polynomial
    for ch in list(challenges_mont)[:-1]:
        cur = fold_polynomial(cur, ch)
        out.append(cur)
    return out


class TensorcheckProof:
    def __init__(self, folded_polynomials_commitments, folded_polynomials_evaluations, evaluation_proof, base_polynomials_evaluations):
        self.folded_polynomials_commitments = folded_polynomials_commitments
        self.folded_polynomials_evaluations = folded_polynomials_evaluations
        self.evaluation_proof = evaluation_proof
        self.base_polynomials_evaluations = base_polynomials_evaluations

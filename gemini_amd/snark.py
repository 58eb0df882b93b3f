"""Host mirror of the non-preprocessing SNARK time prover, src/snark/time_prover.rs:19-117: pure
orchestration -- every O(n) step is a device call (MSM, sumcheck rounds, vector passes, SpMV)."""
from __future__ import annotations

import time

import numpy as np

from .circuit import R1cs
from .fr import FrVec, evaluate_le, fr_from_int, fr_to_int, hadamard, linear_combination, powers, tensor, R_MOD
from .kzg import CommitterKey
from .sumcheck import Sumcheck
from .tensorcheck import TensorcheckProof
from .transcript import Transcript, PROTOCOL_NAME


class Proof:
    """src/snark/mod.rs:76-82"""

    def __init__(self, witness_commitment, zc_alpha, first_sumcheck_msgs, second_sumcheck_msgs, tensorcheck_proof):
        self.witness_commitment = witness_commitment
        self.zc_alpha = zc_alpha
        self.first_sumcheck_msgs = first_sumcheck_msgs
        self.second_sumcheck_msgs = second_sumcheck_msgs
        self.tensorcheck_proof = tensorcheck_proof
        self.spans = {}

    @staticmethod
    def new_time(r1cs: R1cs, ck: CommitterKey) -> "Proof":
        spans = {}
        t_all = time.perf_counter()
        z_a = r1cs.a.mul(r1cs.z)  # :32-34
        z_b = r1cs.b.mul(r1cs.z)
        z_c = r1cs.c.mul(r1cs.z)
        transcript = Transcript(PROTOCOL_NAME)
        spans["product_matrix_vector x3"] = time.perf_counter() - t_all

        t0 = time.perf_counter()
        witness_commitment = ck.commit(r1cs.w)  # :42
        spans["Commitment to w"] = time.perf_counter() - t0
        transcript.append_g1(b"witness", witness_commitment)
        alpha = transcript.get_challenge(b"alpha")
        zc_alpha = evaluate_le(z_c, alpha.reshape(1, 4))[0]  # :48
        transcript.append_fr(b"zc(alpha)", zc_alpha)

        t0 = time.perf_counter()
        first_proof = Sumcheck.new_time(transcript, z_a, z_b, alpha)  # :52
        spans["First sumcheck"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        b_challenges = tensor(np.stack(first_proof.challenges))  # :56-58
        c_challenges = powers(alpha, len(b_challenges))
        a_challenges = hadamard(b_challenges, c_challenges)
        eta = transcript.get_challenge(b"eta")
        eta_i = fr_to_int(eta)
        eta2 = fr_from_int(eta_i * eta_i % R_MOD)

        # abc_tensored[col] = sum_rows rA[i] A[i,col] + eta rB[i] B[i,col] + eta^2 rC[i] C[i,col]   :63-81
        ta = r1cs.at.mul(a_challenges)
        tb = r1cs.bt.mul(b_challenges)
        tc = r1cs.ct.mul(c_challenges)
        abc_tensored = linear_combination([ta, tb, tc], np.stack([fr_from_int(1), eta, eta2]))
        # the reference allocates vec![0; z.len()] (no trimming): restore the full logical length, the
        # trimmed tail is already zero on the device
        abc_tensored.set_len(len(r1cs.z))
        for v in (ta, tb, tc, a_challenges, b_challenges, c_challenges):
            v.free()
        spans["tensor/powers/hadamard/abc_tensored"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        second_proof = Sumcheck.new_time(transcript, abc_tensored, r1cs.z, fr_from_int(1))  # :84-89
        spans["Second sumcheck"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        tensorcheck_proof = TensorcheckProof.new_time(  # :101-106
            transcript, ck, [r1cs.w], [([abc_tensored, r1cs.z], second_proof.challenges)]
        )
        spans["Tensorcheck"] = time.perf_counter() - t0
        for v in (z_a, z_b, z_c, abc_tensored):
            v.free()
        transcript.free()
        spans["ark_gemini::snark::time_prover"] = time.perf_counter() - t_all
        proof = Proof(witness_commitment, zc_alpha, (first_proof.messages, first_proof.final_foldings),
                      (second_proof.messages, second_proof.final_foldings), tensorcheck_proof)
        proof.spans = spans
        return proof

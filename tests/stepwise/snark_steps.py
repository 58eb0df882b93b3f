"""TEST INFRASTRUCTURE: the STEP-WISE orchestration of snark::Proof::{new_time, new_elastic} (src/snark/time_prover.rs:19-117,
src/snark/elastic_prover.rs:105-266) over the Python mirror of the device primitives -- one FFI call per step.  The product's provers are the
ones compiled into the library (gm_snark_new_time / gm_snark_new_elastic, gemini_amd/csrc/snark.cpp); this second statement of the same
sequence is the cross-check the tests hold them to (byte for byte), and what runs on key types the native entry does not take (the Python-level
sharded keys of tests/stepwise/dist.py).  Registered with gemini_amd.snark by tests/stepwise/__init__.py."""
from __future__ import annotations

import time

import numpy as np

from gemini_amd.fr import FrVec, evaluate_le, evaluate_le_batch, fr_from_int, fr_to_int, hadamard, linear_combination, powers, reverse, tensor, R_MOD
from gemini_amd.snark import Proof
from gemini_amd.sumcheck import Sumcheck
from gemini_amd.tensorcheck import TensorcheckProof
from gemini_amd.transcript import Transcript, PROTOCOL_NAME
from tests.stepwise.tensorcheck_steps import tensorcheck_new_time


def new_time(r1cs, ck) -> Proof:
    """src/snark/time_prover.rs:19-117"""
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
    tensorcheck_proof = tensorcheck_new_time(  # :101-106
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


def _evaluate_be(stream: FrVec, xs) -> np.ndarray:
    """evaluate_be over a big-endian stream (src/misc.rs:180-190) = evaluate_le of the reversed vector"""
    le = reverse(stream)
    try:
        return evaluate_le(le, xs)
    finally:
        le.free()


def elastic_tensorcheck(transcript, ck, base_polynomial: FrVec, body_stream: FrVec, challenges, max_msm_buffer: int) -> TensorcheckProof:
    """src/snark/elastic_prover.rs:105-168 (`tensorcheck`): commit_folding, evaluate_folding at +-beta,
    open_multi_points(w) + open_folding(foldings)"""
    from gemini_amd.kzg import FoldedPolynomialTree
    from gemini_amd.msm import g1_sum

    tc_challenges = list(challenges)[:-1]  # strip_last
    tree = FoldedPolynomialTree(body_stream, tc_challenges)
    # The reference re-streams the folded polynomial tree for the commitments, the evaluations and the opening
    # (O(log n) memory); with the streams resident in HBM the levels (n / 2 + n / 4 + ... elements) are folded ONCE.
    levels = ck._foldings_le(tree)
    commitments = ck.commit_folding(tree, max_msm_buffer, levels=levels)
    for c in commitments:
        transcript.append_g1(b"commitment", c)
    eval_chal = transcript.get_challenge(b"evaluation-chal")
    ec = fr_to_int(eval_chal)
    pts = np.stack([fr_from_int(ec * ec % R_MOD), eval_chal, fr_from_int((-ec) % R_MOD)])
    # evaluate_folding (tensorcheck/mod.rs:73-88): f^(j)(x) for every folding level, one wait for all of them
    fold_evals = list(evaluate_le_batch(levels, pts[1:]))
    evaluations_w = _evaluate_be(base_polynomial, pts)
    for e in evaluations_w:
        transcript.append_fr(b"eval", e)
    for e2 in fold_evals:
        for e in e2:
            transcript.append_fr(b"eval", e)
    open_chal = transcript.get_challenge(b"open-chal")
    open_chals = powers(open_chal, len(challenges) + 1)
    oc = open_chals.to_host()
    open_chals.free()
    _, proof_w = ck.open_multi_points(base_polynomial, pts, max_msm_buffer)
    _, proof = ck.open_folding(tree, pts, oc[1:], max_msm_buffer, levels=levels)  # frees the levels
    evaluation_proof = g1_sum(np.stack([proof_w, proof]))
    return TensorcheckProof(commitments, fold_evals, evaluation_proof, [evaluations_w])


def new_elastic(r1cs_stream, ck_stream, max_msm_buffer: int) -> Proof:
    """src/snark/elastic_prover.rs:174-266 over device-resident streams"""
    spans = {}
    t_all = time.perf_counter()
    transcript = Transcript(PROTOCOL_NAME)
    t0 = time.perf_counter()
    witness_commitment = ck_stream.commit(r1cs_stream.witness)  # :209
    spans["Commitment to w"] = time.perf_counter() - t0
    transcript.append_g1(b"witness", witness_commitment)
    alpha = transcript.get_challenge(b"alpha")
    zc_alpha = _evaluate_be(r1cs_stream.z_c, alpha.reshape(1, 4))[0]  # :216
    transcript.append_fr(b"zc(alpha)", zc_alpha)
    t0 = time.perf_counter()
    first_proof = Sumcheck.new_elastic(transcript, r1cs_stream.z_a, r1cs_stream.z_b, alpha)  # :222
    spans["First sumcheck"] = time.perf_counter() - t0
    eta = transcript.get_challenge(b"eta")
    eta_i = fr_to_int(eta)
    # MatrixTensor streams (:233-238): A^T tensor(a_tensors) etc.; tensor(powers2(alpha)) = powers(alpha)
    b_challenges = tensor(np.stack(first_proof.challenges))
    c_challenges = powers(alpha, len(b_challenges))
    a_challenges = hadamard(b_challenges, c_challenges)
    ta, tb, tc = r1cs_stream.at.mul(a_challenges), r1cs_stream.bt.mul(b_challenges), r1cs_stream.ct.mul(c_challenges)
    lhs_le = linear_combination([ta, tb, tc], np.stack([fr_from_int(1), eta, fr_from_int(eta_i * eta_i % R_MOD)]))
    lhs_le.set_len(len(r1cs_stream.z))
    lhs = reverse(lhs_le)
    for v in (ta, tb, tc, a_challenges, b_challenges, c_challenges):
        v.free()
    t0 = time.perf_counter()
    second_proof = Sumcheck.new_elastic(transcript, lhs, r1cs_stream.z, fr_from_int(1))  # :241
    spans["Second sumcheck"] = time.perf_counter() - t0
    batch_challenge = transcript.get_challenge(b"batch_challenge")
    t0 = time.perf_counter()
    z_le = reverse(r1cs_stream.z)
    body_le = linear_combination([lhs_le, z_le], np.stack([fr_from_int(1), batch_challenge]))
    body = reverse(body_le)
    tensorcheck_proof = elastic_tensorcheck(transcript, ck_stream, r1cs_stream.witness, body, second_proof.challenges, max_msm_buffer)
    spans["Tensorcheck"] = time.perf_counter() - t0
    for v in (lhs_le, lhs, z_le, body_le, body):
        v.free()
    transcript.free()
    spans["ark_gemini::snark::elastic_prover"] = time.perf_counter() - t_all
    proof = Proof(witness_commitment, zc_alpha, (first_proof.messages, first_proof.final_foldings),
                  (second_proof.messages, second_proof.final_foldings), tensorcheck_proof)
    proof.spans = spans
    return proof



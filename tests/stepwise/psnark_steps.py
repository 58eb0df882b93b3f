"""TEST INFRASTRUCTURE: the STEP-WISE orchestration of psnark::Proof::{new_time, new_elastic} (src/psnark/time_prover.rs:69-384,
src/psnark/elastic_prover.rs:60-634) and of the entry-product argument (src/subprotocols/entryproduct/time_prover.rs:53-114) over the Python
mirror of the device primitives.  The product's provers are the ones compiled into the library (gm_psnark_new_time / gm_psnark_new_elastic /
gm_psnark_new_time_sharded); this second statement of the same sequences is the cross-check the tests hold them to, byte for byte.
Registered with gemini_amd.psnark by tests/stepwise/__init__.py."""
from __future__ import annotations

import time

import numpy as np

from gemini_amd.fr import (FrVec, IdxVec, R_MOD, accumulated_product_monic, alg_hash, element, evaluate_le, evaluate_le_batch, fr_from_int, fr_to_int, hadamard, ip,
                           linear_combination, lookup, plookup_set, plookup_subset, powers, shift_monic, tensor)
from gemini_amd.psnark import EntryProductMsgs, Proof, _joint_device, plookup
from gemini_amd.sumcheck import Sumcheck, TimeProver
from gemini_amd.transcript import PROTOCOL_NAME, Transcript
from tests.stepwise.snark_steps import _evaluate_be
from gemini_amd.tensorcheck import TensorcheckProof
from tests.stepwise.tensorcheck_steps import tensorcheck_new_time

_ONE = fr_from_int(1)


class EntryProduct:
    """entryproduct/mod.rs:27-31"""

    def __init__(self, msgs, chal, provers):
        self.msgs, self.chal, self.provers = msgs, chal, provers

    @staticmethod
    def new_time_batch(transcript, ck, vs, claimed_products, acc_vs=None) -> "EntryProduct":
        """entryproduct/time_prover.rs:53-114.  acc_vs: accumulated_product(monic(v)) when the caller has
        them already (the psnark prover does)."""
        assert len(vs) == len(claimed_products)
        own = acc_vs is None
        if own:
            acc_vs = [accumulated_product_monic(v) for v in vs]
        rrot_vs = [shift_monic(v) for v in vs]
        acc_v_commitments = ck.batch_commit(acc_vs)
        for c in acc_v_commitments:
            transcript.append_g1(b"acc_v", c)
        chal = transcript.get_challenge(b"ep-chal")
        ci = fr_to_int(chal)
        provers = [TimeProver(acc_v, rrot_v, chal) for rrot_v, acc_v in zip(rrot_vs, acc_vs)]
        claimed_sumchecks = []
        acc_v_chals = evaluate_le_batch(list(acc_vs), chal.reshape(1, 4))
        for cp, acc_v, av in zip(claimed_products, acc_vs, acc_v_chals):
            acc_v_chal = fr_to_int(av[0])
            chal_n = pow(ci, len(acc_v), R_MOD)
            claimed_sumchecks.append(fr_from_int((acc_v_chal * ci + fr_to_int(cp) - chal_n) % R_MOD))
        for v in rrot_vs + (acc_vs if own else []):
            v.free()
        return EntryProduct(EntryProductMsgs(acc_v_commitments, claimed_sumchecks), chal, provers)

    @staticmethod
    def new_time(transcript, ck, v, claimed_product) -> "EntryProduct":
        """entryproduct/time_prover.rs:116-147"""
        return EntryProduct.new_time_batch(transcript, ck, [v], [claimed_product])

    @staticmethod
    def new_elastic(transcript, ck_stream, v_stream, claimed_product) -> "EntryProduct":
        """entryproduct/elastic_prover.rs:32-63 over a big-endian device stream: ProductStream /
        RightRotationStreamer are the reversed accumulated-product / shifted vectors"""
        from gemini_amd.fr import reverse
        from gemini_amd.sumcheck import ElasticProver

        v = reverse(v_stream)
        acc_s, rrot_s = reverse(accumulated_product_monic(v)), reverse(shift_monic(v))
        cm = ck_stream.commit(acc_s)
        transcript.append_g1(b"acc_v", cm)
        chal = transcript.get_challenge(b"ep-chal")
        ci = fr_to_int(chal)
        claimed = fr_from_int((ci * fr_to_int(_evaluate_be(acc_s, chal.reshape(1, 4))[0]) + fr_to_int(claimed_product) - pow(ci, len(acc_s), R_MOD)) % R_MOD)
        provers = [ElasticProver(acc_s, rrot_s, chal)]
        for x in (v, acc_s, rrot_s):
            x.free()
        return EntryProduct(EntryProductMsgs([cm], [claimed]), chal, provers)



def new_time(ck, r1cs, index: list) -> Proof:
    """src/psnark/time_prover.rs:69-384"""
    spans = {}
    keep = []  # device vectors freed at the end

    def K(v):
        keep.append(v)
        return v

    t_all = time.perf_counter()
    z_a = K(r1cs.a.mul(r1cs.z))  # :74-76
    z_b = K(r1cs.b.mul(r1cs.z))
    z_c = K(r1cs.c.mul(r1cs.z))
    transcript = Transcript(PROTOCOL_NAME)
    t0 = time.perf_counter()
    witness_commitment = ck.commit(r1cs.w)  # :79
    spans["Commitment to w"] = time.perf_counter() - t0

    transcript.append_g1(b"witness", witness_commitment)  # :82-86
    transcript.append_message(b"ck", ck.powers_of_g2_bytes())
    transcript.append_g1(b"instance", np.stack(index), with_len=True)
    alpha = transcript.get_challenge(b"alpha")

    zc_alpha = evaluate_le(z_c, alpha.reshape(1, 4))[0]  # :88-89
    transcript.append_fr(b"zc(alpha)", zc_alpha)

    t0 = time.perf_counter()
    first_proof = Sumcheck.new_time(transcript, z_a, z_b, alpha)  # :92
    spans["First sumcheck"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    b_challenges = K(tensor(np.stack(first_proof.challenges)))  # :95-97
    c_challenges = K(powers(alpha, len(b_challenges)))
    a_challenges = K(hadamard(b_challenges, c_challenges))

    jd = _joint_device(r1cs)  # :99-110, resident with the instance
    row_index, col_index, row, col = jd.row_index, jd.col_index, jd.row, jd.col
    val_a, val_b, val_c = jd.val_a, jd.val_b, jd.val_c
    num_non_zero = len(row_index)
    spans["joint matrices"] = time.perf_counter() - t0

    ralpha_star = K(lookup(a_challenges, row_index))  # :114-117
    r_star = K(lookup(b_challenges, row_index))
    alpha_star = K(lookup(c_challenges, row_index))
    z_star = K(lookup(r1cs.z, col_index))

    # :119-127.  ck.index_by(row_index).commit(a_challenges) = sum_j a_challenges[row_index[j]] * g_j, the
    # commitment to the looked-up vector under ck itself (the reference's commented-out line :126): one
    # MSM over the resident key instead of building an indexed key.  The index_by zip needs as many
    # powers as indices.
    assert ck.num_powers() >= num_non_zero, "committer key shorter than the number of non-zero entries"
    t0 = time.perf_counter()
    z_r_commitments = ck.batch_commit([ralpha_star, r_star, alpha_star]) + [ck.commit(z_star)]
    spans["Commitments to z* and r*"] = time.perf_counter() - t0

    transcript.append_g1(b"ra*", z_r_commitments[0])  # :129-132
    transcript.append_g1(b"rb*", z_r_commitments[1])
    transcript.append_g1(b"rc*", z_r_commitments[2])
    transcript.append_g1(b"z*", z_r_commitments[3])

    eta = transcript.get_challenge(b"chal")  # :134-135
    eta_i = fr_to_int(eta)
    challenges = np.stack([_ONE, eta, fr_from_int(eta_i * eta_i % R_MOD)])

    h_a, h_b, h_c = hadamard(ralpha_star, val_a), hadamard(r_star, val_b), hadamard(alpha_star, val_c)
    r_star_val = K(linear_combination([h_a, h_b, h_c], challenges))  # :137-144
    for v in (h_a, h_b, h_c):
        v.free()

    t0 = time.perf_counter()
    second_proof = Sumcheck.new_time(transcript, z_star, r_star_val, _ONE)  # :147-152
    second_challenges = K(tensor(np.stack(second_proof.challenges)))
    assert len(second_challenges) >= num_non_zero
    second_challenges_head = second_challenges
    second_challenges_head.set_len(num_non_zero)  # &second_challenges[..num_non_zero]
    spans["Second sumcheck"] = time.perf_counter() - t0

    zeta = transcript.get_challenge(b"zeta")  # :157

    t0 = time.perf_counter()
    alg_hash_poly = [K(alg_hash(b_challenges, None, zeta)), K(alg_hash(c_challenges, None, zeta)), K(alg_hash(r1cs.z, None, zeta))]  # :160-164
    ext_fre = jd.extended_frequencies(len(alg_hash_poly[0]), len(alg_hash_poly[2]))  # :165-168, :175-178
    sorted_polynomials = [K(lookup(alg_hash_poly[0], ext_fre[0])), K(lookup(alg_hash_poly[1], ext_fre[0])),
                          K(lookup(alg_hash_poly[2], ext_fre[1]))]  # :169-173
    # :179-183: ck.index_by(ext_fre).commit(alg_hash_poly) = commitment to the sorted vector under ck (:183)
    assert ck.num_powers() >= max(len(ext_fre[0]), len(ext_fre[1])), "committer key shorter than the sorted vectors"
    sorted_commitments = ck.batch_commit(sorted_polynomials)
    spans["Commitments to sorted vectors"] = time.perf_counter() - t0

    transcript.append_g1(b"sorted_alpha_commitment", sorted_commitments[1])  # :186-188
    transcript.append_g1(b"sorted_r_commitment", sorted_commitments[0])
    transcript.append_g1(b"sorted_z_commitment", sorted_commitments[2])

    gamma = transcript.get_challenge(b"gamma")  # :190-191
    chi = transcript.get_challenge(b"chi")

    t0 = time.perf_counter()
    r_lookup_vec = [K(v) for v in plookup(r_star, b_challenges, row_index, ext_fre[0], gamma, chi, zeta)]  # :194-204
    alpha_lookup_vec = [K(v) for v in plookup(alpha_star, c_challenges, row_index, ext_fre[0], gamma, chi, zeta)]
    z_lookup_vec = [K(v) for v in plookup(z_star, r1cs.z, col_index, ext_fre[1], gamma, chi, zeta)]
    lookup_vec = r_lookup_vec + alpha_lookup_vec + z_lookup_vec  # :206-209
    accumulated_vec = [K(accumulated_product_monic(v)) for v in lookup_vec]  # accproduct3, :211-214
    prod = [element(acc, 0) for acc in accumulated_vec]  # product3: the full product is the first accumulated entry
    r_prod_vec, alpha_prod_vec, z_prod_vec = prod[0:3], prod[3:6], prod[6:9]
    spans["plookup vectors + accumulated products"] = time.perf_counter() - t0

    transcript.append_fr(b"set_r_ep", alpha_prod_vec[0])  # :216-221 (labels as in the reference)
    transcript.append_fr(b"subset_r_ep", alpha_prod_vec[1])
    transcript.append_fr(b"set_r_ep", r_prod_vec[0])
    transcript.append_fr(b"subset_r_ep", r_prod_vec[1])
    transcript.append_fr(b"set_z_ep", z_prod_vec[0])
    transcript.append_fr(b"subset_z_ep", z_prod_vec[1])

    t0 = time.perf_counter()
    entry_products = EntryProduct.new_time_batch(transcript, ck, lookup_vec, prod, acc_vs=accumulated_vec)  # :223-239
    spans["Entry products"] = time.perf_counter() - t0

    psi = entry_products.chal  # :241-242
    open_chal = transcript.get_challenge(b"open-chal")

    t0 = time.perf_counter()
    polynomials = [ralpha_star] + accumulated_vec  # :244-251
    ralpha_star_acc_mu_proof = ck.batch_open_multi_points(polynomials, psi.reshape(1, 4), open_chal)
    ralpha_star_acc_mu_evals = [e[0] for e in evaluate_le_batch(polynomials, psi.reshape(1, 4))]
    spans["Opening at psi"] = time.perf_counter() - t0

    h_a, h_b = hadamard(ralpha_star, val_a), hadamard(r_star, val_b)  # :253-254
    s_0_prime, s_1_prime = ip(h_a, second_challenges_head), ip(h_b, second_challenges_head)
    h_a.free()
    h_b.free()
    for e in ralpha_star_acc_mu_evals:  # :258-261
        transcript.append_fr(b"ralpha_star_acc_mu", e)
    transcript.append_g1(b"ralpha_star_mu_proof", ralpha_star_acc_mu_proof)

    provers = list(entry_products.provers)  # :263-290
    for lhs, rhs in ((ralpha_star, val_a), (r_star, val_b), (alpha_star, val_c)):
        h = hadamard(lhs, second_challenges_head)
        provers.append(TimeProver(h, rhs, _ONE))
        h.free()
    provers.append(TimeProver(r_star, alpha_star, psi))

    t0 = time.perf_counter()
    third_proof = Sumcheck.prove_batch(transcript, provers)  # :293
    for p in provers:
        p.free()
    spans["Third sumcheck"] = time.perf_counter() - t0

    tc_base_polynomials = [r1cs.w, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c] + sorted_polynomials + accumulated_vec  # :296-319

    third_ch = [fr_to_int(c) for c in third_proof.challenges]
    second_ch = [fr_to_int(c) for c in second_proof.challenges]
    twist_powers2 = [pow(fr_to_int(psi), 1 << j, R_MOD) for j in range(len(third_ch))]  # :321

    shift_monic_lookup_vec = [K(shift_monic(v)) for v in lookup_vec]  # :323-326
    third_proof_vec = shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star]  # :329-330
    body_polynomials_0 = accumulated_vec + [r_star]  # :334-345
    head = third_ch[: len(second_ch)]  # :346
    F = lambda ints: [fr_from_int(v) for v in ints]
    tc_body_polynomials = [  # :347-359
        (body_polynomials_0, F([a * b % R_MOD for a, b in zip(third_ch, twist_powers2)])),
        (third_proof_vec, F(third_ch)),
        ([z_star], F(second_ch)),
        ([ralpha_star, r_star, alpha_star], F([a * b % R_MOD for a, b in zip(second_ch, head)])),
    ]

    t0 = time.perf_counter()
    tensorcheck_proof = tensorcheck_new_time(transcript, ck, tc_base_polynomials, tc_body_polynomials)  # :362-367
    spans["Tensorcheck"] = time.perf_counter() - t0

    for v in keep:
        v.free()
    transcript.free()
    spans["ark_gemini::psnark::time_prover"] = time.perf_counter() - t_all
    proof = Proof(
        witness_commitment=witness_commitment, zc_alpha=zc_alpha,
        first_sumcheck_msgs=(first_proof.messages, first_proof.final_foldings),
        r_star_commitments=z_r_commitments[:3], z_star_commitment=z_r_commitments[3],
        second_sumcheck_msgs=(second_proof.messages, second_proof.final_foldings),
        set_r_ep=r_prod_vec[0], subset_r_ep=r_prod_vec[1], sorted_r_commitment=sorted_commitments[0],
        set_alpha_ep=alpha_prod_vec[0], subset_alpha_ep=alpha_prod_vec[1], sorted_alpha_commitment=sorted_commitments[1],
        set_z_ep=z_prod_vec[0], subset_z_ep=z_prod_vec[1], sorted_z_commitment=sorted_commitments[2],
        ep_msgs=entry_products.msgs, ralpha_star_acc_mu_evals=ralpha_star_acc_mu_evals,
        ralpha_star_acc_mu_proof=ralpha_star_acc_mu_proof, rstars_vals=[s_0_prime, s_1_prime],
        third_sumcheck_msgs=(third_proof.messages, third_proof.final_foldings), tensorcheck_proof=tensorcheck_proof)
    proof.spans = spans
    return proof


def new_elastic(ck, r1cs_stream, index: list, max_msm_buffer: int) -> Proof:
    """src/psnark/elastic_prover.rs:60-634 over device-resident streams: `ck` is a CommitterKeyStream, every polynomial a big-endian stream (reversed
    device vector); commitments are chunked stream MSMs, sumchecks run on the space / elastic provers, the tensor check on FoldedPolynomialTrees"""
    from gemini_amd.fr import fold_polynomial, reverse
    from gemini_amd.kzg import FoldedPolynomialTree
    from gemini_amd.msm import g1_sum
    from gemini_amd.sumcheck import ElasticProver

    spans = {}
    keep = []

    def K(v):
        keep.append(v)
        return v

    S = lambda v: K(reverse(v))  # little-endian vector -> big-endian stream
    t_all = time.perf_counter()
    r1cs = r1cs_stream.r1cs
    transcript = Transcript(PROTOCOL_NAME)
    witness_commitment = ck.commit(r1cs_stream.witness)  # :82
    transcript.append_g1(b"witness", witness_commitment)  # :86-89
    transcript.append_message(b"ck", ck.powers_of_g2_bytes())
    transcript.append_g1(b"instance", np.stack(index), with_len=True)
    alpha = transcript.get_challenge(b"alpha")
    zc_alpha = _evaluate_be(r1cs_stream.z_c, alpha.reshape(1, 4))[0]  # :92-93
    transcript.append_fr(b"zc(alpha)", zc_alpha)
    t0 = time.perf_counter()
    sumcheck1 = Sumcheck.new_space(transcript, r1cs_stream.z_a, r1cs_stream.z_b, alpha)  # :97
    spans["sumcheck1"] = time.perf_counter() - t0

    # the Joint{Row,Col,Val} streams (:100-146) walk the joint support; here its index / value vectors
    jd = _joint_device(r1cs)
    row_index, col_index, row, col = jd.row_index, jd.col_index, jd.row, jd.col
    val_a, val_b, val_c = jd.val_a, jd.val_b, jd.val_c
    num_non_zero = len(row_index)
    z_le = K(reverse(r1cs_stream.z))
    w_le = K(reverse(r1cs_stream.witness))
    z_star = K(lookup(z_le, col_index))  # :148
    rs = K(tensor(np.stack(sumcheck1.challenges)))  # Tensor(r_short)                         :150-157
    alphas = K(powers(alpha, len(rs)))  # Tensor(powers2(alpha)) = powers of alpha
    ralphas = K(hadamard(rs, alphas))
    ralpha_star, r_star, alpha_star = K(lookup(ralphas, row_index)), K(lookup(rs, row_index)), K(lookup(alphas, row_index))  # :159-161

    t0 = time.perf_counter()
    r_star_commitments = [ck.commit(S(ralpha_star)), ck.commit(S(r_star)), ck.commit(S(alpha_star))]  # :164-172
    z_star_commitment = ck.commit(S(z_star))
    spans["Commitments to z* and r*"] = time.perf_counter() - t0
    transcript.append_g1(b"ra*", r_star_commitments[0])
    transcript.append_g1(b"rb*", r_star_commitments[1])
    transcript.append_g1(b"rc*", r_star_commitments[2])
    transcript.append_g1(b"z*", z_star_commitment)

    challenge = transcript.get_challenge(b"chal")  # :181-192
    ci = fr_to_int(challenge)
    h_a, h_b, h_c = hadamard(ralpha_star, val_a), hadamard(r_star, val_b), hadamard(alpha_star, val_c)
    rhs = K(linear_combination([h_a, h_b, h_c], np.stack([_ONE, challenge, fr_from_int(ci * ci % R_MOD)])))
    for v in (h_a, h_b, h_c):
        v.free()
    t0 = time.perf_counter()
    sumcheck2 = Sumcheck.new_elastic(transcript, S(z_star), S(rhs), _ONE)  # :195
    spans["sumcheck2"] = time.perf_counter() - t0

    zeta = transcript.get_challenge(b"zeta")  # :199
    hashed_r, hashed_alpha, hashed_z = K(alg_hash(rs, None, zeta)), K(alg_hash(alphas, None, zeta)), K(alg_hash(z_le, None, zeta))  # :205-210
    ext_fre = jd.extended_frequencies(len(rs), len(z_le))
    sorted_r, sorted_alpha, sorted_z = K(lookup(hashed_r, ext_fre[0])), K(lookup(hashed_alpha, ext_fre[0])), K(lookup(hashed_z, ext_fre[1]))  # :212-214
    t0 = time.perf_counter()
    sorted_r_commitment, sorted_alpha_commitment, sorted_z_commitment = ck.commit(S(sorted_r)), ck.commit(S(sorted_alpha)), ck.commit(S(sorted_z))
    spans["Commitments to sorted vectors"] = time.perf_counter() - t0
    transcript.append_g1(b"sorted_alpha_commitment", sorted_alpha_commitment)  # :220-222
    transcript.append_g1(b"sorted_r_commitment", sorted_r_commitment)
    transcript.append_g1(b"sorted_z_commitment", sorted_z_commitment)
    gamma = transcript.get_challenge(b"gamma")
    chi = transcript.get_challenge(b"chi")

    pl_r = [K(v) for v in plookup(r_star, rs, row_index, ext_fre[0], gamma, chi, zeta)]  # plookup_streams, :227-232
    pl_alpha = [K(v) for v in plookup(alpha_star, alphas, row_index, ext_fre[0], gamma, chi, zeta)]
    pl_z = [K(v) for v in plookup(z_star, z_le, col_index, ext_fre[1], gamma, chi, zeta)]
    pls = pl_r + pl_alpha + pl_z
    accs = [K(accumulated_product_monic(v)) for v in pls]  # ProductStream
    shifts = [K(shift_monic(v)) for v in pls]  # RightRotationStreamer
    prod = [element(a, 0) for a in accs]  # :235-243
    transcript.append_fr(b"set_r_ep", prod[3])  # :245-250
    transcript.append_fr(b"subset_r_ep", prod[4])
    transcript.append_fr(b"set_r_ep", prod[0])
    transcript.append_fr(b"subset_r_ep", prod[1])
    transcript.append_fr(b"set_z_ep", prod[6])
    transcript.append_fr(b"subset_z_ep", prod[7])

    assert len(K(tensor(np.stack(sumcheck2.challenges)))) >= num_non_zero
    ep_r = keep[-1]  # Tensor(&sumcheck2.challenges), cut to the looked-up length             :254-257
    ep_r.set_len(num_non_zero)

    # EntryProduct::new_elastic_batch (entryproduct/elastic_prover.rs:66-127)
    t0 = time.perf_counter()
    acc_streams = [S(a) for a in accs]
    acc_v_commitments = []
    for a in acc_streams:
        cm = ck.commit(a)
        transcript.append_g1(b"acc_v", cm)
        acc_v_commitments.append(cm)
    psi = transcript.get_challenge(b"ep-chal")
    pi = fr_to_int(psi)
    claimed_sumchecks, provers = [], []
    for cp, a, a_s, sh in zip(prod, accs, acc_streams, shifts):
        acc_v_chal = fr_to_int(_evaluate_be(a_s, psi.reshape(1, 4))[0])
        claimed_sumchecks.append(fr_from_int((acc_v_chal * pi + fr_to_int(cp) - pow(pi, len(a), R_MOD)) % R_MOD))
        provers.append(ElasticProver(a_s, S(sh), psi))
    msgs = EntryProductMsgs(acc_v_commitments, claimed_sumchecks)
    spans["Entry products"] = time.perf_counter() - t0

    open_chal = transcript.get_challenge(b"open-chal")  # :313-330
    oc10 = powers(open_chal, 10)
    polynomial = K(linear_combination([ralpha_star] + accs, oc10.to_host()))
    oc10.free()
    ralpha_star_acc_mu_proof = ck.open(S(polynomial), psi, max_msm_buffer)[1]
    ralpha_star_acc_mu_evals = [e[0] for e in evaluate_le_batch([ralpha_star] + accs, psi.reshape(1, 4))]  # :332-343
    lhs = [K(hadamard(v, ep_r)) for v in (ralpha_star, r_star, alpha_star)]
    r_val_chal_a, r_val_chal_b = ip(lhs[0], val_a), ip(lhs[1], val_b)  # :348-349
    for e in ralpha_star_acc_mu_evals:
        transcript.append_fr(b"ralpha_star_acc_mu", e)
    transcript.append_g1(b"ralpha_star_mu_proof", ralpha_star_acc_mu_proof)
    for l, v in zip(lhs, (val_a, val_b, val_c)):  # :358-377
        provers.append(ElasticProver(S(l), S(v), _ONE))
    provers.append(ElasticProver(S(r_star), S(alpha_star), psi))
    t0 = time.perf_counter()
    sumcheck3 = Sumcheck.prove_batch_generic(transcript, provers)  # :380
    for p in provers:
        p.free()
    spans["sumcheck3"] = time.perf_counter() - t0

    # tensorcheck (:384-600)
    t0 = time.perf_counter()
    tc_chal = transcript.get_challenge(b"batch_challenge")
    tcc_v = powers(tc_chal, 13)
    tcc = tcc_v.to_host()
    tcc_v.free()
    bodies = [K(linear_combination(accs + [r_star], tcc)), K(linear_combination(shifts + [val_a, val_b, val_c, alpha_star], tcc)), z_star,
              K(linear_combination([ralpha_star, r_star, alpha_star], tcc))]
    ch2 = [fr_to_int(c) for c in sumcheck2.challenges]
    ch3 = [fr_to_int(c) for c in sumcheck3.challenges]
    psi_squares = [pow(pi, 1 << j, R_MOD) for j in range(len(ch3))]
    F = lambda ints: [fr_from_int(v) for v in ints]
    tc_challenges = [F([a * b % R_MOD for a, b in zip(ch3, psi_squares)][:-1]), F(ch3[:-1]), F(ch2[:-1]),
                     F([a * b % R_MOD for a, b in zip(ch2, ch3[: len(ch2)])][:-1])]
    trees = [FoldedPolynomialTree(S(b), c) for b, c in zip(bodies, tc_challenges)]
    folded_polynomials_commitments = []
    for t in trees:
        folded_polynomials_commitments.extend(ck.commit_folding(t, max_msm_buffer))
    for c in folded_polynomials_commitments:
        transcript.append_g1(b"commitment", c)
    eval_chal = transcript.get_challenge(b"evaluation-chal")
    ec = fr_to_int(eval_chal)
    pts = np.stack([fr_from_int(ec * ec % R_MOD), eval_chal, fr_from_int((-ec) % R_MOD)])
    folded_polynomials_evaluations = []  # evaluate_folding at +-eval_chal, tree by tree
    for b, chs in zip(bodies, tc_challenges):
        cur = b
        for ch in chs:
            nxt = fold_polynomial(cur, ch)
            if cur is not b:
                cur.free()
            cur = nxt
            folded_polynomials_evaluations.append(evaluate_le(cur, pts[1:]))
        if cur is not b:
            cur.free()
    base = [w_le, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c, sorted_r, sorted_alpha, sorted_z] + accs
    base_polynomials_evaluations = []
    for p in base:  # evaluate_base_polynomial appends as it goes (:36-57)
        e3 = evaluate_le(p, pts)
        for e in e3:
            transcript.append_fr(b"eval", e)
        base_polynomials_evaluations.append(e3)
    for e2 in folded_polynomials_evaluations:
        for e in e2:
            transcript.append_fr(b"eval", e)
    open_chal = transcript.get_challenge(b"open-chal")
    open_chal_len = len(folded_polynomials_evaluations) * trees[2].depth() + 3 * len(base)
    ocv = powers(open_chal, max(open_chal_len, len(base) + len(folded_polynomials_evaluations)))
    oc = ocv.to_host()
    ocv.free()
    partial_eval = K(linear_combination(base, oc[: len(base)]))
    parts = [ck.open_multi_points(S(partial_eval), pts, max_msm_buffer)[1]]
    off = len(base)
    for t in trees:
        parts.append(ck.open_folding(t, pts, oc[off: off + t.depth()], max_msm_buffer)[1])
        off += t.depth()
    evaluation_proof = g1_sum(np.stack(parts))
    tensorcheck_proof = TensorcheckProof(folded_polynomials_commitments, folded_polynomials_evaluations, evaluation_proof, base_polynomials_evaluations)
    spans["tensorcheck"] = time.perf_counter() - t0

    for v in keep:
        v.free()
    transcript.free()
    spans["ark_gemini::psnark::elastic_prover"] = time.perf_counter() - t_all
    proof = Proof(
        witness_commitment=witness_commitment, zc_alpha=zc_alpha,
        first_sumcheck_msgs=(sumcheck1.messages, sumcheck1.final_foldings),
        r_star_commitments=r_star_commitments, z_star_commitment=z_star_commitment,
        second_sumcheck_msgs=(sumcheck2.messages, sumcheck2.final_foldings),
        set_r_ep=prod[0], subset_r_ep=prod[1], sorted_r_commitment=sorted_r_commitment,
        set_alpha_ep=prod[3], subset_alpha_ep=prod[4], sorted_alpha_commitment=sorted_alpha_commitment,
        set_z_ep=prod[6], subset_z_ep=prod[7], sorted_z_commitment=sorted_z_commitment,
        ep_msgs=msgs, ralpha_star_acc_mu_evals=ralpha_star_acc_mu_evals, ralpha_star_acc_mu_proof=ralpha_star_acc_mu_proof,
        rstars_vals=[r_val_chal_a, r_val_chal_b], third_sumcheck_msgs=(sumcheck3.messages, sumcheck3.final_foldings),
        tensorcheck_proof=tensorcheck_proof)
    proof.spans = spans
    return proof

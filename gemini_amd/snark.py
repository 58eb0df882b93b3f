"""Host mirror of the non-preprocessing SNARK time prover, src/snark/time_prover.rs:19-117: pure
orchestration -- every O(n) step is a device call (MSM, sumcheck rounds, vector passes, SpMV)."""
from __future__ import annotations

import time

import numpy as np

from .circuit import R1cs
from .fr import FrVec, evaluate_le, evaluate_le_batch, fold_polynomial, fr_from_int, fr_to_int, hadamard, linear_combination, powers, reverse, tensor, R_MOD
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
    def new_time(r1cs: R1cs, ck: CommitterKey, native: bool = False) -> "Proof":
        """src/snark/time_prover.rs:19-117.  native=True: the same sequence compiled into the library
        (gm_snark_new_time, gemini_amd/csrc/snark.cpp), one call per proof; the two are byte for byte equal."""
        if native and type(ck) is CommitterKey:
            return _new_time_native(r1cs, ck)
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


class _GmSnarkProof(__import__("ctypes").Structure):
    import ctypes as _C

    _fields_ = [("witness_commitment", _C.c_uint64 * 18), ("zc_alpha", _C.c_uint64 * 4), ("rounds", _C.c_size_t * 2),
                ("messages", _C.POINTER(_C.c_uint64) * 2), ("final_foldings", (_C.c_uint64 * 8) * 2), ("nfold", _C.c_size_t),
                ("fold_commitments", _C.POINTER(_C.c_uint64)), ("fold_evaluations", _C.POINTER(_C.c_uint64)),
                ("evaluation_proof", _C.c_uint64 * 18), ("base_evaluations", _C.c_uint64 * 12), ("spans", _C.c_double * 7)]


_SPAN_NAMES = ["product_matrix_vector x3", "Commitment to w", "First sumcheck", "tensor/powers/hadamard/abc_tensored", "Second sumcheck",
               "Tensorcheck", "ark_gemini::snark::time_prover"]


def _new_time_native(r1cs: R1cs, ck: CommitterKey) -> "Proof":
    import ctypes as C

    from . import capi
    from .transcript import default_group_encoding

    cap = max(len(r1cs.z), 2).bit_length() + 2
    m = [np.zeros((cap, 8), dtype=np.uint64) for _ in range(2)]
    fc = np.zeros((cap, 18), dtype=np.uint64)
    fe = np.zeros((cap, 8), dtype=np.uint64)
    P = _GmSnarkProof()
    U = C.POINTER(C.c_uint64)
    for k in range(2):
        P.messages[k] = m[k].ctypes.data_as(U)
    P.fold_commitments = fc.ctypes.data_as(U)
    P.fold_evaluations = fe.ctypes.data_as(U)
    mats = (C.c_uint64 * 6)(*[x.handle for x in (r1cs.a, r1cs.b, r1cs.c, r1cs.at, r1cs.bt, r1cs.ct)])
    capi.check(capi.load().gm_snark_new_time(mats, C.c_uint64(r1cs.z.handle), C.c_uint64(r1cs.w.handle), C.c_uint64(ck.powers_of_g.handle),
                                             C.c_int(int(default_group_encoding())), C.c_size_t(cap), C.byref(P)))
    return _unpack_native(P, m, fc, fe, _SPAN_NAMES)


def _unpack_native(P, m, fc, fe, span_names) -> "Proof":
    A = lambda a: np.array(a, dtype=np.uint64)  # noqa: E731
    msgs = []
    for k in range(2):
        r = P.rounds[k]
        ff = A(P.final_foldings[k])
        msgs.append(([(m[k][i, :4].copy(), m[k][i, 4:].copy()) for i in range(r)], [(ff[:4].copy(), ff[4:].copy())]))
    nf = P.nfold
    be = A(P.base_evaluations).reshape(3, 4)
    tc = TensorcheckProof([fc[i].copy() for i in range(nf)], [fe[i].reshape(2, 4).copy() for i in range(nf)], A(P.evaluation_proof), [be])
    proof = Proof(A(P.witness_commitment), A(P.zc_alpha), msgs[0], msgs[1], tc)
    proof.spans = {name: P.spans[i] for i, name in enumerate(span_names) if name}
    return proof


_ELASTIC_SPAN_NAMES = [None, "Commitment to w", "First sumcheck", "MatrixTensor streams", "Second sumcheck", "Tensorcheck",
                       "ark_gemini::snark::elastic_prover"]


def _new_elastic_native(r1cs_stream, ck_stream, max_msm_buffer: int) -> "Proof":
    """gm_snark_new_elastic: the elastic prover's orchestration compiled into the library (gemini_amd/csrc/snark.cpp), one call"""
    import ctypes as C

    from . import capi
    from .transcript import default_group_encoding

    cap = max(len(r1cs_stream.z), 2).bit_length() + 2
    m = [np.zeros((cap, 8), dtype=np.uint64) for _ in range(2)]
    fc = np.zeros((cap, 18), dtype=np.uint64)
    fe = np.zeros((cap, 8), dtype=np.uint64)
    P = _GmSnarkProof()
    U = C.POINTER(C.c_uint64)
    for k in range(2):
        P.messages[k] = m[k].ctypes.data_as(U)
    P.fold_commitments = fc.ctypes.data_as(U)
    P.fold_evaluations = fe.ctypes.data_as(U)
    mats = (C.c_uint64 * 3)(*[x.handle for x in (r1cs_stream.at, r1cs_stream.bt, r1cs_stream.ct)])
    h = lambda v: C.c_uint64(v.handle)  # noqa: E731
    capi.check(capi.load().gm_snark_new_elastic(mats, h(r1cs_stream.z), h(r1cs_stream.witness), h(r1cs_stream.z_a), h(r1cs_stream.z_b), h(r1cs_stream.z_c),
                                                C.c_uint64(ck_stream.powers_of_g.handle), C.c_size_t(max_msm_buffer), C.c_size_t(ck_stream.min_device_chunk),
                                                C.c_int(int(default_group_encoding())), C.c_size_t(cap), C.byref(P)))
    return _unpack_native(P, m, fc, fe, _ELASTIC_SPAN_NAMES)


# ---- CanonicalSerialize of the proof (src/snark/mod.rs:75-82): gemini_amd/wire.py holds the formats ------------
def _proof_serialize(self, compress: bool = True, enc=0) -> bytes:
    from . import wire

    return wire.serialize(wire.SNARK_PROOF, self, compress, enc)


def _proof_deserialize(data: bytes, compress: bool = True, enc=0, validate: bool = True) -> "Proof":
    from . import wire

    return wire.deserialize(wire.SNARK_PROOF, data, compress, enc, validate)


def _proof_eq(self, other) -> bool:
    """derive(PartialEq, Eq)"""
    from . import wire

    return isinstance(other, Proof) and wire.equal(wire.SNARK_PROOF, self, other)


Proof.serialize = _proof_serialize
Proof.serialize_compressed = lambda self, enc=0: _proof_serialize(self, True, enc)      # `proof-size` of examples/snark.rs:96
Proof.serialize_uncompressed = lambda self, enc=0: _proof_serialize(self, False, enc)
Proof.deserialize = staticmethod(_proof_deserialize)
Proof.compressed_size = lambda self: len(_proof_serialize(self, True))
Proof.__eq__ = _proof_eq
Proof.__hash__ = None


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
    from .kzg import FoldedPolynomialTree
    from .msm import g1_sum

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


def new_elastic(r1cs_stream, ck_stream, max_msm_buffer: int, native: bool = False) -> Proof:
    """src/snark/elastic_prover.rs:174-266 over device-resident streams.  native: the same sequence compiled into the library
    (gm_snark_new_elastic, one call per proof) -- for a key whose stream view is the resident key itself (CommitterKeyStream)."""
    from .kzg import CommitterKeyStream

    if native and type(ck_stream) is CommitterKeyStream:
        return _new_elastic_native(r1cs_stream, ck_stream, max_msm_buffer)
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


Proof.new_elastic = staticmethod(new_elastic)

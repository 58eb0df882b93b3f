"""snark::Proof (src/snark/mod.rs:76-82) and its provers as calls into the library: gm_snark_new_time / gm_snark_new_elastic
(src/snark/time_prover.rs:19-117, elastic_prover.rs:174-266 compiled into libgemini_hip.so, gemini_amd/csrc/snark.cpp)."""
from __future__ import annotations

import time

import numpy as np

from .circuit import R1cs
from .fr import FrVec, evaluate_le, evaluate_le_batch, fold_polynomial, fr_from_int, fr_to_int, hadamard, linear_combination, powers, reverse, tensor, R_MOD
from .kzg import CommitterKey
from .sumcheck import Sumcheck
from .tensorcheck import TensorcheckProof
from .transcript import Transcript, PROTOCOL_NAME


# The product has ONE orchestration per prover: the one compiled into the library.  The step-wise Python statement of the same sequences is test
# infrastructure (tests/stepwise/snark_steps.py: the byte-for-byte cross-check); importing tests.stepwise registers it here.
_STEPWISE = {}


def register_stepwise(name: str, fn) -> None:
    _STEPWISE[name] = fn


def _stepwise(name: str):
    if name not in _STEPWISE:
        raise RuntimeError(f"snark.{name}: the native prover does not take these arguments, and the step-wise orchestration is not part of the "
                           "product (it is the tests' cross-check: `import tests.stepwise` registers it)")
    return _STEPWISE[name]


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
    def new_time(r1cs: R1cs, ck: CommitterKey, native: bool = True) -> "Proof":
        """src/snark/time_prover.rs:19-117: gm_snark_new_time, the orchestration compiled into the library (gemini_amd/csrc/snark.cpp), one call
        per proof.  native=False (or a key type the native entry does not take): the step-wise statement of the same sequence that lives with
        the tests as their cross-check (tests/stepwise), when it has been registered."""
        if native and type(ck) is CommitterKey:
            return _new_time_native(r1cs, ck)
        return _stepwise("new_time")(r1cs, ck)


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


def new_elastic(r1cs_stream, ck_stream, max_msm_buffer: int, native: bool = True) -> Proof:
    """src/snark/elastic_prover.rs:174-266 over device-resident streams: gm_snark_new_elastic, the orchestration compiled into the library, one call
    per proof -- for a key whose stream view is the resident key itself (CommitterKeyStream).  native=False (or another key type): the step-wise
    cross-check of tests/stepwise, when registered."""
    from .kzg import CommitterKeyStream

    if native and type(ck_stream) is CommitterKeyStream:
        return _new_elastic_native(r1cs_stream, ck_stream, max_msm_buffer)
    return _stepwise("new_elastic")(r1cs_stream, ck_stream, max_msm_buffer)


Proof.new_elastic = staticmethod(new_elastic)

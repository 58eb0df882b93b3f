"""psnark::Proof (src/psnark/mod.rs:29-51), its index (src/psnark/time_prover.rs:49-64, the joint matrices of src/misc.rs:269-366) and its provers as
calls into the library: gm_psnark_new_time / gm_psnark_new_elastic (src/psnark/time_prover.rs:69-384, elastic_prover.rs:60-634 compiled into
libgemini_hip.so); the plookup vector builders (src/subprotocols/plookup/time_prover.rs:89-112) as primitives."""
from __future__ import annotations

import time

import numpy as np

from .circuit import R1cs
from .fr import (FrVec, IdxVec, R_MOD, accumulated_product_monic, alg_hash, element, evaluate_le, evaluate_le_batch, fr_from_int, fr_to_int, hadamard, ip,
                 linear_combination, lookup, plookup_set, plookup_subset, powers, shift_monic, tensor)
from .kzg import CommitterKey
from .sumcheck import Sumcheck, TimeProver
from .tensorcheck import TensorcheckProof
from .transcript import PROTOCOL_NAME, Transcript

_ONE = fr_from_int(1)


# ---- src/misc.rs:269-366 on the host CSR arrays -------------------------------------------------------
def joint_matrices(a, b, c, num_constraints: int, num_variables: int):
    """sum_matrices + joint_matrices: the union of the supports of A, B, C walked column-major
    (column ascending, row ascending inside a column -- BTreeSet order), with the three value
    vectors (zero where a matrix has no entry; the last duplicate wins like BTreeMap::collect).
    Returns (row_index, col_index, val_a, val_b, val_c) as numpy arrays (values Montgomery (nnz, 4))."""
    keys, vals = {}, {}
    for m in (a, b, c):
        if id(m) in keys:  # the same matrix object used twice (dummy_r1cs: A = B = C)
            continue
        rowptr, cols, v = m.csr
        rows = np.repeat(np.arange(m.nrows, dtype=np.uint64), np.diff(rowptr.astype(np.int64)))
        assert cols.size == 0 or int(cols.max()) < num_variables, "column index outside num_variables"
        keys[id(m)] = cols.astype(np.uint64) * np.uint64(num_constraints) + rows
        vals[id(m)] = np.asarray(v, dtype=np.uint64).reshape(-1, 4)
    union = np.unique(np.concatenate(list(keys.values())))
    row_index = (union % np.uint64(num_constraints)).astype(np.uint32)
    col_index = (union // np.uint64(num_constraints)).astype(np.uint32)
    dense = {}
    for mid, k in keys.items():
        d = np.zeros((len(union), 4), dtype=np.uint64)
        order = np.argsort(k, kind="stable")
        ks = k[order]
        last = np.ones(len(ks), dtype=bool)
        last[:-1] = ks[1:] != ks[:-1]  # the last occurrence of every key
        if not last.all():
            import warnings

            warnings.warn("psnark: a matrix row repeats a column; the joint-matrix encoding keeps the last entry only (as the "
                          "reference's BTreeMap does, src/misc.rs:269-366) and the resulting proof will NOT verify -- coalesce "
                          "repeated (row, column) entries before building the R1CS", RuntimeWarning, stacklevel=3)
        d[np.searchsorted(union, ks[last])] = vals[mid][order][last]
        dense[mid] = d
    return row_index, col_index, dense[id(a)], dense[id(b)], dense[id(c)]


def _joint(r1cs: R1cs):
    """joint_matrices of an instance, computed once per R1cs object: they depend on the matrices only
    (the reference recomputes them in `index` and in `new_time`, src/psnark/time_prover.rs:53-61,102-110)."""
    j = getattr(r1cs, "_joint_matrices", None)
    if j is None:
        j = joint_matrices(r1cs.a, r1cs.b, r1cs.c, r1cs.a.nrows, len(r1cs.z))
        r1cs._joint_matrices = j
    return j


class _JointDevice:
    """the joint-matrix vectors of an instance resident in HBM: a function of the matrices only (what `index` commits to,
    src/psnark/time_prover.rs:49-64), so they are uploaded ONCE per R1cs object and live with it -- like its CSR matrices,
    they are part of the instance that is resident before the prover starts.  (The reference recomputes joint_matrices on the
    CPU inside new_time, :99-110; re-uploading 5 vectors per proof kept the GPU idle for 4 % of psnark -i 22.)"""

    def __init__(self, r1cs: R1cs):
        row_index_h, col_index_h, val_a_h, val_b_h, val_c_h = _joint(r1cs)
        self.row_index_h, self.col_index_h = row_index_h, col_index_h
        self.row_index, self.col_index = IdxVec.from_host(row_index_h), IdxVec.from_host(col_index_h)
        self.row, self.col = _field_of_index(self.row_index), _field_of_index(self.col_index)
        self.val_a, self.val_b, self.val_c = FrVec.from_host(val_a_h), FrVec.from_host(val_b_h), FrVec.from_host(val_c_h)
        self.ext_fre = {}

    def extended_frequencies(self, len_r: int, len_z: int):
        """[extend_frequency(compute_frequency(..)) for the row and the column index] as device index vectors
        (plookup/time_prover.rs:66-79; src/psnark/time_prover.rs:165-178): instance-only as well -- the bincount / repeat over
        the index arrays kept the GPU idle for 44 ms of a 480 ms proof at 2^22 when done per proof"""
        key = (len_r, len_z)
        if key not in self.ext_fre:
            self.ext_fre[key] = [IdxVec.from_host(extend_frequency(compute_frequency(len_r, self.row_index_h))),
                                 IdxVec.from_host(extend_frequency(compute_frequency(len_z, self.col_index_h)))]
        return self.ext_fre[key]

    def free(self):
        for v in (self.row_index, self.col_index, self.row, self.col, self.val_a, self.val_b, self.val_c):
            v.free()
        for pair in self.ext_fre.values():
            for v in pair:
                v.free()
        self.ext_fre = {}


def _joint_device(r1cs: R1cs) -> "_JointDevice":
    cache = r1cs.__dict__.setdefault("_device_cache", {})
    if "joint" not in cache:
        cache["joint"] = _JointDevice(r1cs)
    return cache["joint"]


class _JointNative:
    """the same matrix-only part of the instance built INSIDE the library (gm_psnark_preprocess: joint support, value vectors,
    row / col, extended frequencies; gemini_amd/csrc/psnark.cpp) -- what a Rust / C++ integrator calls instead of restating
    src/misc.rs:269-366 in the shim.  Kept with the R1cs object like _JointDevice and freed with it."""

    def __init__(self, r1cs: R1cs):
        import ctypes as C

        from . import capi

        Instance, _ = _psnark_ctypes()
        self.rec = Instance()
        capi.check(capi.load().gm_psnark_preprocess(C.c_uint64(r1cs.a.handle), C.c_uint64(r1cs.b.handle), C.c_uint64(r1cs.c.handle),
                                                    C.c_size_t(len(r1cs.z)), C.byref(self.rec)))

    def index(self, ck: CommitterKey) -> list:
        import ctypes as C

        from . import capi

        out = np.zeros((5, 18), dtype=np.uint64)
        capi.check(capi.load().gm_psnark_index(C.byref(self.rec), C.c_uint64(ck.powers_of_g.handle), capi.ptr(out)))
        return [out[k].copy() for k in range(5)]

    def free(self):
        import ctypes as C

        from . import capi

        capi.check(capi.load().gm_psnark_preprocess_free(C.byref(self.rec)))


def _joint_native(r1cs: R1cs) -> "_JointNative":
    cache = r1cs.__dict__.setdefault("_device_cache", {})
    if "joint_native" not in cache:
        cache["joint_native"] = _JointNative(r1cs)
    return cache["joint_native"]


def _field_of_index(index: IdxVec) -> FrVec:
    """[F::from(i) for i in index] (the `row` / `col` vectors, src/misc.rs:343-349)"""
    zeros = FrVec.alloc(len(index))
    zeros.fill(fr_from_int(0))
    out = alg_hash(zeros, index, _ONE)
    zeros.free()
    return out


# ---- src/subprotocols/plookup/time_prover.rs ------------------------------------------------------------
def compute_frequency(set_len: int, index: np.ndarray) -> np.ndarray:
    """:66-70"""
    return 1 + np.bincount(index, minlength=set_len).astype(np.int64)


def extend_frequency(frequency: np.ndarray) -> np.ndarray:
    """:72-79"""
    return np.repeat(np.arange(len(frequency), dtype=np.uint32), frequency)


def plookup(subset: FrVec, set_: FrVec, index: IdxVec, ext_fre: IdxVec, y, z, zeta):
    """:89-112 -> [lookup_set, lookup_subset, lookup_sorted]; ext_fre = extend_frequency(compute_frequency(..))
    so that sorted(set, frequency) = [set[i] for i in ext_fre]"""
    tmp = []
    if fr_to_int(zeta) != 0:
        set_h = alg_hash(set_, None, zeta)
        subset_h = alg_hash(subset, index, zeta)
        tmp += [set_h, subset_h]
    else:
        set_h, subset_h = set_, subset
    lookup_set = plookup_set(set_h, y, z)
    lookup_subset = plookup_subset(subset_h, y)
    srt = lookup(set_h, ext_fre)
    lookup_sorted = plookup_set(srt, y, z)
    for v in tmp + [srt]:
        v.free()
    return [lookup_set, lookup_subset, lookup_sorted]


# ---- src/subprotocols/entryproduct -----------------------------------------------------------------------
class EntryProductMsgs:
    """entryproduct/mod.rs:21-25"""

    def __init__(self, acc_v_commitments, claimed_sumchecks):
        self.acc_v_commitments = acc_v_commitments
        self.claimed_sumchecks = claimed_sumchecks


# ONE orchestration per prover in the product: the one compiled into the library.  The step-wise statement (and the entry-product argument's,
# `EntryProduct`) is test infrastructure: tests/stepwise/psnark_steps.py, registered here by `import tests.stepwise`.
_STEPWISE = {}


def register_stepwise(name: str, fn) -> None:
    _STEPWISE[name] = fn


def _stepwise(name: str):
    if name not in _STEPWISE:
        raise RuntimeError(f"psnark.{name}: the native prover does not take these arguments, and the step-wise orchestration is not part of the "
                           "product (it is the tests' cross-check: `import tests.stepwise` registers it)")
    return _STEPWISE[name]


class Proof:
    """src/psnark/mod.rs:29-51"""

    FIELDS = ("witness_commitment", "zc_alpha", "first_sumcheck_msgs", "r_star_commitments", "z_star_commitment", "second_sumcheck_msgs",
              "set_r_ep", "subset_r_ep", "sorted_r_commitment", "set_alpha_ep", "subset_alpha_ep", "sorted_alpha_commitment", "set_z_ep",
              "subset_z_ep", "sorted_z_commitment", "ep_msgs", "ralpha_star_acc_mu_evals", "ralpha_star_acc_mu_proof", "rstars_vals",
              "third_sumcheck_msgs", "tensorcheck_proof")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])
        self.spans = {}

    @staticmethod
    def index(ck: CommitterKey, r1cs: R1cs, native: bool = False) -> list:
        """src/psnark/time_prover.rs:49-64.  native: joint matrices and commitments inside the library (gm_psnark_preprocess +
        gm_psnark_index)"""
        if native:
            return _joint_native(r1cs).index(ck)
        jd = _joint_device(r1cs)
        return ck.batch_commit([jd.row, jd.col, jd.val_a, jd.val_b, jd.val_c])

    @staticmethod
    def new_time(ck: CommitterKey, r1cs: R1cs, index: list, native=True) -> "Proof":
        """src/psnark/time_prover.rs:69-384: gm_psnark_new_time, the orchestration compiled into the library, one call per proof -- for a
        CommitterKey (whole, or a cyclic share); native = "preprocess": the matrix-only part of the instance record comes from
        gm_psnark_preprocess as well (nothing of src/misc.rs:269-366 is left to the caller).  native=False (or another key type): the step-wise
        cross-check of tests/stepwise, when registered."""
        if native and type(ck) is CommitterKey:
            return _new_time_native(ck, r1cs, index, preprocess_in_library=native == "preprocess")
        return _stepwise("new_time")(ck, r1cs, index)

    @staticmethod
    def new_elastic(ck, r1cs_stream, index: list, max_msm_buffer: int, native: bool = True) -> "Proof":
        """src/psnark/elastic_prover.rs:60-634 over device-resident streams (`ck`: a CommitterKeyStream): gm_psnark_new_elastic
        (gemini_amd/csrc/psnark_elastic.cpp), one call per proof.  The reference's test asserts this proof equals new_time's
        (src/psnark/tests.rs:56-124).  native=False (or another key type): the step-wise cross-check of tests/stepwise, when registered."""
        from .kzg import CommitterKeyStream

        if native and type(ck) is CommitterKeyStream:
            return _new_time_native(ck, r1cs_stream.r1cs, index, elastic=(ck, r1cs_stream, max_msm_buffer))
        return _stepwise("new_elastic")(ck, r1cs_stream, index, max_msm_buffer)

    def serialize(self, compress: bool = True, enc=0) -> bytes:
        """derive(CanonicalSerialize) of src/psnark/mod.rs:29-51 (formats: gemini_amd/wire.py)"""
        from . import wire

        return wire.serialize(wire.PSNARK_PROOF, self, compress, enc)

    def serialize_compressed(self, enc=0) -> bytes:
        return self.serialize(True, enc)

    def serialize_uncompressed(self, enc=0) -> bytes:
        return self.serialize(False, enc)

    @staticmethod
    def deserialize(data: bytes, compress: bool = True, enc=0, validate: bool = True) -> "Proof":
        from . import wire

        return wire.deserialize(wire.PSNARK_PROOF, data, compress, enc, validate)

    def __eq__(self, other) -> bool:
        """derive(PartialEq, Eq)"""
        from . import wire

        return isinstance(other, Proof) and wire.equal(wire.PSNARK_PROOF, self, other)

    __hash__ = None

    def compressed_size(self) -> int:
        return len(self.serialize_compressed())


# ---- gm_psnark_new_time: the orchestration above compiled into the library (gemini_amd/csrc/psnark.cpp) -------------------
_PSNARK_SPANS = ["Commitment to w", "First sumcheck", "joint matrices", "Commitments to z* and r*", "Second sumcheck",
                 "Commitments to sorted vectors", "plookup vectors + accumulated products", "Entry products", "Opening at psi", "Third sumcheck",
                 "Tensorcheck", "ark_gemini::psnark::time_prover"]


def _psnark_ctypes():
    import ctypes as C

    U64P = C.POINTER(C.c_uint64)

    class Instance(C.Structure):
        _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("z", C.c_uint64), ("w", C.c_uint64), ("row_index", C.c_uint64),
                    ("col_index", C.c_uint64), ("nnz", C.c_size_t), ("row", C.c_uint64), ("col", C.c_uint64), ("val_a", C.c_uint64),
                    ("val_b", C.c_uint64), ("val_c", C.c_uint64), ("ext_fre_row", C.c_uint64), ("ext_fre_col", C.c_uint64),
                    ("ext_fre_row_len", C.c_size_t), ("ext_fre_col_len", C.c_size_t), ("index_commitments", U64P),
                    ("ck_g2_bytes", C.POINTER(C.c_uint8)), ("ck_g2_len", C.c_size_t)]

    class ProofRec(C.Structure):
        _fields_ = [("witness_commitment", C.c_uint64 * 18), ("zc_alpha", C.c_uint64 * 4), ("rounds", C.c_size_t * 3), ("messages", U64P * 3),
                    ("final_foldings", (C.c_uint64 * 8) * 2), ("third_final_foldings", (C.c_uint64 * 8) * 13),
                    ("r_star_commitments", (C.c_uint64 * 18) * 3), ("z_star_commitment", C.c_uint64 * 18),
                    ("sorted_commitments", (C.c_uint64 * 18) * 3), ("products", (C.c_uint64 * 4) * 9),
                    ("acc_v_commitments", (C.c_uint64 * 18) * 9), ("claimed_sumchecks", (C.c_uint64 * 4) * 9),
                    ("ralpha_star_acc_mu_evals", (C.c_uint64 * 4) * 10), ("ralpha_star_acc_mu_proof", C.c_uint64 * 18),
                    ("rstars_vals", (C.c_uint64 * 4) * 2), ("nfold", C.c_size_t), ("cap_folds", C.c_size_t), ("fold_commitments", U64P),
                    ("fold_evaluations", U64P), ("evaluation_proof", C.c_uint64 * 18), ("base_evaluations", (C.c_uint64 * 12) * 22),
                    ("spans", C.c_double * 12)]

    return Instance, ProofRec


def _proof_buffers(num_variables: int, nnz: int):
    """a gm_psnark_proof record with its caller-owned arrays: (record, cap_rounds, (messages, fold commitments, fold evaluations))"""
    import ctypes as C

    _, ProofRec = _psnark_ctypes()
    U = C.POINTER(C.c_uint64)
    cap = max(2 * max(num_variables, nnz) + 4, 4).bit_length() + 3
    m = [np.zeros((cap, 8), dtype=np.uint64) for _ in range(3)]
    cap_folds = 4 * cap
    fc = np.zeros((cap_folds, 18), dtype=np.uint64)
    fe = np.zeros((cap_folds, 8), dtype=np.uint64)
    P = ProofRec()
    for k in range(3):
        P.messages[k] = m[k].ctypes.data_as(U)
    P.cap_folds = cap_folds
    P.fold_commitments = fc.ctypes.data_as(U)
    P.fold_evaluations = fe.ctypes.data_as(U)
    return P, cap, (m, fc, fe)


def _unpack_proof(P, bufs) -> "Proof":
    m, fc, fe = bufs
    A = lambda a: np.array(a, dtype=np.uint64)  # noqa: E731
    msgs = lambda k: [(m[k][i, :4].copy(), m[k][i, 4:].copy()) for i in range(P.rounds[k])]  # noqa: E731
    ff = lambda rec: [(A(rec)[:4].copy(), A(rec)[4:].copy())]  # noqa: E731
    prods = [A(P.products[k]) for k in range(9)]
    nf = P.nfold
    tc = TensorcheckProof([fc[i].copy() for i in range(nf)], [fe[i].reshape(2, 4).copy() for i in range(nf)], A(P.evaluation_proof),
                          [A(P.base_evaluations[k]).reshape(3, 4) for k in range(22)])
    ep = EntryProductMsgs([A(P.acc_v_commitments[k]) for k in range(9)], [A(P.claimed_sumchecks[k]) for k in range(9)])
    third_ff = []
    for j in range(13):
        r = A(P.third_final_foldings[j])
        third_ff.append((r[:4].copy(), r[4:].copy()))
    sc = [A(P.sorted_commitments[k]) for k in range(3)]
    proof = Proof(
        witness_commitment=A(P.witness_commitment), zc_alpha=A(P.zc_alpha), first_sumcheck_msgs=(msgs(0), ff(P.final_foldings[0])),
        r_star_commitments=[A(P.r_star_commitments[k]) for k in range(3)], z_star_commitment=A(P.z_star_commitment),
        second_sumcheck_msgs=(msgs(1), ff(P.final_foldings[1])),
        set_r_ep=prods[0], subset_r_ep=prods[1], sorted_r_commitment=sc[0], set_alpha_ep=prods[3], subset_alpha_ep=prods[4],
        sorted_alpha_commitment=sc[1], set_z_ep=prods[6], subset_z_ep=prods[7], sorted_z_commitment=sc[2], ep_msgs=ep,
        ralpha_star_acc_mu_evals=[A(P.ralpha_star_acc_mu_evals[k]) for k in range(10)], ralpha_star_acc_mu_proof=A(P.ralpha_star_acc_mu_proof),
        rstars_vals=[A(P.rstars_vals[0]), A(P.rstars_vals[1])], third_sumcheck_msgs=(msgs(2), third_ff), tensorcheck_proof=tc)
    return proof


def _new_time_native(ck: CommitterKey, r1cs: R1cs, index: list, preprocess_in_library: bool = False, elastic=None) -> "Proof":
    """gm_psnark_new_time; elastic = (ck_stream, r1cs_stream, max_msm_buffer): gm_psnark_new_elastic over the same instance record"""
    import ctypes as C

    from . import capi
    from .transcript import default_group_encoding

    Instance, ProofRec = _psnark_ctypes()
    U = C.POINTER(C.c_uint64)
    idx = np.ascontiguousarray(np.stack(index), dtype=np.uint64)
    g2 = ck.powers_of_g2_bytes()
    g2buf = (C.c_uint8 * len(g2)).from_buffer_copy(g2)
    if preprocess_in_library:  # the matrix-only part of the record comes from gm_psnark_preprocess; z, w, index, G2 bytes are ours
        I = Instance.from_buffer_copy(_joint_native(r1cs).rec)
        I.z, I.w = r1cs.z.handle, r1cs.w.handle
        I.index_commitments = idx.ctypes.data_as(U)
        I.ck_g2_bytes, I.ck_g2_len = C.cast(g2buf, C.POINTER(C.c_uint8)), len(g2)
        nnz = I.nnz
    else:
        jd = _joint_device(r1cs)
        nrows = max(r1cs.a.nrows, r1cs.b.nrows)
        len_r = 1 << max(nrows - 1, 0).bit_length()  # the first sumcheck's tensor: 2^rounds, rounds = ceil(log2 max(|A z|, |B z|))
        ext = jd.extended_frequencies(len_r, len(r1cs.z))
        I = Instance(r1cs.a.handle, r1cs.b.handle, r1cs.c.handle, r1cs.z.handle, r1cs.w.handle, jd.row_index.handle, jd.col_index.handle,
                     len(jd.row_index), jd.row.handle, jd.col.handle, jd.val_a.handle, jd.val_b.handle, jd.val_c.handle, ext[0].handle, ext[1].handle,
                     len(ext[0]), len(ext[1]), idx.ctypes.data_as(U), C.cast(g2buf, C.POINTER(C.c_uint8)), len(g2))
        nnz = len(jd.row_index)
    P, cap, bufs = _proof_buffers(len(r1cs.z), nnz)
    if elastic is None:
        capi.check(capi.load().gm_psnark_new_time(C.byref(I), C.c_uint64(ck.powers_of_g.handle), C.c_int(int(default_group_encoding())), C.c_size_t(cap),
                                                  C.byref(P)))
    else:
        cks, st, max_msm_buffer = elastic
        h = lambda v: C.c_uint64(v.handle)  # noqa: E731
        capi.check(capi.load().gm_psnark_new_elastic(C.byref(I), h(st.z), h(st.witness), h(st.z_a), h(st.z_b), h(st.z_c), C.c_uint64(cks.powers_of_g.handle),
                                                     C.c_size_t(max_msm_buffer), C.c_size_t(cks.min_device_chunk), C.c_int(int(default_group_encoding())),
                                                     C.c_size_t(cap), C.byref(P)))
    proof = _unpack_proof(P, bufs)
    proof.spans = {name: P.spans[i] for i, name in enumerate(_PSNARK_SPANS)}
    if elastic is not None:
        proof.spans["ark_gemini::psnark::elastic_prover"] = proof.spans.pop("ark_gemini::psnark::time_prover")
    return proof

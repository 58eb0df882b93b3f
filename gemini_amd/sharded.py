"""ctypes face of the N-GPU provers compiled into the library (gemini_amd/csrc/sharded.cpp over gemini_amd/csrc/dist.cpp):

  * `cyclic_committer_key`   a `CommitterKey` (src/kzg/time.rs:24-27) whose G1 half is this rank's ELEMENT-CYCLIC share; every
                             native prover handed such a key (Proof.new_time(.., native=True), new_elastic, psnark) runs on N
                             GPUs with the MSMs sharded and one all-gather of k x 144 bytes per batch_commit (:81-107)
  * `R1csShard`, `ShardKey`, `new_time_sharded`
                             snark::Proof::new_time (src/snark/time_prover.rs:19-117) with every vector block-sharded:
                             gm_snark_new_time_sharded, general sparse matrices (row blocks, global columns) or block-diagonal.

Which transport carries the all-gathers is chosen with gemini_amd.collective (RCCL, shared memory, a hook)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi, collective
from .fr import R_MOD, FrVec, fr_from_int


def cyclic_committer_key(max_degree: int, max_eval_points: int, tau_canonical, g_affine=None, with_g2: bool = True):
    """CommitterKey::new (src/kzg/time.rs:49-72), each rank generating its own share: powers i = rank (mod world)"""
    from . import g2 as G2
    from .kzg import CommitterKey, g1_generator_mont
    from .msm import G1Bases

    capi.ensure_init()
    rank, world, _ = collective.info()
    g = g1_generator_mont() if g_affine is None else g_affine
    tau = np.ascontiguousarray(tau_canonical, dtype=np.uint64).reshape(4)
    h = C.c_uint64()
    n = max_degree + 1
    capi.check(capi.load().gm_g1_srs_register_cyclic(capi.ptr(capi.u64(g).reshape(12)), capi.ptr(tau), C.c_size_t(n), C.c_int(rank), C.c_int(world), C.byref(h)))
    local = -(-(n - rank) // world) if n > rank else 0
    t = sum(int(v) << (64 * i) for i, v in enumerate(tau))
    g2s = [G2.mul(G2.generator(), pow(t, i, R_MOD)) for i in range(max_eval_points + 1)] if with_g2 else None
    ck = CommitterKey(G1Bases(h.value, local), max_eval_points, g2s)
    ck.n_global = n
    return ck


class _GmSnarkShard(C.Structure):
    _fields_ = [("matrices", C.c_uint64 * 6), ("z", C.c_uint64), ("w_block", C.c_uint64), ("key", C.c_uint64), ("key_offsets", C.POINTER(C.c_size_t)),
                ("key_counts", C.POINTER(C.c_size_t)), ("key_segments", C.c_size_t), ("n", C.c_size_t), ("tail_log", C.c_size_t)]


class ShardKey:
    """gm_snark_shard_key_new: this rank's slice of the key for every block-sharded level + the replicated prefix, one handle"""

    def __init__(self, n: int, tail_log: int, tau_canonical, g_affine=None):
        from .kzg import g1_generator_mont
        from .msm import G1Bases

        capi.ensure_init()
        g = g1_generator_mont() if g_affine is None else g_affine
        self.offsets = (C.c_size_t * 64)()
        self.counts = (C.c_size_t * 64)()
        h, nseg = C.c_uint64(), C.c_size_t()
        capi.check(capi.load().gm_snark_shard_key_new(capi.ptr(capi.u64(g).reshape(12)), capi.ptr(np.ascontiguousarray(tau_canonical, dtype=np.uint64).reshape(4)),
                                                      C.c_size_t(n), C.c_size_t(tail_log), C.byref(h), self.offsets, self.counts, C.byref(nseg)))
        self.n, self.tail_log, self.segments = n, tail_log, nseg.value
        self.bases = G1Bases(h.value, sum(self.counts[i] for i in range(nseg.value)))

    def free(self):
        self.bases.free()


class R1csShard:
    """rows [r m, (r + 1) m) of A, B, C and of their transposes, z (whole: global columns; or this rank's block: block-diagonal)
    and this rank's block of w"""

    def __init__(self, mats, z: FrVec, w_block: FrVec, n: int):
        self.mats, self.z, self.w_block, self.n = list(mats), z, w_block, n

    @classmethod
    def dummy(cls, e_canonical: int, n: int, global_columns: bool = False) -> "R1csShard":
        """this rank's rows of dummy_r1cs(e, n) (src/circuit.rs:349-365: z = [e; n], w = [e; n - 1], A = B = C = diag(1 / e)),
        as a block-diagonal instance (local columns) or as a general one (global columns, z whole)"""
        from .circuit import SparseMatrix

        rank, world, _ = collective.info()
        m = n // world
        e = e_canonical % R_MOD
        cols = np.arange(m, dtype=np.uint32) + (np.uint32(rank * m) if global_columns else np.uint32(0))
        d = SparseMatrix.from_csr(np.arange(m + 1, dtype=np.uint64), cols, np.tile(fr_from_int(pow(e, -1, R_MOD)), (m, 1)), m, n if global_columns else m)
        z = FrVec.alloc(n if global_columns else m)
        z.fill(fr_from_int(e))
        w = FrVec.alloc(m - 1 if rank == world - 1 else m)
        w.fill(fr_from_int(e))
        return cls([d] * 6, z, w, n)

    @classmethod
    def from_rows(cls, a_rows, b_rows, c_rows, z_mont: np.ndarray, nx: int) -> "R1csShard":
        """a general instance given whole on every rank (rows of (value_mont, column) pairs, src/circuit.rs:43): keeps this rank's
        row blocks of the matrices and of their transposes, z whole, its block of w = z[nx:]"""
        from .circuit import SparseMatrix

        rank, world, _ = collective.info()
        n = len(z_mont)
        m = n // world
        lo, hi = rank * m, (rank + 1) * m
        mats = [SparseMatrix.from_rows(rows[lo:hi], n) for rows in (a_rows, b_rows, c_rows)]
        for rows in (a_rows, b_rows, c_rows):
            t_rows = [[] for _ in range(m)]
            for i, row in enumerate(rows):
                for v, c in row:
                    if lo <= c < hi:
                        t_rows[c - lo].append((v, i))
            mats.append(SparseMatrix.from_rows(t_rows, n))
        w = np.ascontiguousarray(z_mont[nx:][lo:hi])
        return cls(mats, FrVec.from_host(z_mont), FrVec.from_host(w), n)

    def free(self):
        seen = set()
        for mtx in self.mats:
            if id(mtx) not in seen:
                seen.add(id(mtx))
                mtx.free()
        self.z.free()
        self.w_block.free()


def new_time_sharded(shard: R1csShard, key: ShardKey):
    """gm_snark_new_time_sharded; the same `snark.Proof` on every rank"""
    from .snark import _SPAN_NAMES, _GmSnarkProof, _unpack_native
    from .transcript import default_group_encoding

    cap = max(shard.n, 2).bit_length() + 2
    m = [np.zeros((cap, 8), dtype=np.uint64) for _ in range(2)]
    fc = np.zeros((cap, 18), dtype=np.uint64)
    fe = np.zeros((cap, 8), dtype=np.uint64)
    P = _GmSnarkProof()
    U = C.POINTER(C.c_uint64)
    for k in range(2):
        P.messages[k] = m[k].ctypes.data_as(U)
    P.fold_commitments = fc.ctypes.data_as(U)
    P.fold_evaluations = fe.ctypes.data_as(U)
    S = _GmSnarkShard()
    for k in range(6):
        S.matrices[k] = shard.mats[k].handle
    S.z, S.w_block, S.key = shard.z.handle, shard.w_block.handle, key.bases.handle
    S.key_offsets = C.cast(key.offsets, C.POINTER(C.c_size_t))
    S.key_counts = C.cast(key.counts, C.POINTER(C.c_size_t))
    S.key_segments, S.n, S.tail_log = key.segments, shard.n, key.tail_log
    capi.check(capi.load().gm_snark_new_time_sharded(C.byref(S), C.c_int(int(default_group_encoding())), C.c_size_t(cap), C.byref(P)))
    return _unpack_native(P, m, fc, fe, _SPAN_NAMES)

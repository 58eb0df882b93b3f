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


def new_time_sharded(shard: R1csShard, key: ShardKey, elastic=None):
    """gm_snark_new_time_sharded; the same `snark.Proof` on every rank.  elastic = (max_msm_buffer, min_device_chunk): gm_snark_new_elastic_sharded
    (the resident schedule of the elastic prover over the same blocks)"""
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
    if elastic is None:
        capi.check(capi.load().gm_snark_new_time_sharded(C.byref(S), C.c_int(int(default_group_encoding())), C.c_size_t(cap), C.byref(P)))
        return _unpack_native(P, m, fc, fe, _SPAN_NAMES)
    capi.check(capi.load().gm_snark_new_elastic_sharded(C.byref(S), C.c_size_t(elastic[0]), C.c_size_t(elastic[1]), C.c_int(int(default_group_encoding())), C.c_size_t(cap),
                                                        C.byref(P)))
    proof = _unpack_native(P, m, fc, fe, _SPAN_NAMES)
    proof.spans["ark_gemini::snark::elastic_prover"] = proof.spans.pop("ark_gemini::snark::time_prover")
    return proof


# ---- psnark::Proof::new_time with every vector block-sharded (gemini_amd/csrc/psnark_sharded.cpp) ---------------------------------
class _GmPsnarkShard(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("z", C.c_uint64), ("w_block", C.c_uint64), ("w_len", C.c_size_t),
                ("row_index", C.c_uint64), ("col_index", C.c_uint64), ("row", C.c_uint64), ("col", C.c_uint64), ("val_a", C.c_uint64),
                ("val_b", C.c_uint64), ("val_c", C.c_uint64), ("ext_fre_row", C.c_uint64), ("ext_fre_col", C.c_uint64),
                ("ext_fre_row_len", C.c_size_t), ("ext_fre_col_len", C.c_size_t), ("num_constraints", C.c_size_t), ("num_variables", C.c_size_t),
                ("nnz", C.c_size_t), ("block", C.c_size_t), ("tail_log", C.c_size_t), ("key", C.c_uint64), ("key_offsets", C.POINTER(C.c_size_t)),
                ("key_counts", C.POINTER(C.c_size_t)), ("key_segments", C.c_size_t), ("key_len", C.c_size_t),
                ("index_commitments", C.POINTER(C.c_uint64)), ("ck_g2_bytes", C.POINTER(C.c_uint8)), ("ck_g2_len", C.c_size_t)]


def psnark_shard_block(longest: int, world: int) -> int:
    lib = capi.load()
    lib.gm_psnark_shard_block.restype = C.c_size_t
    return int(lib.gm_psnark_shard_block(C.c_size_t(longest), C.c_int(world)))


class PsnarkShardKey:
    """gm_psnark_shard_key_new: this rank's slices of a key of max_degree + 1 powers for block size `block` (levels of M >> j while they stay
    sharded, then the replicated prefix), one handle; plus the G2 half every rank holds whole (the transcript absorbs it)"""

    def __init__(self, max_degree: int, block: int, tail_log: int, tau_canonical, max_eval_points: int = 5, g_affine=None):
        from . import g2 as G2

        n_key = max_degree + 1  # CommitterKey::new(max_degree, ..) holds max_degree + 1 powers (src/kzg/time.rs:49-72)
        from .kzg import g1_generator_mont
        from .msm import G1Bases

        capi.ensure_init()
        g = g1_generator_mont() if g_affine is None else g_affine
        tau = np.ascontiguousarray(tau_canonical, dtype=np.uint64).reshape(4)
        self.offsets = (C.c_size_t * 64)()
        self.counts = (C.c_size_t * 64)()
        h, nseg = C.c_uint64(), C.c_size_t()
        capi.check(capi.load().gm_psnark_shard_key_new(capi.ptr(capi.u64(g).reshape(12)), capi.ptr(tau), C.c_size_t(n_key), C.c_size_t(block), C.c_size_t(tail_log),
                                                       C.byref(h), self.offsets, self.counts, C.byref(nseg)))
        self.n_key, self.block, self.tail_log, self.segments = n_key, block, tail_log, nseg.value
        self.bases = G1Bases(h.value, 0)
        t = sum(int(v) << (64 * i) for i, v in enumerate(tau))
        self.g2_bytes = G2.serialize_vec_uncompressed([G2.mul(G2.generator(), pow(t, i, R_MOD)) for i in range(max_eval_points + 1)])

    def free(self):
        self.bases.free()


class PsnarkShard:
    """this rank's blocks of a psnark instance: its row block of A, B, C (global columns), z whole, the blocks of w and of the joint-matrix
    vectors (src/misc.rs:269-366) and extended frequencies (plookup/time_prover.rs:66-79), each family cut at ITS level (blocks of
    block >> level: gm_psnark_shard_level).  Built from a whole `R1cs` (every rank walks the same host arrays and keeps its slice)."""

    def __init__(self, r1cs, tail_log: int = 10, block: int = None):
        from .circuit import SparseMatrix
        from .fr import IdxVec
        from .psnark import _field_of_index, _joint, compute_frequency, extend_frequency

        rank, world, _ = collective.info()
        row_index, col_index, val_a, val_b, val_c = _joint(r1cs)
        nrows, nz, nnz = r1cs.a.nrows, len(r1cs.z), len(row_index)
        nt = 1 << max(nrows - 1, 0).bit_length()
        ext_row = extend_frequency(compute_frequency(nt, row_index))
        ext_col = extend_frequency(compute_frequency(nz, col_index))
        self.longest = max(len(ext_row) + 2, len(ext_col) + 2, nz + 2, nrows + 2, nnz + 1)
        self.block = block or psnark_shard_block(self.longest, world)
        self.tail_log = tail_log
        lib = capi.load()
        lib.gm_psnark_shard_level.restype = C.c_size_t
        self.levels = {}  # whole length of a family -> its level

        def cut(length: int):
            """[lo, hi) of this rank's block of a family whose longest member has `length` elements (gm_psnark_shard_level)"""
            s = int(lib.gm_psnark_shard_level(C.c_size_t(length), C.c_size_t(self.block), C.c_size_t(tail_log), C.c_int(world)))
            self.levels[length] = s
            b = self.block >> s
            return rank * b, (rank + 1) * b

        self.num_constraints, self.num_variables, self.nnz = nrows, nz, nnz
        self.ext_row_len, self.ext_col_len, self.w_len = len(ext_row), len(ext_col), len(r1cs.w)
        self.z = r1cs.z
        self._own = []

        def keep(x):
            self._own.append(x)
            return x

        built = {}
        self.mats = []
        lo, hi = cut(nrows)
        for m in (r1cs.a, r1cs.b, r1cs.c):
            if id(m) not in built:
                rowptr, cols, vals = m.csr
                a, b = min(lo, m.nrows), min(hi, m.nrows)
                if b > a:
                    e0, e1 = int(rowptr[a]), int(rowptr[b])
                    built[id(m)] = keep(SparseMatrix.from_csr(rowptr[a:b + 1] - rowptr[a], cols[e0:e1], vals[e0:e1], b - a, nz))
                else:
                    built[id(m)] = None
            self.mats.append(built[id(m)])

        def idx(arr, rng):
            part = arr[rng[0]:rng[1]]
            return keep(IdxVec.from_host(part)) if len(part) else None

        def vec(arr, rng):
            part = arr[rng[0]:rng[1]]
            return keep(FrVec.from_host(np.ascontiguousarray(part))) if len(part) else None

        c_n = cut(nnz + 1)
        self.row_index, self.col_index = idx(row_index, c_n), idx(col_index, c_n)
        self.row = keep(_field_of_index(self.row_index)) if self.row_index else None
        self.col = keep(_field_of_index(self.col_index)) if self.col_index else None
        self.val_a, self.val_b, self.val_c = vec(val_a, c_n), vec(val_b, c_n), vec(val_c, c_n)
        self.ext_fre_row, self.ext_fre_col = idx(ext_row, cut(len(ext_row) + 2)), idx(ext_col, cut(len(ext_col) + 2))
        lo, hi = cut(self.w_len)
        w_cnt = max(min(hi, self.w_len) - lo, 0)
        self.w_block = None
        if w_cnt:
            self.w_block = keep(FrVec.alloc(w_cnt))
            capi.check(capi.load().gm_fr_stride(C.c_uint64(r1cs.w.handle), C.c_size_t(lo), C.c_size_t(1), C.c_size_t(w_cnt), C.c_uint64(self.w_block.handle)))

    @classmethod
    def dummy(cls, e_canonical: int, n: int, tail_log: int = 10) -> "PsnarkShard":
        """this rank's blocks of dummy_r1cs(e, n) (src/circuit.rs:349-365: z = [e; n], w = [e; n - 1], A = B = C = diag(1 / e)) in closed form --
        nothing of size n is built on the host (the joint support is the diagonal: row = col = 0 .. n - 1, the three value vectors 1 / e, every
        index looked up twice in the extended frequencies): what lets 8 ranks set up `psnark -i 26 .. 28` without 8 whole host copies"""
        from .circuit import SparseMatrix
        from .fr import IdxVec
        from .psnark import _field_of_index

        self = cls.__new__(cls)
        rank, world, _ = collective.info()
        e = e_canonical % R_MOD
        em, im = fr_from_int(e), fr_from_int(pow(e, -1, R_MOD))
        nt = 1 << max(n - 1, 0).bit_length()
        self.num_constraints = self.num_variables = self.nnz = n
        self.ext_row_len, self.ext_col_len, self.w_len = nt + n, 2 * n, n - 1
        self.longest = max(self.ext_row_len + 2, self.ext_col_len + 2, n + 2, n + 1)
        self.block = psnark_shard_block(self.longest, world)
        self.tail_log = tail_log
        self.levels = {}
        self._own = []
        lib = capi.load()
        lib.gm_psnark_shard_level.restype = C.c_size_t

        def keep(x):
            self._own.append(x)
            return x

        def cut(length: int, whole: int):
            s_ = int(lib.gm_psnark_shard_level(C.c_size_t(length), C.c_size_t(self.block), C.c_size_t(tail_log), C.c_int(world)))
            self.levels[length] = s_
            b = self.block >> s_
            return min(rank * b, whole), min((rank + 1) * b, whole)

        lo, hi = cut(n, n)  # the row block of the diagonal matrix
        d = keep(SparseMatrix.from_csr(np.arange(hi - lo + 1, dtype=np.uint64), np.arange(lo, hi, dtype=np.uint32), np.tile(im, (hi - lo, 1)), hi - lo, n)) if hi > lo else None
        self.mats = [d, d, d]
        self.z = keep(FrVec.alloc(n))
        self.z.fill(em)
        lo, hi = cut(n + 1, n)  # the joint support and everything indexed by it
        self.row_index = keep(IdxVec.from_host(np.arange(lo, hi, dtype=np.uint32))) if hi > lo else None
        self.col_index = keep(IdxVec.from_host(np.arange(lo, hi, dtype=np.uint32))) if hi > lo else None
        self.row = keep(_field_of_index(self.row_index)) if self.row_index else None
        self.col = keep(_field_of_index(self.col_index)) if self.col_index else None

        def vals():
            if hi <= lo:
                return None
            v = keep(FrVec.alloc(hi - lo))
            v.fill(im)
            return v

        self.val_a, self.val_b, self.val_c = vals(), vals(), vals()

        def ext(whole: int):
            # extend_frequency(compute_frequency(set_len, 0 .. n - 1)): i < n twice, n <= i < set_len once
            a, b = cut(whole + 2, whole)
            if b <= a:
                return None
            j = np.arange(a, b, dtype=np.int64)
            return keep(IdxVec.from_host(np.where(j < 2 * n, j // 2, n + (j - 2 * n)).astype(np.uint32)))

        self.ext_fre_row, self.ext_fre_col = ext(self.ext_row_len), ext(self.ext_col_len)
        lo, hi = cut(self.w_len, self.w_len)
        self.w_block = None
        if hi > lo:
            self.w_block = keep(FrVec.alloc(hi - lo))
            self.w_block.fill(em)
        return self

    def record(self, key: PsnarkShardKey, index=None) -> _GmPsnarkShard:
        h = lambda x: x.handle if x is not None else 0  # noqa: E731
        S = _GmPsnarkShard()
        S.a, S.b, S.c = (h(m) for m in self.mats)
        S.z, S.w_block, S.w_len = self.z.handle, h(self.w_block), self.w_len
        S.row_index, S.col_index, S.row, S.col = h(self.row_index), h(self.col_index), h(self.row), h(self.col)
        S.val_a, S.val_b, S.val_c = h(self.val_a), h(self.val_b), h(self.val_c)
        S.ext_fre_row, S.ext_fre_col, S.ext_fre_row_len, S.ext_fre_col_len = h(self.ext_fre_row), h(self.ext_fre_col), self.ext_row_len, self.ext_col_len
        S.num_constraints, S.num_variables, S.nnz = self.num_constraints, self.num_variables, self.nnz
        S.block, S.tail_log = self.block, self.tail_log
        S.key, S.key_segments, S.key_len = key.bases.handle, key.segments, key.n_key
        S.key_offsets = C.cast(key.offsets, C.POINTER(C.c_size_t))
        S.key_counts = C.cast(key.counts, C.POINTER(C.c_size_t))
        self._g2buf = (C.c_uint8 * len(key.g2_bytes)).from_buffer_copy(key.g2_bytes)
        S.ck_g2_bytes, S.ck_g2_len = C.cast(self._g2buf, C.POINTER(C.c_uint8)), len(key.g2_bytes)
        if index is not None:
            self._idx = np.ascontiguousarray(np.stack(index), dtype=np.uint64)
            S.index_commitments = self._idx.ctypes.data_as(C.POINTER(C.c_uint64))
        return S

    def index(self, key: PsnarkShardKey) -> list:
        """psnark::Proof::index (src/psnark/time_prover.rs:49-64) over the blocks: gm_psnark_index_sharded"""
        out = np.zeros((5, 18), dtype=np.uint64)
        S = self.record(key)
        capi.check(capi.load().gm_psnark_index_sharded(C.byref(S), capi.ptr(out)))
        return [out[k].copy() for k in range(5)]

    def free(self):
        for x in self._own:
            x.free()
        self._own = []


def psnark_new_time_sharded(shard: PsnarkShard, key: PsnarkShardKey, index: list):
    """gm_psnark_new_time_sharded; the same `psnark.Proof` on every rank"""
    from .psnark import _PSNARK_SPANS, _proof_buffers, _unpack_proof
    from .transcript import default_group_encoding

    P, cap, bufs = _proof_buffers(shard.num_variables, shard.nnz)
    S = shard.record(key, index)
    capi.check(capi.load().gm_psnark_new_time_sharded(C.byref(S), C.c_int(int(default_group_encoding())), C.c_size_t(cap), C.byref(P)))
    proof = _unpack_proof(P, bufs)
    proof.spans = {name: P.spans[i] for i, name in enumerate(_PSNARK_SPANS)}
    return proof

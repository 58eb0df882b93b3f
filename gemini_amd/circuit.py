"""R1CS instances for the time prover (src/circuit.rs): sparse matrices resident in HBM as CSR,
plus their transposes (what `abc_tensored` needs, src/snark/time_prover.rs:63-81)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .fr import FrVec, R_MOD, fr_from_int


class SparseMatrix:
    def __init__(self, handle: int, nrows: int, ncols: int):
        self.handle, self.nrows, self.ncols = handle, nrows, ncols
        self.csr = None

    @classmethod
    def from_csr(cls, rowptr, cols, vals_mont, nrows: int, ncols: int) -> "SparseMatrix":
        capi.ensure_init()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        vals = capi.u64(vals_mont).reshape(-1, 4)
        h = C.c_uint64()
        capi.check(capi.load().gm_spm_register(capi.ptr(rowptr), capi.ptr(cols), capi.ptr(vals), C.c_size_t(nrows), C.c_size_t(ncols),
                                               C.c_size_t(len(cols)), C.byref(h)))
        m = cls(h.value, nrows, ncols)
        m.csr = (rowptr, cols, vals)  # host copy (the preprocessing SNARK's joint matrices are built from it)
        return m

    @classmethod
    def from_rows(cls, rows, ncols: int, transpose: bool = False) -> "SparseMatrix":
        """rows: list of [(value_mont(4,), col)] -- `Matrix<F>` of src/circuit.rs:43"""
        nrows = len(rows)
        trip = [(i, c, v) for i, row in enumerate(rows) for (v, c) in row]
        if transpose:
            trip = sorted(((c, i, v) for (i, c, v) in trip), key=lambda t: (t[0], t[1]))
            nrows, ncols = ncols, nrows
        rowptr = np.zeros(nrows + 1, dtype=np.uint64)
        for i, _, _ in trip:
            rowptr[i + 1] += 1
        rowptr = np.cumsum(rowptr).astype(np.uint64)
        cols = np.array([c for _, c, _ in trip], dtype=np.uint32)
        vals = np.stack([np.asarray(v, dtype=np.uint64).reshape(4) for _, _, v in trip]) if trip else np.empty((0, 4), dtype=np.uint64)
        return cls.from_csr(rowptr, cols, vals, nrows, ncols)

    def mul(self, x: FrVec) -> FrVec:
        """product_matrix_vector (src/misc.rs:100-110)"""
        y = FrVec.alloc(self.nrows)
        capi.check(capi.load().gm_spm_mul(C.c_uint64(self.handle), C.c_uint64(x.handle), C.c_uint64(y.handle)))
        return y

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_spm_free(C.c_uint64(self.handle)))
            self.handle = 0


class R1cs:
    """src/circuit.rs R1cs {a, b, c, z, w, x} with everything the prover touches on the device."""

    def __init__(self, a, b, c, at, bt, ct, z: FrVec, w: FrVec, x: FrVec):
        self.a, self.b, self.c = a, b, c
        self.at, self.bt, self.ct = at, bt, ct
        self.z, self.w, self.x = z, w, x

    def free(self):
        for v in getattr(self, "_device_cache", {}).values():  # instance-only vectors of the preprocessing prover (psnark.py)
            for x in (v if isinstance(v, (list, tuple)) else [v]):
                if hasattr(x, "free"):
                    x.free()
        self._device_cache = {}
        seen = set()
        for m in (self.a, self.b, self.c, self.at, self.bt, self.ct):
            if id(m) not in seen:
                seen.add(id(m))
                m.free()
        for v in (self.z, self.w, self.x):
            v.free()


def dummy_r1cs(e_canonical: int, n: int) -> R1cs:
    """src/circuit.rs:349-365 with the random element passed in: z = [e; n], w = [e; n-1], x = [e],
    A = B = C = diag(e^-1).  The three matrices are one diagonal CSR (its own transpose)."""
    e = e_canonical % R_MOD
    inv_e = pow(e, -1, R_MOD)
    em, im = fr_from_int(e), fr_from_int(inv_e)
    rowptr = np.arange(n + 1, dtype=np.uint64)
    cols = np.arange(n, dtype=np.uint32)
    vals = np.tile(im, (n, 1))
    d = SparseMatrix.from_csr(rowptr, cols, vals, n, n)
    z = FrVec.alloc(n)
    z.fill(em)
    w = FrVec.alloc(n - 1)
    w.fill(em)
    x = FrVec.alloc(1)
    x.fill(em)
    return R1cs(d, d, d, d, d, d, z, w, x)


class R1csStream:
    """src/circuit.rs R1csStream: the same instance as big-endian streams (Reverse(..) of z, witness,
    z_a, z_b, z_c: src/snark/tests.rs:38-52) plus the matrices the MatrixTensor streams walk.  On the
    device the streams are reversed vectors and `MatrixTensor` is a product with the transposed CSR."""

    def __init__(self, r1cs: R1cs):
        from .fr import reverse

        self.r1cs = r1cs  # the matrices (the preprocessing SNARK walks their joint support)
        self.at, self.bt, self.ct = r1cs.at, r1cs.bt, r1cs.ct
        z_a, z_b, z_c = r1cs.a.mul(r1cs.z), r1cs.b.mul(r1cs.z), r1cs.c.mul(r1cs.z)
        self.z = reverse(r1cs.z)
        self.witness = reverse(r1cs.w)
        self.z_a, self.z_b, self.z_c = reverse(z_a), reverse(z_b), reverse(z_c)
        for v in (z_a, z_b, z_c):
            v.free()

    def free(self):
        for v in (self.z, self.witness, self.z_a, self.z_b, self.z_c):
            v.free()

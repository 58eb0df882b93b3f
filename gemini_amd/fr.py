"""Device-resident Fr vectors and the dense vector helpers of src/misc.rs, as calls into
libgemini_hip.so.  Elements are (…, 4) uint64 Montgomery limbs = ark-ff's `Fr` memory image.

The few scalar conversions done here in Python integers (a challenge to/from Montgomery form,
adding two scalars for HashMapPippenger keys) are host bookkeeping on O(1) values; every O(n)
pass runs on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_MONT_R = (1 << 256) % R_MOD
_MONT_RINV = pow(_MONT_R, -1, R_MOD)
_M64 = (1 << 64) - 1


def _to_int(l4) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l4))


def _to_limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def fr_from_int(v: int) -> np.ndarray:
    """canonical integer -> Montgomery limbs (`Fr::from`)."""
    return _to_limbs((v % R_MOD) * _MONT_R % R_MOD)


def fr_to_int(mont) -> int:
    """Montgomery limbs -> canonical integer (`into_bigint`)."""
    return _to_int(np.asarray(mont).reshape(4)) * _MONT_RINV % R_MOD


def fr_add_host(a, b) -> np.ndarray:
    return _to_limbs((_to_int(a) + _to_int(b)) % R_MOD)


def fr_into_bigint(mont: np.ndarray) -> np.ndarray:
    """`into_bigint` for an array of scalars (host, used only by the streaming wrappers that hand
    BigInts to msm_bigint exactly like the reference does)."""
    m = capi.u64(mont).reshape(-1, 4)
    return np.stack([_to_limbs(fr_to_int(x)) for x in m]) if len(m) else np.empty((0, 4), dtype=np.uint64)


class FrVec:
    """`Vec<F>` living in HBM."""

    def __init__(self, handle: int, cap: int):
        self.handle = handle
        self.cap = cap

    @classmethod
    def alloc(cls, n: int) -> "FrVec":
        capi.ensure_init()
        h = C.c_uint64()
        capi.check(capi.load().gm_fr_vec_alloc(C.c_size_t(n), C.byref(h)))
        return cls(h.value, n)

    @classmethod
    def from_host(cls, mont: np.ndarray) -> "FrVec":
        m = capi.u64(mont).reshape(-1, 4)
        v = cls.alloc(len(m))
        capi.check(capi.load().gm_fr_vec_upload(C.c_uint64(v.handle), C.c_size_t(0), capi.ptr(m), C.c_size_t(len(m))))
        return v

    def to_host(self) -> np.ndarray:
        n = len(self)
        out = np.empty((n, 4), dtype=np.uint64)
        capi.check(capi.load().gm_fr_vec_download(C.c_uint64(self.handle), C.c_size_t(0), capi.ptr(out), C.c_size_t(n)))
        return out

    def __len__(self) -> int:
        n = C.c_size_t()
        capi.check(capi.load().gm_fr_vec_len(C.c_uint64(self.handle), C.byref(n)))
        return n.value

    def set_len(self, n: int):
        capi.check(capi.load().gm_fr_vec_set_len(C.c_uint64(self.handle), C.c_size_t(n)))

    def fill(self, value_mont):
        capi.check(capi.load().gm_fr_vec_fill(C.c_uint64(self.handle), capi.ptr(capi.u64(value_mont).reshape(4))))

    def device_ptr(self) -> int:
        p = C.c_void_p()
        capi.check(capi.load().gm_fr_vec_ptr(C.c_uint64(self.handle), C.byref(p)))
        return p.value or 0

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_fr_vec_free(C.c_uint64(self.handle)))
            self.handle = 0


def _as_vec(x):
    return (x, False) if isinstance(x, FrVec) else (FrVec.from_host(x), True)


def reverse(v) -> FrVec:
    """Reverse(slice): big-endian stream <-> little-endian coefficient vector (src/iterable/slice.rs:17-39)"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(vv))
    try:
        capi.check(capi.load().gm_fr_reverse(C.c_uint64(vv.handle), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def stride(v, start: int, step: int, count: int) -> FrVec:
    """out[k] = v[start + k * step], k < count"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(count)
    try:
        capi.check(capi.load().gm_fr_stride(C.c_uint64(vv.handle), C.c_size_t(start), C.c_size_t(step), C.c_size_t(count), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def fold_polynomial(f, r_mont) -> FrVec:
    """src/misc.rs:52-56"""
    fv, tmp = _as_vec(f)
    out = FrVec.alloc((len(fv) + 1) // 2)
    capi.check(capi.load().gm_fr_fold(C.c_uint64(fv.handle), capi.ptr(capi.u64(r_mont).reshape(4)), C.c_uint64(out.handle)))
    if tmp:
        fv.free()
    return out


def powers(x_mont, n: int) -> FrVec:
    """src/misc.rs:59-65"""
    out = FrVec.alloc(n)
    capi.check(capi.load().gm_fr_powers(capi.ptr(capi.u64(x_mont).reshape(4)), C.c_size_t(n), C.c_uint64(out.handle)))
    return out


def tensor(elements_mont) -> FrVec:
    """src/misc.rs:133-149 (asserts at least one element)"""
    e = capi.u64(elements_mont).reshape(-1, 4)
    assert len(e) > 0
    out = FrVec.alloc(1 << len(e))
    capi.check(capi.load().gm_fr_tensor(capi.ptr(e), C.c_size_t(len(e)), C.c_uint64(out.handle)))
    return out


def hadamard(a, b) -> FrVec:
    """src/misc.rs:205-208 (panics on length mismatch -> GM_EINVAL)"""
    av, ta = _as_vec(a)
    bv, tb = _as_vec(b)
    out = FrVec.alloc(len(av))
    try:
        capi.check(capi.load().gm_fr_hadamard(C.c_uint64(av.handle), C.c_uint64(bv.handle), C.c_uint64(out.handle)))
    finally:
        if ta:
            av.free()
        if tb:
            bv.free()
    return out


def ip(a, b) -> np.ndarray:
    """src/misc.rs:215-218"""
    av, ta = _as_vec(a)
    bv, tb = _as_vec(b)
    out = np.empty(4, dtype=np.uint64)
    try:
        capi.check(capi.load().gm_fr_ip(C.c_uint64(av.handle), C.c_uint64(bv.handle), capi.ptr(out)))
    finally:
        if ta:
            av.free()
        if tb:
            bv.free()
    return out


def evaluate_le(poly, xs_mont) -> np.ndarray:
    """src/misc.rs:194-199 at 1..3 points in one pass; returns (npoints, 4)."""
    pv, tmp = _as_vec(poly)
    xs = capi.u64(xs_mont).reshape(-1, 4)
    out = np.empty((len(xs), 4), dtype=np.uint64)
    try:
        capi.check(capi.load().gm_fr_eval_le(C.c_uint64(pv.handle), capi.ptr(xs), C.c_size_t(len(xs)), capi.ptr(out)))
    finally:
        if tmp:
            pv.free()
    return out


def evaluate_le_batch(polys, xs_mont) -> np.ndarray:
    """evaluate_le of k device vectors at the same 1..3 points with one wait for all; returns (k, npoints, 4)"""
    xs = capi.u64(xs_mont).reshape(-1, 4)
    out = np.empty((len(polys), len(xs), 4), dtype=np.uint64)
    if len(polys):
        h = np.array([p.handle for p in polys], dtype=np.uint64)
        capi.check(capi.load().gm_fr_eval_le_batch(capi.ptr(h), C.c_size_t(len(polys)), capi.ptr(xs), C.c_size_t(len(xs)), capi.ptr(out)))
    return out


def linear_combination(polys, challenges_mont) -> FrVec:
    """src/misc.rs:37-48 (zip of polynomials and challenges; trailing zeros trimmed)"""
    ch = capi.u64(challenges_mont).reshape(-1, 4)
    k = min(len(polys), len(ch))
    vecs = [_as_vec(p) for p in polys[:k]]
    n = max([len(v) for v, _ in vecs], default=0)
    out = FrVec.alloc(n)
    handles = np.array([v.handle for v, _ in vecs], dtype=np.uint64)
    try:
        capi.check(capi.load().gm_fr_lincomb(capi.ptr(handles), capi.ptr(ch), C.c_size_t(k), C.c_uint64(out.handle)))
    finally:
        for v, t in vecs:
            if t:
                v.free()
    return out


def div_vanishing(f, points_mont):
    """quotient of f by prod (x - p_j) (DensePolynomial::div in src/kzg/time.rs:134-145)."""
    fv, tmp = _as_vec(f)
    pts = capi.u64(points_mont).reshape(-1, 4)
    q = FrVec.alloc(max(len(fv) - 1, 0))
    rem = np.zeros((len(pts), 4), dtype=np.uint64)
    try:
        capi.check(capi.load().gm_fr_div_vanishing(C.c_uint64(fv.handle), capi.ptr(pts), C.c_size_t(len(pts)), C.c_uint64(q.handle), capi.ptr(rem)))
    finally:
        if tmp:
            fv.free()
    return q, rem


# ---- index vectors + entry-product / plookup builders (psnark) ------------------------------------
class IdxVec:
    """`&[usize]` resident in HBM (row_index, col_index, extended frequencies)"""

    def __init__(self, handle: int, n: int):
        self.handle, self.n = handle, n

    @classmethod
    def from_host(cls, index) -> "IdxVec":
        capi.ensure_init()
        ix = np.ascontiguousarray(index, dtype=np.uint32)
        h = C.c_uint64()
        capi.check(capi.load().gm_idx_register(capi.ptr(ix), C.c_size_t(len(ix)), C.byref(h)))
        return cls(h.value, len(ix))

    def __len__(self) -> int:
        return self.n

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_idx_free(C.c_uint64(self.handle)))
            self.handle = 0


def lookup(v, index: IdxVec) -> FrVec:
    """src/subprotocols/plookup/time_prover.rs:5-8: [v[i] for i in index]"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(index))
    try:
        capi.check(capi.load().gm_fr_gather(C.c_uint64(vv.handle), C.c_uint64(index.handle), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def alg_hash(v, index, chal_mont) -> FrVec:
    """plookup/time_prover.rs:11-21; index None = the range 0..len(v)"""
    vv, tmp = _as_vec(v)
    n = len(vv) if index is None else min(len(vv), len(index))
    out = FrVec.alloc(n)
    try:
        capi.check(capi.load().gm_fr_alg_hash(C.c_uint64(vv.handle), C.c_uint64(0 if index is None else index.handle),
                                              capi.ptr(capi.u64(chal_mont).reshape(4)), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def plookup_set(v, y_mont, z_mont) -> FrVec:
    """plookup/time_prover.rs:23-35"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(vv) + 1 if len(vv) else 0)
    try:
        capi.check(capi.load().gm_fr_plookup_set(C.c_uint64(vv.handle), capi.ptr(capi.u64(y_mont).reshape(4)),
                                                 capi.ptr(capi.u64(z_mont).reshape(4)), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def plookup_subset(v, y_mont) -> FrVec:
    """plookup/time_prover.rs:62-64"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(vv))
    try:
        capi.check(capi.load().gm_fr_add_scalar(C.c_uint64(vv.handle), capi.ptr(capi.u64(y_mont).reshape(4)), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def shift_monic(v) -> FrVec:
    """right_rotation(monic(v)) = [1, v...] (src/subprotocols/entryproduct/time_prover.rs:14-23,47-51)"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(vv) + 1)
    try:
        capi.check(capi.load().gm_fr_shift_monic(C.c_uint64(vv.handle), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def accumulated_product_monic(v) -> FrVec:
    """accumulated_product(monic(v)) (entryproduct/time_prover.rs:25-51): out[i] = prod_{j>=i} v[j], out[len] = 1"""
    vv, tmp = _as_vec(v)
    out = FrVec.alloc(len(vv) + 1)
    try:
        capi.check(capi.load().gm_fr_acc_product(C.c_uint64(vv.handle), C.c_uint64(out.handle)))
    finally:
        if tmp:
            vv.free()
    return out


def element(v: FrVec, i: int) -> np.ndarray:
    """v[i] as a (4,) Montgomery array"""
    out = np.empty((1, 4), dtype=np.uint64)
    capi.check(capi.load().gm_fr_vec_download(C.c_uint64(v.handle), C.c_size_t(i), capi.ptr(out), C.c_size_t(1)))
    return out[0]

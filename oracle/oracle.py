"""ctypes binding of oracle/libgemini_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All field elements cross as numpy uint64 arrays: Fr = 4 limbs, Fq = 6 limbs, Montgomery form
unless a function says "canonical"; affine points = 12 limbs (x, y; (0,0) = identity);
Jacobian points = 18 limbs (X, Y, Z; Z = 0 identity).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgemini_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gemini_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgemini_oracle.so"])
    return _SO


_SO_NATIVE = os.path.join(_HERE, "libgemini_oracle_native.so")
_libs = {}
_which = "portable"


def _open(path):
    l = C.CDLL(path)
    l.go_msm_window.restype = C.c_size_t
    l.go_msm_window.argtypes = [C.c_size_t]
    l.go_fold_polynomial.restype = C.c_size_t
    l.go_linear_combination.restype = C.c_size_t
    l.go_sumcheck_rounds.restype = C.c_size_t
    l.go_sumcheck_rounds.argtypes = [C.c_size_t, C.c_size_t]
    l.go_g1_is_on_curve.restype = C.c_int
    l.go_g1_jac_eq.restype = C.c_int
    return l


def lib():
    """the C restatement every binding below calls: the portable build (the checker) unless a `native()` block is open"""
    if _which not in _libs:
        path = _SO if _which == "portable" else _SO_NATIVE
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s", os.path.basename(path)])
        _libs[_which] = _open(path)
    return _libs[_which]


def native_available() -> bool:
    """libgemini_oracle_native.so needs BMI2 + ADX on the host (every x86-64 server since 2015; checked, not assumed)"""
    try:
        flags = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except Exception:  # noqa: BLE001
        return False
    return {"bmi2", "adx", "avx2"} <= set(flags)


class native:
    """`with oracle.native():` -- the same source built for today's CPUs (oracle/Makefile: x86-64-v3 + ADX, the Fq product in
    mulx / adcx / adox asm), for the TIMED cpu_baseline legs of bench.py only.  The checker stays the portable build."""

    def __enter__(self):
        global _which
        self._prev = _which
        if native_available():
            _which = "native"
        return self

    def __exit__(self, *exc):
        global _which
        _which = self._prev
        return False


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- int <-> limb conversions (host-side convenience) -------------------------------------
def ints_to_limbs(vals, nlimbs: int) -> np.ndarray:
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        for k in range(nlimbs):
            out[i, k] = (v >> (64 * k)) & mask
    return out


def limbs_to_ints(a: np.ndarray):
    a = np.asarray(a, dtype=np.uint64)
    a2 = a.reshape(-1, a.shape[-1])
    return [sum(int(a2[i, k]) << (64 * k) for k in range(a2.shape[1])) for i in range(a2.shape[0])]


# ---- field ----------------------------------------------------------------------------------
def fr_to_mont(canon: np.ndarray) -> np.ndarray:
    canon = _u64(canon).reshape(-1, 4)
    out = np.empty_like(canon)
    lib().go_fr_to_mont(_p(canon), _p(out), C.c_size_t(len(canon)))
    return out


def fr_from_mont(mont: np.ndarray) -> np.ndarray:
    mont = _u64(mont).reshape(-1, 4)
    out = np.empty_like(mont)
    lib().go_fr_from_mont(_p(mont), _p(out), C.c_size_t(len(mont)))
    return out


def fq_to_mont(canon: np.ndarray) -> np.ndarray:
    canon = _u64(canon).reshape(-1, 6)
    out = np.empty_like(canon)
    lib().go_fq_to_mont(_p(canon), _p(out), C.c_size_t(len(canon)))
    return out


def fq_from_mont(mont: np.ndarray) -> np.ndarray:
    mont = _u64(mont).reshape(-1, 6)
    out = np.empty_like(mont)
    lib().go_fq_from_mont(_p(mont), _p(out), C.c_size_t(len(mont)))
    return out


def fr_mul(a, b) -> np.ndarray:
    a = _u64(a).reshape(-1, 4)
    b = _u64(b).reshape(-1, 4)
    out = np.empty_like(a)
    lib().go_fr_mul(_p(a), _p(b), _p(out), C.c_size_t(len(a)))
    return out


def fq_mul(a, b) -> np.ndarray:
    a = _u64(a).reshape(-1, 6)
    b = _u64(b).reshape(-1, 6)
    out = np.empty_like(a)
    lib().go_fq_mul(_p(a), _p(b), _p(out), C.c_size_t(len(a)))
    return out


def fr_inv(a) -> np.ndarray:
    a = _u64(a).reshape(4)
    out = np.empty(4, dtype=np.uint64)
    lib().go_fr_inv(_p(a), _p(out))
    return out


def random_fr(seed: int, n: int) -> np.ndarray:
    """n canonical Fr scalars from SplitMix64(seed) (same stream as pyref.SplitMix64.fr)."""
    out = np.empty((n, 4), dtype=np.uint64)
    lib().go_random_fr(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


# ---- G1 -------------------------------------------------------------------------------------
def g1_generator() -> np.ndarray:
    out = np.empty(12, dtype=np.uint64)
    lib().go_g1_generator(_p(out))
    return out


def g1_fixed_base_mul(base_aff: np.ndarray, scalars_canon: np.ndarray) -> np.ndarray:
    base_aff = _u64(base_aff).reshape(12)
    scalars_canon = _u64(scalars_canon).reshape(-1, 4)
    out = np.empty((len(scalars_canon), 12), dtype=np.uint64)
    lib().go_g1_fixed_base_mul(_p(base_aff), _p(scalars_canon), C.c_size_t(len(scalars_canon)), _p(out))
    return out


def g1_to_affine(jac: np.ndarray) -> np.ndarray:
    jac = _u64(jac).reshape(18)
    out = np.empty(12, dtype=np.uint64)
    lib().go_g1_to_affine(_p(jac), _p(out))
    return out


def g1_add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = _u64(a).reshape(18)
    b = _u64(b).reshape(18)
    out = np.empty(18, dtype=np.uint64)
    lib().go_g1_add(_p(a), _p(b), _p(out))
    return out


def g1_mul(aff: np.ndarray, k_canon: np.ndarray) -> np.ndarray:
    aff = _u64(aff).reshape(12)
    k = _u64(k_canon).reshape(4)
    out = np.empty(18, dtype=np.uint64)
    lib().go_g1_mul(_p(aff), _p(k), _p(out))
    return out


def g1_is_on_curve(aff: np.ndarray) -> bool:
    return bool(lib().go_g1_is_on_curve(_p(_u64(aff).reshape(12))))


def g1_jac_eq(a: np.ndarray, b: np.ndarray) -> bool:
    return bool(lib().go_g1_jac_eq(_p(_u64(a).reshape(18)), _p(_u64(b).reshape(18))))


def affine_to_ints(aff: np.ndarray):
    """affine Montgomery limbs -> (x, y) canonical python ints or None for the identity."""
    aff = _u64(aff).reshape(2, 6)
    if not aff.any():
        return None
    x, y = limbs_to_ints(fq_from_mont(aff))
    return (x, y)


def ints_to_affine(P) -> np.ndarray:
    if P is None:
        return np.zeros(12, dtype=np.uint64)
    return fq_to_mont(ints_to_limbs([P[0], P[1]], 6)).reshape(12)


# ---- MSM ------------------------------------------------------------------------------------
def msm_window(n: int) -> int:
    return int(lib().go_msm_window(n))


def signed_digits(scalar_canon: np.ndarray, w: int, num_bits: int = 255) -> np.ndarray:
    count = (num_bits + w - 1) // w
    out = np.empty(count, dtype=np.int64)
    s = _u64(scalar_canon).reshape(4)
    lib().go_signed_digits(_p(s), C.c_size_t(w), C.c_size_t(num_bits), out.ctypes.data_as(C.c_void_p), C.c_size_t(count))
    return out


def msm_naive(bases: np.ndarray, scalars_canon: np.ndarray) -> np.ndarray:
    bases = _u64(bases).reshape(-1, 12)
    scalars_canon = _u64(scalars_canon).reshape(-1, 4)
    n = min(len(bases), len(scalars_canon))
    out = np.empty(18, dtype=np.uint64)
    lib().go_msm_naive(_p(bases), _p(scalars_canon), C.c_size_t(n), _p(out))
    return out


def msm_pippenger(bases: np.ndarray, scalars_canon: np.ndarray, threads: int = 0, c: int = 0) -> np.ndarray:
    """The reference algorithm (variable_base.rs:99-176); truncates to the shorter input like msm_unchecked."""
    bases = _u64(bases).reshape(-1, 12)
    scalars_canon = _u64(scalars_canon).reshape(-1, 4)
    n = min(len(bases), len(scalars_canon))
    out = np.empty(18, dtype=np.uint64)
    lib().go_msm_pippenger_c(_p(bases), _p(scalars_canon), C.c_size_t(n), _p(out), C.c_int(threads), C.c_size_t(c))
    return out


def chunked_pippenger(bases, scalars_canon, buf_size: int, threads: int = 0) -> np.ndarray:
    bases = _u64(bases).reshape(-1, 12)
    scalars_canon = _u64(scalars_canon).reshape(-1, 4)
    n = min(len(bases), len(scalars_canon))
    out = np.empty(18, dtype=np.uint64)
    lib().go_chunked_pippenger(_p(bases), _p(scalars_canon), C.c_size_t(n), C.c_size_t(buf_size), _p(out), C.c_int(threads))
    return out


def msm_chunks(bases_stream, scalars_stream_canon, threads: int = 0) -> np.ndarray:
    bases = _u64(bases_stream).reshape(-1, 12)
    sc = _u64(scalars_stream_canon).reshape(-1, 4)
    assert len(sc) <= len(bases)
    out = np.empty(18, dtype=np.uint64)
    lib().go_msm_chunks(_p(bases), C.c_size_t(len(bases)), _p(sc), C.c_size_t(len(sc)), _p(out), C.c_int(threads))
    return out


def hashmap_pippenger(bases, scalars_mont, cap: int, threads: int = 0) -> np.ndarray:
    bases = _u64(bases).reshape(-1, 12)
    sm = _u64(scalars_mont).reshape(-1, 4)
    n = min(len(bases), len(sm))
    out = np.empty(18, dtype=np.uint64)
    lib().go_hashmap_pippenger(_p(bases), _p(sm), C.c_size_t(n), C.c_size_t(cap), _p(out), C.c_int(threads))
    return out


# ---- field vectors (Montgomery Fr) -----------------------------------------------------------
def fold_polynomial(f, r) -> np.ndarray:
    f = _u64(f).reshape(-1, 4)
    r = _u64(r).reshape(4)
    out = np.empty(((len(f) + 1) // 2, 4), dtype=np.uint64)
    lib().go_fold_polynomial(_p(f), C.c_size_t(len(f)), _p(r), _p(out))
    return out


def powers(x, n: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().go_powers(_p(_u64(x).reshape(4)), C.c_size_t(n), _p(out))
    return out


def powers2(x, n: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().go_powers2(_p(_u64(x).reshape(4)), C.c_size_t(n), _p(out))
    return out


def tensor(elements) -> np.ndarray:
    e = _u64(elements).reshape(-1, 4)
    out = np.empty((1 << len(e), 4), dtype=np.uint64)
    lib().go_tensor(_p(e), C.c_size_t(len(e)), _p(out))
    return out


def evaluate_le(poly, x) -> np.ndarray:
    poly = _u64(poly).reshape(-1, 4)
    out = np.empty(4, dtype=np.uint64)
    lib().go_evaluate_le(_p(poly), C.c_size_t(len(poly)), _p(_u64(x).reshape(4)), _p(out))
    return out


def hadamard(a, b) -> np.ndarray:
    a = _u64(a).reshape(-1, 4)
    b = _u64(b).reshape(-1, 4)
    assert len(a) == len(b)
    out = np.empty_like(a)
    lib().go_hadamard(_p(a), _p(b), C.c_size_t(len(a)), _p(out))
    return out


def ip(a, b) -> np.ndarray:
    a = _u64(a).reshape(-1, 4)
    b = _u64(b).reshape(-1, 4)
    assert len(a) == len(b)
    out = np.empty(4, dtype=np.uint64)
    lib().go_ip(_p(a), _p(b), C.c_size_t(len(a)), _p(out))
    return out


def linear_combination(polys, challenges) -> np.ndarray:
    polys = [_u64(p).reshape(-1, 4) for p in polys]
    ch = _u64(challenges).reshape(-1, 4)
    k = min(len(polys), len(ch))
    if k == 0:
        return np.empty((0, 4), dtype=np.uint64)
    n = max(len(p) for p in polys[:k])
    out = np.empty((n, 4), dtype=np.uint64)
    ptrs = (C.c_void_p * k)(*[p.ctypes.data for p in polys[:k]])
    lens = (C.c_size_t * k)(*[len(p) for p in polys[:k]])
    m = lib().go_linear_combination(ptrs, lens, C.c_size_t(k), _p(ch), _p(out), C.c_size_t(n))
    return out[:m]


def poly_div_monic(f, z):
    f = _u64(f).reshape(-1, 4)
    z = _u64(z).reshape(-1, 4)
    d = len(z) - 1
    q = np.zeros((max(len(f) - d, 0), 4), dtype=np.uint64)
    rem = np.zeros((d, 4), dtype=np.uint64)
    lib().go_poly_div_monic(_p(f), C.c_size_t(len(f)), _p(z), C.c_size_t(d), _p(q), _p(rem))
    return q, rem


# ---- sumcheck ---------------------------------------------------------------------------------
class TimeProver:
    """src/subprotocols/sumcheck/time_prover.rs:42-137 on Montgomery Fr arrays (C restatement)."""

    def __init__(self, f, g, twist):
        self.f = _u64(f).reshape(-1, 4).copy()
        self.g = _u64(g).reshape(-1, 4).copy()
        self.nf = len(self.f)
        self.ng = len(self.g)
        self.twist = _u64(twist).reshape(4).copy()
        self.round = 0
        self.tot_rounds = int(lib().go_sumcheck_rounds(self.nf, self.ng))

    def fold(self, r):
        nf = C.c_size_t(self.nf)
        ng = C.c_size_t(self.ng)
        lib().go_sumcheck_fold(_p(self.f), C.byref(nf), _p(self.g), C.byref(ng), _p(self.twist), _p(_u64(r).reshape(4)))
        self.nf, self.ng = nf.value, ng.value

    def next_message(self, verifier_message=None):
        assert self.round <= self.tot_rounds
        if verifier_message is not None:
            self.fold(verifier_message)
        if self.round == self.tot_rounds:
            return None
        a = np.empty(4, dtype=np.uint64)
        b = np.empty(4, dtype=np.uint64)
        lib().go_sumcheck_message(_p(self.f), C.c_size_t(self.nf), _p(self.g), C.c_size_t(self.ng), _p(self.twist), _p(a), _p(b))
        self.round += 1
        return a, b

    def final_foldings(self):
        if self.round == self.tot_rounds:
            return self.f[0].copy(), self.g[0].copy()
        return None

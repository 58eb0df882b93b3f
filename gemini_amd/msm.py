"""Host mirror of the MSM surface Gemini uses from ark-ec 0.4.2:
`VariableBaseMSM::{msm, msm_unchecked, msm_bigint}`, `ChunkedPippenger`, `HashMapPippenger`
(in-tree statements: src/kzg/msm/variable_base.rs, src/kzg/msm/stream_pippenger.rs) and
`msm_chunks` (src/kzg/space.rs:22-55).  All arithmetic happens in libgemini_hip.so.

Data conventions (numpy uint64): affine bases (n, 12) Montgomery x||y with (0,0) = identity, or
(n, 13) with column 12 = ark-ec's `infinity` flag word (stride 104, the Rust layout); scalars
(n, 4): `BigInt<4>` canonical for msm_bigint, Montgomery `Fr` for msm / msm_unchecked; results
(18,) Jacobian X, Y, Z Montgomery, normalised.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .fr import fr_into_bigint


class G1Bases:
    """SRS resident in HBM (`powers_of_g`, src/kzg/time.rs:24-27)."""

    def __init__(self, handle: int, n: int):
        self.handle = handle
        self.n = n

    @classmethod
    def register(cls, bases: np.ndarray) -> "G1Bases":
        capi.ensure_init()
        bases = capi.u64(bases)
        assert bases.ndim == 2 and bases.shape[1] in (12, 13)
        h = C.c_uint64()
        capi.check(capi.load().gm_g1_bases_register(capi.ptr(bases), C.c_size_t(bases.shape[1] * 8), C.c_size_t(len(bases)), C.byref(h)))
        return cls(h.value, len(bases))

    @classmethod
    def fixed_base(cls, base_affine: np.ndarray, scalars_canonical: np.ndarray) -> "G1Bases":
        """[s_i * base] generated on device (FixedBase::msm + normalize_batch, src/kzg/time.rs:55-59)."""
        capi.ensure_init()
        base_affine = capi.u64(base_affine).reshape(12)
        sc = capi.u64(scalars_canonical).reshape(-1, 4)
        h = C.c_uint64()
        capi.check(capi.load().gm_g1_fixed_base_register(capi.ptr(base_affine), capi.ptr(sc), C.c_size_t(len(sc)), C.byref(h)))
        return cls(h.value, len(sc))

    @classmethod
    def srs(cls, base_affine: np.ndarray, tau_canonical: np.ndarray, n: int) -> "G1Bases":
        """powers_of_g[i] = tau^i * g (src/kzg/time.rs:51-59)."""
        capi.ensure_init()
        h = C.c_uint64()
        capi.check(capi.load().gm_g1_srs_register(capi.ptr(capi.u64(base_affine).reshape(12)), capi.ptr(capi.u64(tau_canonical).reshape(4)),
                                                  C.c_size_t(n), C.byref(h)))
        return cls(h.value, n)

    def precompute(self, c: int = 0):
        """build the fixed-base window tables (gm_g1_bases_precompute); setup, outside any prover timer"""
        capi.check(capi.load().gm_g1_bases_precompute(C.c_uint64(self.handle), C.c_int(c)))
        return self

    def download(self, offset: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n - offset if n is None else n
        out = np.empty((n, 12), dtype=np.uint64)
        capi.check(capi.load().gm_g1_bases_download(C.c_uint64(self.handle), C.c_size_t(offset), C.c_size_t(n), capi.ptr(out)))
        return out

    def msm_bigint(self, scalars: np.ndarray, offset: int = 0, reversed_: bool = False) -> np.ndarray:
        sc = capi.u64(scalars).reshape(-1, 4)
        out = np.empty(18, dtype=np.uint64)
        capi.check(capi.load().gm_g1_msm_h(C.c_uint64(self.handle), C.c_size_t(offset), C.c_int(int(reversed_)), capi.ptr(sc),
                                           C.c_size_t(len(sc)), capi.ptr(out)))
        return out

    def msm_device(self, d_scalars_ptr: int, n: int, mont: bool, offset: int = 0, reversed_: bool = False, partial: bool = False) -> np.ndarray:
        out = np.empty(18, dtype=np.uint64)
        fn = capi.load().gm_g1_msm_d_partial if partial else capi.load().gm_g1_msm_d
        capi.check(fn(C.c_uint64(self.handle), C.c_size_t(offset), C.c_int(int(reversed_)), C.c_void_p(d_scalars_ptr), C.c_int(int(mont)),
                      C.c_size_t(n), capi.ptr(out)))
        return out

    def msm_vec(self, vec, n: int | None = None, voffset: int = 0, offset: int = 0, reversed_: bool = False) -> np.ndarray:
        n = len(vec) - voffset if n is None else n
        out = np.empty(18, dtype=np.uint64)
        capi.check(capi.load().gm_g1_msm_v(C.c_uint64(self.handle), C.c_size_t(offset), C.c_int(int(reversed_)), C.c_uint64(vec.handle),
                                           C.c_size_t(voffset), C.c_size_t(n), capi.ptr(out)))
        return out

    def msm_vec_batch(self, vecs, ns, offset: int = 0, reversed_: bool = False) -> np.ndarray:
        """k MSMs, vector j's first ns[j] elements against bases[offset ...]; returns (k, 18)"""
        k = len(vecs)
        out = np.empty((k, 18), dtype=np.uint64)
        if k == 0:
            return out
        handles = np.array([v.handle for v in vecs], dtype=np.uint64)
        nn = np.array(ns, dtype=np.uintp)
        capi.check(capi.load().gm_g1_msm_v_batch(C.c_uint64(self.handle), C.c_size_t(offset), C.c_int(int(reversed_)), capi.ptr(handles),
                                                 nn.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(k), capi.ptr(out)))
        return out

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_g1_bases_free(C.c_uint64(self.handle)))
            self.handle = 0

    def __len__(self):
        return self.n


def g1_sum(points: np.ndarray) -> np.ndarray:
    """normalise(sum of Jacobian points) -- `result += chunk` / the EC-add after an all-gather."""
    pts = capi.u64(points).reshape(-1, 18)
    out = np.empty(18, dtype=np.uint64)
    capi.check(capi.load().gm_g1_sum(capi.ptr(pts), C.c_size_t(len(pts)), capi.ptr(out)))
    return out


G1_ZERO = None


def g1_zero() -> np.ndarray:
    return g1_sum(np.empty((0, 18), dtype=np.uint64))


class VariableBaseMSM:
    """ark_ec::VariableBaseMSM for G1Projective."""

    @staticmethod
    def msm_bigint(bases: np.ndarray, bigints: np.ndarray) -> np.ndarray:
        capi.ensure_init()
        bases = capi.u64(bases)
        sc = capi.u64(bigints).reshape(-1, 4)
        n = min(len(bases), len(sc))  # zip semantics of the reference
        out = np.empty(18, dtype=np.uint64)
        capi.check(capi.load().gm_g1_msm(capi.ptr(bases), C.c_size_t(bases.shape[1] * 8), capi.ptr(sc), C.c_size_t(n), capi.ptr(out)))
        return out

    @staticmethod
    def msm_unchecked(bases: np.ndarray, scalars_mont: np.ndarray) -> np.ndarray:
        """Silently truncates to the shorter input, like the reference (src/kzg/time.rs:82)."""
        from .fr import FrVec

        capi.ensure_init()
        bases = capi.u64(bases)
        sc = capi.u64(scalars_mont).reshape(-1, 4)
        n = min(len(bases), len(sc))
        if n == 0:
            return g1_zero()
        reg = G1Bases.register(bases[:n])
        vec = FrVec.from_host(sc[:n])
        try:
            return reg.msm_vec(vec)  # into_bigint happens on device
        finally:
            vec.free()
            reg.free()

    @staticmethod
    def msm(bases: np.ndarray, scalars_mont: np.ndarray):
        """Ok(result) or Err(min_len) on length mismatch -- returned as (result, None) / (None, min_len)."""
        bases = capi.u64(bases)
        sc = capi.u64(scalars_mont).reshape(-1, 4)
        if len(bases) != len(sc):
            return None, min(len(bases), len(sc))
        return VariableBaseMSM.msm_unchecked(bases, sc), None


class ChunkedPippenger:
    """src/kzg/msm/stream_pippenger.rs:209-271: buffer pairs, flush an MSM every buf_size pairs."""

    def __init__(self, max_msm_buffer: int):
        self.buf_size = max_msm_buffer
        self.scalars_buffer: list = []
        self.bases_buffer: list = []
        self.result = g1_zero()

    @classmethod
    def with_size(cls, buf_size: int) -> "ChunkedPippenger":
        return cls(buf_size)

    def _flush(self):
        part = VariableBaseMSM.msm_bigint(np.stack(self.bases_buffer), np.stack(self.scalars_buffer))
        self.result = g1_sum(np.stack([self.result, part]))
        self.scalars_buffer.clear()
        self.bases_buffer.clear()

    def add(self, base: np.ndarray, scalar_bigint: np.ndarray):
        self.scalars_buffer.append(capi.u64(scalar_bigint).reshape(4))
        self.bases_buffer.append(capi.u64(base).reshape(-1))
        if len(self.scalars_buffer) == self.buf_size:
            self._flush()

    def finalize(self) -> np.ndarray:
        if self.scalars_buffer:
            self._flush()
        return self.result


class HashMapPippenger:
    """src/kzg/msm/stream_pippenger.rs:143-206: equal bases have their scalars added in Fr first."""

    def __init__(self, max_msm_buffer: int):
        from .fr import fr_add_host

        self._fr_add = fr_add_host
        self.capacity = max_msm_buffer
        self.buffer: dict = {}
        self.result = g1_zero()

    def _flush(self):
        bases = np.stack([np.frombuffer(k, dtype=np.uint64) for k in self.buffer.keys()])
        scalars = np.stack(list(self.buffer.values()))
        part = VariableBaseMSM.msm_bigint(bases, fr_into_bigint(scalars))
        self.result = g1_sum(np.stack([self.result, part]))
        self.buffer.clear()

    def add(self, base: np.ndarray, scalar_mont: np.ndarray):
        key = capi.u64(base).reshape(-1).tobytes()
        s = capi.u64(scalar_mont).reshape(4)
        cur = self.buffer.get(key)
        self.buffer[key] = s.copy() if cur is None else self._fr_add(cur, s)
        if len(self.buffer) == self.capacity:
            self._flush()

    def finalize(self) -> np.ndarray:
        if self.buffer:
            self._flush()
        return self.result


def msm_chunks(bases_stream: np.ndarray, scalars_stream_mont: np.ndarray) -> np.ndarray:
    """src/kzg/space.rs:22-55: skip len(bases) - len(scalars) bases, then 2^20-pair MSMs, summed."""
    bases = capi.u64(bases_stream)
    sc = capi.u64(scalars_stream_mont).reshape(-1, 4)
    assert len(sc) <= len(bases)
    bases = bases[len(bases) - len(sc):]
    step = 1 << 20
    result = g1_zero()
    for off in range(0, len(sc), step):
        part, err = VariableBaseMSM.msm(bases[off:off + step], sc[off:off + step])
        assert err is None
        result = g1_sum(np.stack([result, part]))
    return result

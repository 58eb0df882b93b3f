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

    @classmethod
    def srs_segments(cls, base_affine: np.ndarray, tau_canonical: np.ndarray, starts, counts) -> "G1Bases":
        """segment s = tau^(starts[s] + i) * g, i < counts[s], back to back in one handle (gm_g1_srs_register_segments)"""
        capi.ensure_init()
        st = np.array(starts, dtype=np.uintp)
        ct = np.array(counts, dtype=np.uintp)
        h = C.c_uint64()
        capi.check(capi.load().gm_g1_srs_register_segments(capi.ptr(capi.u64(base_affine).reshape(12)), capi.ptr(capi.u64(tau_canonical).reshape(4)),
                                                           st.ctypes.data_as(C.POINTER(C.c_size_t)), ct.ctypes.data_as(C.POINTER(C.c_size_t)),
                                                           C.c_size_t(len(st)), C.byref(h)))
        return cls(h.value, int(ct.sum()))

    def table_info(self):
        """(window width, bytes) of the fixed-base tables of this handle; (0, 0) without tables"""
        c, b = C.c_int(0), C.c_size_t(0)
        capi.check(capi.load().gm_g1_bases_table_info(C.c_uint64(self.handle), C.byref(c), C.byref(b)))
        return c.value, b.value

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

    def msm_vec_batch(self, vecs, ns, offset: int = 0, reversed_: bool = False, partial: bool = False) -> np.ndarray:
        """k MSMs, vector j's first ns[j] elements against bases[offset ...]; returns (k, 18).  partial: un-normalised
        results (the per-rank shares of a sharded batch_commit)"""
        k = len(vecs)
        out = np.empty((k, 18), dtype=np.uint64)
        if k == 0:
            return out
        handles = np.array([v.handle for v in vecs], dtype=np.uint64)
        nn = np.array(ns, dtype=np.uintp)
        fn = capi.load().gm_g1_msm_v_batch_partial if partial else capi.load().gm_g1_msm_v_batch
        capi.check(fn(C.c_uint64(self.handle), C.c_size_t(offset), C.c_int(int(reversed_)), capi.ptr(handles),
                                                 nn.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(k), capi.ptr(out)))
        return out

    def msm_vec_batch_at(self, vecs, ns, offsets, reversed_: bool = False, partial: bool = False) -> np.ndarray:
        """k MSMs, vector j's first ns[j] elements against bases[offsets[j] ...] (gm_g1_msm_v_batch_at); returns (k, 18)"""
        k = len(vecs)
        out = np.empty((k, 18), dtype=np.uint64)
        if k == 0:
            return out
        handles = np.array([v.handle for v in vecs], dtype=np.uint64)
        nn = np.array(ns, dtype=np.uintp)
        off = np.array(offsets, dtype=np.uintp)
        capi.check(capi.load().gm_g1_msm_v_batch_at(C.c_uint64(self.handle), off.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_int(int(reversed_)),
                                                    capi.ptr(handles), nn.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(k),
                                                    C.c_int(int(partial)), capi.ptr(out)))
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


class _PinnedBlock:
    def __init__(self, nbytes: int):
        self.p = C.c_void_p()
        capi.check(capi.load().gm_host_alloc(C.c_size_t(nbytes), C.byref(self.p)))

    def __del__(self):
        try:
            if self.p:
                capi.load().gm_host_free(self.p)
                self.p = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint64) -> np.ndarray:
    """numpy array over page-locked host memory (gm_host_alloc): HostMsmStream copies it by DMA at the PCIe rate
    instead of through the runtime's staging buffer.  Freed when the array (and every view of it) is gone."""
    capi.ensure_init()
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    blk = _PinnedBlock(max(nbytes, 1))
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(blk.p.value)
    buf._gm_block = blk  # keeps the allocation alive as long as numpy holds the buffer
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


class HostMsmStream:
    """gm_g1_msm_stream_*: an MSM over HOST-resident pairs in bounded device memory -- two device slots of `chunk`
    pairs, the copy of chunk i + 1 under the kernels of chunk i.  The MI355X form of the reference's streaming
    MSMs (ChunkedPippenger, src/kzg/msm/stream_pippenger.rs:209-272; msm_chunks, src/kzg/space.rs:22-55) for keys
    and polynomials that do not fit (or are not wanted) in HBM.

    HostMsmStream(chunk, mont=False)                         pairs carry their bases (n x 12 or n x 13 uint64 records)
    HostMsmStream(chunk, bases=G1Bases, offset, reversed_)   scalars only, against registered bases"""

    def __init__(self, chunk: int, mont: bool = False, bases: "G1Bases | None" = None, offset: int = 0, reversed_: bool = False, base_words: int = 12):
        capi.ensure_init()
        h = C.c_uint64()
        self.bases = bases
        self.base_words = base_words
        if bases is None:
            capi.check(capi.load().gm_g1_msm_stream_new(C.c_size_t(chunk), C.c_size_t(base_words * 8), C.c_int(int(mont)), C.byref(h)))
        else:
            capi.check(capi.load().gm_g1_msm_stream_new_h(C.c_uint64(bases.handle), C.c_size_t(offset), C.c_int(int(reversed_)), C.c_size_t(chunk),
                                                          C.c_int(int(mont)), C.byref(h)))
        self.handle = h.value

    def add(self, bases, scalars):
        """push a block of pairs (bases: (n, base_words) uint64 or None for a stream over registered bases; scalars:
        (n, 4) uint64).  Blocks of several chunks overlap copy and compute; the arrays are free again on return."""
        sc = np.ascontiguousarray(capi.u64(scalars).reshape(-1, 4))
        n = len(sc)
        if self.bases is None:
            b = np.ascontiguousarray(capi.u64(bases).reshape(-1, self.base_words))
            assert len(b) == n, "one base per scalar"
            bp = capi.ptr(b)
        else:
            bp = None
        if n:
            capi.check(capi.load().gm_g1_msm_stream_add(C.c_uint64(self.handle), bp, capi.ptr(sc), C.c_size_t(n)))

    def finalize(self) -> np.ndarray:
        """the normalised sum of everything pushed; the stream starts over"""
        out = np.empty(18, dtype=np.uint64)
        capi.check(capi.load().gm_g1_msm_stream_finalize(C.c_uint64(self.handle), capi.ptr(out), None))
        return out

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_g1_msm_stream_free(C.c_uint64(self.handle)))
            self.handle = 0


class ChunkedPippenger:
    """src/kzg/msm/stream_pippenger.rs:209-271: buffer pairs, flush an MSM every buf_size pairs.  The buffer is the
    device slot of a HostMsmStream (chunk = buf_size); pairs added one at a time are staged on the host in blocks."""

    _STAGE = 1 << 14

    def __init__(self, max_msm_buffer: int):
        self.buf_size = max_msm_buffer
        self._stream = None
        self._bases = None
        self._scalars = np.empty((min(self._STAGE, max_msm_buffer), 4), dtype=np.uint64)
        self._fill = 0

    @classmethod
    def with_size(cls, buf_size: int) -> "ChunkedPippenger":
        return cls(buf_size)

    def _push(self):
        if self._fill:
            self._stream.add(self._bases[: self._fill], self._scalars[: self._fill])
            self._fill = 0

    def _open(self, words: int):
        if self._stream is None:
            self._stream = HostMsmStream(min(self.buf_size, 1 << 26), mont=False, base_words=words)
            self._bases = np.empty((len(self._scalars), words), dtype=np.uint64)

    def add(self, base: np.ndarray, scalar_bigint: np.ndarray):
        b = capi.u64(base).reshape(-1)
        self._open(len(b))
        self._bases[self._fill] = b
        self._scalars[self._fill] = capi.u64(scalar_bigint).reshape(4)
        self._fill += 1
        if self._fill == len(self._scalars):
            self._push()

    def add_pairs(self, bases: np.ndarray, scalars_bigint: np.ndarray):
        """a whole block at once (no per-pair Python work)"""
        bases = capi.u64(bases)
        self._open(bases.shape[1])
        self._push()
        self._stream.add(bases, scalars_bigint)

    def finalize(self) -> np.ndarray:
        if self._stream is None:
            return g1_zero()
        self._push()
        out = self._stream.finalize()
        self._stream.free()
        self._stream = None
        return out


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
    """src/kzg/space.rs:22-55: skip len(bases) - len(scalars) bases, then 2^20-pair MSMs, summed -- one
    HostMsmStream with 2^20-pair slots (the streams stay on the host)."""
    bases = capi.u64(bases_stream)
    sc = capi.u64(scalars_stream_mont).reshape(-1, 4)
    assert len(sc) <= len(bases)
    if len(sc) == 0:
        return g1_zero()
    st = HostMsmStream(1 << 20, mont=True, base_words=bases.shape[1])
    try:
        st.add(bases[len(bases) - len(sc):], sc)
        return st.finalize()
    finally:
        st.free()

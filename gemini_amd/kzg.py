"""Host mirror of the time-efficient KZG committer, src/kzg/time.rs: every commitment and opening
is one device MSM against the SRS resident in HBM; quotients are computed on device."""
from __future__ import annotations

import numpy as np

from . import capi
from .fr import FrVec, R_MOD, _as_vec, div_vanishing, fold_polynomial, fr_from_int, fr_to_int, linear_combination, powers, reverse
from .msm import g1_sum, g1_zero
from .msm import G1Bases

# BLS12-381 G1 generator, Montgomery limbs (the reference draws g = G1::rand(rng), src/kzg/time.rs:54)
_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
_GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def g1_generator_mont() -> np.ndarray:
    mont = lambda v: [(((v << 384) % _Q) >> (64 * i)) & (2**64 - 1) for i in range(6)]
    return np.array(mont(_GX) + mont(_GY), dtype=np.uint64)


def _warn_truncated(poly_len: int, key_len: int):
    """`msm_unchecked` pairs as many coefficients as there are powers and drops the rest without a word
    (src/kzg/time.rs:81-83) -- kept, because callers rely on it, but a commitment that ignores coefficients opens to
    nothing: e.g. examples/psnark.rs:76 builds its time-prover key with 2n + 1 powers while the accumulated products of
    the sorted vectors have 2n + 2 coefficients, and that proof does not verify."""
    if poly_len > key_len:
        import warnings

        warnings.warn(f"commit: polynomial of {poly_len} coefficients against {key_len} powers -- the top {poly_len - key_len} "
                      "are dropped as in the reference (src/kzg/time.rs:82); the commitment will not verify", RuntimeWarning, stacklevel=3)


class CommitterKey:
    """src/kzg/time.rs:24-27.  powers_of_g lives on the GPU; powers_of_g2 is only used by the
    verifier (out of scope) so just its length (max_eval_points + 1) is kept."""

    def __init__(self, powers_of_g: G1Bases, max_eval_points: int, powers_of_g2=None):
        self.powers_of_g = powers_of_g
        self._max_eval_points = max_eval_points
        self.powers_of_g2 = powers_of_g2  # affine G2 points (gemini_amd.g2), only absorbed into transcripts

    @classmethod
    def new(cls, max_degree: int, max_eval_points: int, tau_canonical: np.ndarray, g_affine: np.ndarray | None = None, g2_affine=None) -> "CommitterKey":
        """src/kzg/time.rs:49-72 with the trapdoor passed in (the reference draws tau, g and g2 from rng;
        ark_std::test_rng() is not reproducible without Rust): g, g2 default to the standard generators, or are the
        draws a reference run recorded (tools/refvectors)."""
        from . import g2 as G2

        g = g1_generator_mont() if g_affine is None else g_affine
        tau = sum(int(v) << (64 * i) for i, v in enumerate(np.asarray(tau_canonical, dtype=np.uint64).reshape(4)))
        g2 = G2.generator() if g2_affine is None else g2_affine
        powers_of_g2 = [G2.mul(g2, pow(tau, i, R_MOD)) for i in range(max_eval_points + 1)]  # :60-67
        return cls(G1Bases.srs(g, tau_canonical, max_degree + 1), max_eval_points, powers_of_g2)

    @classmethod
    def from_powers(cls, powers_of_g: np.ndarray, max_eval_points: int, powers_of_g2=None) -> "CommitterKey":
        """a key whose G1 powers come from the host (a deserialised `CommitterKey`, src/kzg/time.rs:24-27): uploaded once,
        and -- a key stays resident -- given its fixed-base tables when they fit (gm_g1_bases_precompute(handle, -1); a plain
        `G1Bases.register` builds none, it may serve one MSM only)"""
        reg = G1Bases.register(powers_of_g)
        reg.precompute(-1)
        return cls(reg, max_eval_points, powers_of_g2)

    def powers_of_g2_bytes(self) -> bytes:
        """serialize_uncompressed(&self.powers_of_g2), what `append_serializable(b"ck", ..)` absorbs"""
        from . import g2 as G2

        assert self.powers_of_g2 is not None, "this key was built without its G2 half"
        return G2.serialize_vec_uncompressed(self.powers_of_g2)

    def max_eval_points(self) -> int:  # :75-78
        return self._max_eval_points

    def num_powers(self) -> int:
        """len(powers_of_g) of the whole key (a sharded key holds only a slice of it)"""
        return len(self.powers_of_g)

    def commit(self, polynomial) -> np.ndarray:
        """:81-83  msm_unchecked(&powers_of_g, polynomial): truncates to the shorter side."""
        v, tmp = _as_vec(polynomial)
        try:
            n = min(len(v), len(self.powers_of_g))
            _warn_truncated(len(v), len(self.powers_of_g))
            return self.powers_of_g.msm_vec(v, n=n)
        finally:
            if tmp:
                v.free()

    def batch_commit(self, polynomials) -> list:
        """:98-107: one MSM per polynomial, issued as one pipelined batch (gm_g1_msm_v_batch)"""
        vecs = [_as_vec(p) for p in polynomials]
        try:
            nb = len(self.powers_of_g)
            for v, _ in vecs:
                _warn_truncated(len(v), nb)
            out = self.powers_of_g.msm_vec_batch([v for v, _ in vecs], [min(len(v), nb) for v, _ in vecs])
            return [out[j] for j in range(len(vecs))]
        finally:
            for v, tmp in vecs:
                if tmp:
                    v.free()

    def open(self, polynomial, evaluation_point_mont):
        """:112-131 -> (evaluation, proof).  The Horner quotient is the device linear-factor division."""
        v, tmp = _as_vec(polynomial)
        try:
            if len(v) == 0:
                return np.zeros(4, dtype=np.uint64), self.powers_of_g.msm_bigint(np.empty((0, 4), dtype=np.uint64))
            q, rem = div_vanishing(v, capi.u64(evaluation_point_mont).reshape(1, 4))
            proof = self.commit(q)
            q.free()
            return rem[0], proof
        finally:
            if tmp:
                v.free()

    def open_multi_points(self, polynomial, eval_points_mont) -> np.ndarray:
        """:134-145  commit(f / vanishing(eval_points))"""
        v, tmp = _as_vec(polynomial)
        try:
            q, _ = div_vanishing(v, eval_points_mont)
            proof = self.commit(q)
            q.free()
            return proof
        finally:
            if tmp:
                v.free()

    def batch_open_multi_points(self, polynomials, eval_points_mont, eval_chal_mont) -> np.ndarray:
        """:149-159"""
        pts = capi.u64(eval_points_mont).reshape(-1, 4)
        assert len(pts) < self._max_eval_points + 1
        etas = powers(eval_chal_mont, len(polynomials))
        batched = linear_combination(polynomials, etas.to_host())
        etas.free()
        try:
            return self.open_multi_points(batched, pts)
        finally:
            batched.free()


class FoldedPolynomialTree:
    """src/subprotocols/sumcheck/streams.rs:13-37: a big-endian coefficient stream plus folding challenges;
    level i (1..depth) is the stream folded with challenges[0..i)."""

    def __init__(self, coefficients_stream, challenges_mont):
        self.coefficients = coefficients_stream
        self.challenges = [capi.u64(c).reshape(4) for c in challenges_mont]

    def depth(self) -> int:
        return len(self.challenges)

    def __len__(self) -> int:
        return len(self.coefficients)


def _newton_to_monomial_be(rem_newton, points_int):
    """f mod prod (x - p_j) from the successive-division remainders r_j (Newton form), returned
    big-endian like the `state` deque of src/kzg/space.rs:145-163"""
    coeffs = [0] * len(points_int)
    basis = [1]  # prod_{t<j} (x - p_t), little-endian
    for j, r in enumerate(rem_newton):
        rj = fr_to_int(r)
        for d, b in enumerate(basis):
            coeffs[d] = (coeffs[d] + rj * b) % R_MOD
        nxt = [0] * (len(basis) + 1)
        for d, b in enumerate(basis):
            nxt[d] = (nxt[d] - points_int[j] * b) % R_MOD
            nxt[d + 1] = (nxt[d + 1] + b) % R_MOD
        basis = nxt
    return np.stack([fr_from_int(c) for c in reversed(coeffs)])


class CommitterKeyStream:
    """src/kzg/space.rs:59-69.  `powers_of_g` is the big-endian stream Reverse(time key's powers)
    (:287-296); in HBM the SRS stays in time order and the stream view is the `reversed` addressing
    of gm_g1_msm_*.  Polynomials are big-endian coefficient streams (numpy arrays or FrVec)."""

    # The reference flushes its Pippenger buffers every max_msm_buffer (/ depth) pairs to bound HOST memory
    # (src/kzg/space.rs:105,139,205); the sum does not depend on where it is cut.  Scalars and SRS are already
    # resident in HBM here, so `max_msm_buffer` is ADVISORY on the device path: cuts shorter than
    # `min_device_chunk` pairs are merged into one device MSM (2^20 / 26 = 40 k-pair flushes would each cost a
    # latency-bound launch chain: snark -i 26 elastic 4.4 s instead of 1.46 s).  Pass min_device_chunk=1 to cut
    # literally where the caller says (tests/test_gpu_snark.py does, at logn 22 with max_msm_buffer 2^20).
    # The floor is 2^26 pairs (2 GB of scalars, ~15 GB of sort workspace): an MSM gets cheaper per pair up to there
    # (wider windows, one launch chain) -- snark -i 26 elastic 0.84 s with a 2^22 floor, 0.72 s with 2^24, 0.67 s with 2^26.
    DEFAULT_MIN_DEVICE_CHUNK = 1 << 26

    def __init__(self, powers_of_g: G1Bases, max_eval_points: int, powers_of_g2=None, min_device_chunk: int = None):
        self.powers_of_g = powers_of_g
        self._max_eval_points = max_eval_points
        self.powers_of_g2 = powers_of_g2
        self.min_device_chunk = self.DEFAULT_MIN_DEVICE_CHUNK if min_device_chunk is None else int(min_device_chunk)

    @classmethod
    def from_committer_key(cls, ck: "CommitterKey", min_device_chunk: int = None) -> "CommitterKeyStream":
        return cls(ck.powers_of_g, ck.max_eval_points(), ck.powers_of_g2, min_device_chunk=min_device_chunk)

    def powers_of_g2_bytes(self) -> bytes:
        from . import g2 as G2

        assert self.powers_of_g2 is not None, "this key was built without its G2 half"
        return G2.serialize_vec_uncompressed(self.powers_of_g2)

    def _n(self) -> int:
        """number of powers of the (whole) key"""
        return len(self.powers_of_g)

    def as_committer_key(self, max_degree: int) -> "CommitterKey":
        """:77-92 (keeps the first max_degree powers; shares the resident SRS)"""
        assert max_degree <= self._n()
        return CommitterKey(self.powers_of_g, self._max_eval_points)

    def _msm_stream(self, scalars_stream: FrVec, first_stream_pos: int, chunk: int) -> np.ndarray:
        """sum over stream positions: pair k = (base_stream[first_stream_pos + k], scalars_stream[k]),
        flushed every `chunk` pairs like ChunkedPippenger / msm_chunks"""
        n = self._n()
        total = len(scalars_stream)
        chunk = max(chunk, self.min_device_chunk)
        result = g1_zero()
        for off in range(0, total, chunk):
            m = min(chunk, total - off)
            part = self.powers_of_g.msm_vec(scalars_stream, n=m, voffset=off, offset=n - 1 - (first_stream_pos + off), reversed_=True)
            result = g1_sum(np.stack([result, part]))
        return result

    def commit(self, polynomial_stream) -> np.ndarray:
        """:169-177 -> msm_chunks (:22-55): skip len(powers) - len(poly) bases, 2^20-pair MSMs, summed.
        A HOST-resident coefficient stream (numpy array) of 2^22 or more elements is not uploaded whole: it goes
        through two 2^20-element device slots (HostMsmStream), copy under compute."""
        if isinstance(polynomial_stream, np.ndarray) and type(self) is CommitterKeyStream:
            # normalise FIRST: flat (4n,) uint64 input is accepted everywhere else (FrVec.from_host, HostMsmStream.add)
            coeffs = capi.u64(polynomial_stream).reshape(-1, 4)
            if len(coeffs) >= (1 << 22):
                return self._commit_host_stream(coeffs)
        v, tmp = _as_vec(polynomial_stream)
        try:
            assert self._n() >= len(v)
            return self._msm_stream(v, self._n() - len(v), 1 << 20)
        finally:
            if tmp:
                v.free()

    def _commit_host_stream(self, coeffs_be: np.ndarray, chunk: int = 1 << 20) -> np.ndarray:
        from .msm import HostMsmStream

        coeffs_be = capi.u64(coeffs_be).reshape(-1, 4)
        n = self._n()
        assert n >= len(coeffs_be)
        # stream position k pairs with time-order power n - 1 - (n - len) - k = len - 1 - k: walk down from len - 1
        st = HostMsmStream(chunk, mont=True, bases=self.powers_of_g, offset=len(coeffs_be) - 1, reversed_=True)
        try:
            st.add(None, coeffs_be)
            return st.finalize()
        finally:
            st.free()

    def open(self, polynomial_stream, alpha_mont, max_msm_buffer: int):
        """:95-125 -> (evaluation, proof): pairs (base_i, previous_i), previous_{i+1} = previous_i*alpha + c_i"""
        v, tmp = _as_vec(polynomial_stream)
        le = reverse(v)
        try:
            q, rem = div_vanishing(le, capi.u64(alpha_mont).reshape(1, 4))
            # stream-order scalars: [0, q_{d-1}, ..., q_0]
            qs = FrVec.alloc(len(v))
            qs.fill(fr_from_int(0))
            if len(q):
                qr = reverse(q)
                import ctypes as C

                host = qr.to_host()
                capi.check(capi.load().gm_fr_vec_upload(C.c_uint64(qs.handle), C.c_size_t(1), capi.ptr(host), C.c_size_t(len(host))))
                qr.free()
            proof = self._msm_stream(qs, self._n() - len(v), max_msm_buffer)
            q.free()
            qs.free()
            return rem[0], proof
        finally:
            le.free()
            if tmp:
                v.free()

    def open_multi_points(self, polynomial_stream, points_mont, max_msm_buffer: int):
        """:128-166 -> (remainder (big-endian, len(points) values), proof)"""
        v, tmp = _as_vec(polynomial_stream)
        pts = capi.u64(points_mont).reshape(-1, 4)
        le = reverse(v)
        try:
            q, rem = div_vanishing(le, pts)
            remainder = _newton_to_monomial_be(rem, [fr_to_int(p) for p in pts])
            qs = reverse(q)
            proof = self._msm_stream(qs, self._n() - len(v) + len(pts), max_msm_buffer)
            q.free()
            qs.free()
            return remainder, proof
        finally:
            le.free()
            if tmp:
                v.free()

    def _foldings_le(self, polynomials: FoldedPolynomialTree):
        v, tmp = _as_vec(polynomials.coefficients)
        cur = reverse(v)
        if tmp:
            v.free()
        out = []
        first = cur
        for ch in polynomials.challenges:
            cur = fold_polynomial(cur, ch)
            out.append(cur)
        first.free()
        return out

    def commit_folding(self, polynomials: FoldedPolynomialTree, max_msm_buffer: int, levels=None) -> list:
        """:192-223: one ChunkedPippenger of size max_msm_buffer / depth per folding level.  `levels` (optional): the
        little-endian folding levels when the caller holds them already (they are left alone then).  When no level is
        cut -- every level fits the effective flush size -- the depth MSMs go through ONE pipelined batch call
        (gm_g1_msm_v_batch, like CommitterKey.batch_commit): the commitment of level j is sum_i level_j[i] * tau^i g
        whichever way the pairs are walked."""
        n = polynomials.depth()
        own = levels is None
        if own:
            levels = self._foldings_le(polynomials)
        chunk = max(max(1, max_msm_buffer // max(n, 1)), self.min_device_chunk)
        try:
            if type(self) is CommitterKeyStream and levels and all(len(l) <= chunk for l in levels):
                assert self._n() >= max(len(l) for l in levels)
                return list(self.powers_of_g.msm_vec_batch(levels, [len(l) for l in levels], offset=0, reversed_=False))
            out = []
            for lvl in levels:
                s = reverse(lvl)
                out.append(self._msm_stream(s, self._n() - len(s), max(1, max_msm_buffer // max(n, 1))))
                s.free()
            return out
        finally:
            if own:
                for lvl in levels:
                    lvl.free()

    def open_folding(self, polynomials: FoldedPolynomialTree, points_mont, etas_mont, max_msm_buffer: int, levels=None):
        """:229-285 -> (remainders per level (big-endian), proof = sum_i etas[i-1] * commit(quotient_i)).
        The reference merges equal bases in a HashMapPippenger; on the device that is the linear
        combination of the quotients followed by one (chunked) MSM.  `levels`: as in commit_folding (consumed here)."""
        pts = capi.u64(points_mont).reshape(-1, 4)
        etas = capi.u64(etas_mont).reshape(-1, 4)
        pts_int = [fr_to_int(p) for p in pts]
        if levels is None:
            levels = self._foldings_le(polynomials)
        quotients, remainders = [], []
        for lvl in levels:
            if len(lvl) > len(pts):
                q, rem = div_vanishing(lvl, pts)
                remainders.append(_newton_to_monomial_be(rem, pts_int))
            else:  # shorter than the divisor: quotient empty, remainder = the polynomial, zero padded at the top
                q = FrVec.alloc(0)
                host = lvl.to_host()
                pad = np.zeros((len(pts) - len(host), 4), dtype=np.uint64)
                remainders.append(np.concatenate([pad, host[::-1]]))
            quotients.append(q)
            lvl.free()
        batched = linear_combination(quotients, etas[: len(quotients)])
        for q in quotients:
            q.free()
        if len(batched) == 0:
            batched.free()
            return remainders, g1_zero()
        s = reverse(batched)
        # bases: tau^j for coefficient j -> stream position n - len(batched)
        proof = self._msm_stream(s, self._n() - len(s), max_msm_buffer)
        s.free()
        batched.free()
        return remainders, proof


class HostCommitterKeyStream:
    """A committer key whose powers stay in HOST memory, in stream order (`Reverse(powers_of_g)`: highest power first,
    src/kzg/space.rs:287-296) -- for keys that do not fit in HBM or are not worth registering.  commit() is msm_chunks
    (src/kzg/space.rs:22-55, 169-177) over a HostMsmStream: both streams pass through two device slots of `chunk` pairs
    (copy of chunk i + 1 under the kernels of chunk i), device memory is O(chunk), the result equals
    CommitterKey.commit of the same polynomial.  Page-locked arrays (gemini_amd.msm.pinned_empty) stream several times
    faster than pageable ones."""

    def __init__(self, powers_of_g_stream: np.ndarray, max_eval_points: int, chunk: int = 1 << 20):
        self.powers_of_g = capi.u64(powers_of_g_stream)
        self._max_eval_points = max_eval_points
        self.chunk = chunk

    def commit(self, polynomial_stream: np.ndarray) -> np.ndarray:
        from .msm import HostMsmStream

        sc = capi.u64(polynomial_stream).reshape(-1, 4)
        assert len(self.powers_of_g) >= len(sc)
        if len(sc) == 0:
            return g1_zero()
        st = HostMsmStream(self.chunk, mont=True, base_words=self.powers_of_g.shape[1])
        try:
            st.add(self.powers_of_g[len(self.powers_of_g) - len(sc):], sc)  # align the streams (:36-40)
            return st.finalize()
        finally:
            st.free()

"""Host mirror of the time-efficient KZG committer, src/kzg/time.rs: every commitment and opening
is one device MSM against the SRS resident in HBM; quotients are computed on device."""
from __future__ import annotations

import numpy as np

from . import capi
from .fr import FrVec, _as_vec, div_vanishing, fr_from_int, linear_combination, powers
from .msm import G1Bases

# BLS12-381 G1 generator, Montgomery limbs (the reference draws g = G1::rand(rng), src/kzg/time.rs:54)
_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
_GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def g1_generator_mont() -> np.ndarray:
    mont = lambda v: [(((v << 384) % _Q) >> (64 * i)) & (2**64 - 1) for i in range(6)]
    return np.array(mont(_GX) + mont(_GY), dtype=np.uint64)


class CommitterKey:
    """src/kzg/time.rs:24-27.  powers_of_g lives on the GPU; powers_of_g2 is only used by the
    verifier (out of scope) so just its length (max_eval_points + 1) is kept."""

    def __init__(self, powers_of_g: G1Bases, max_eval_points: int):
        self.powers_of_g = powers_of_g
        self._max_eval_points = max_eval_points

    @classmethod
    def new(cls, max_degree: int, max_eval_points: int, tau_canonical: np.ndarray, g_affine: np.ndarray | None = None) -> "CommitterKey":
        """src/kzg/time.rs:49-72 with the trapdoor passed in (the reference draws tau and g from rng;
        ark_std::test_rng() is not reproducible without Rust)."""
        g = g1_generator_mont() if g_affine is None else g_affine
        return cls(G1Bases.srs(g, tau_canonical, max_degree + 1), max_eval_points)

    def max_eval_points(self) -> int:  # :75-78
        return self._max_eval_points

    def commit(self, polynomial) -> np.ndarray:
        """:81-83  msm_unchecked(&powers_of_g, polynomial): truncates to the shorter side."""
        v, tmp = _as_vec(polynomial)
        try:
            n = min(len(v), len(self.powers_of_g))
            return self.powers_of_g.msm_vec(v, n=n)
        finally:
            if tmp:
                v.free()

    def batch_commit(self, polynomials) -> list:
        """:98-107 (sequential loop of MSMs, like the reference)"""
        return [self.commit(p) for p in polynomials]

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

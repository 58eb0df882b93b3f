"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def rand_bases(orc, seed: int, n: int) -> np.ndarray:
    """n affine points k_i * G (Montgomery, (n, 12)) from the C oracle's fixed-base routine."""
    ks = orc.random_fr(seed, n)
    return orc.g1_fixed_base_mul(orc.g1_generator(), ks)


def jac_to_affine_ints(orc, jac):
    return orc.affine_to_ints(orc.g1_to_affine(jac))


def assert_same_point(orc, got_jac, exp_jac):
    """bit-exact after normalisation: both sides reduced to canonical affine integers."""
    assert jac_to_affine_ints(orc, got_jac) == jac_to_affine_ints(orc, exp_jac)


def is_normalised(orc, jac) -> bool:
    jac = np.asarray(jac, dtype=np.uint64).reshape(3, 6)
    one = orc.fq_to_mont(orc.ints_to_limbs([1], 6))[0]
    if not jac[2].any():
        return bool((jac[0] == one).all() and (jac[1] == one).all())
    return bool((jac[2] == one).all())

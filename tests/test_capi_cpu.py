"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/gemini_hip.h
declares, reports errors without a GPU instead of falling back, and its host-only entry point
(gm_g1_sum) agrees with the oracle.  No device compute happens here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ROOT, "gemini_amd", "libgemini_hip.so")):
        ge.build()
    from gemini_amd import capi

    return capi.load()


def test_exports_match_header(lib):
    from gemini_amd import capi

    hdr = open(os.path.join(ROOT, "include", "gemini_hip.h")).read()
    declared = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.gm_abi_version() == 1


def test_no_silent_fallback_without_gpu(lib):
    """Without a device (this container) every compute entry must fail loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; this test is about the CPU-only box")
    from gemini_amd import capi

    with pytest.raises(capi.GeminiHipError):
        capi.init(0)
    out = np.zeros(18, dtype=np.uint64)
    rc = lib.gm_g1_msm(None, C.c_size_t(96), None, C.c_size_t(0), capi.ptr(out))
    assert rc == -2  # GM_ENOTINIT
    assert b"gm_init" in lib.gm_last_error()
    h = C.c_uint64()
    assert lib.gm_fr_vec_alloc(C.c_size_t(4), C.byref(h)) == -2
    assert lib.gm_sc_round(C.c_uint64(1), None, capi.ptr(out), capi.ptr(out), C.byref(C.c_int())) == -2
    # the entry points added for the preprocessing SNARK and the batched MSM
    idx = np.zeros(4, dtype=np.uint32)
    assert lib.gm_idx_register(capi.ptr(idx), C.c_size_t(4), C.byref(h)) == -2
    assert lib.gm_fr_acc_product(C.c_uint64(1), C.c_uint64(2)) == -2
    assert lib.gm_fr_gather(C.c_uint64(1), C.c_uint64(2), C.c_uint64(3)) == -2
    hs = np.zeros(2, dtype=np.uint64)
    ns = np.zeros(2, dtype=np.uintp)
    assert lib.gm_g1_msm_v_batch(C.c_uint64(1), C.c_size_t(0), C.c_int(0), capi.ptr(hs), ns.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(2),
                                 capi.ptr(np.zeros(36, dtype=np.uint64))) == -2
    assert lib.gm_set_msm_affine_levels(C.c_int(0)) == -2
    assert lib.gm_sp_new_v(C.c_uint64(1), C.c_uint64(2), capi.ptr(out), C.byref(h)) == -2


def test_g1_sum_host(lib, oracle, pyref):
    """gm_g1_sum is pure host code (the EC add after the all-gather): check vs the oracle."""
    from gemini_amd.msm import g1_sum

    ks = oracle.random_fr(5, 6)
    pts = oracle.g1_fixed_base_mul(oracle.g1_generator(), ks)
    jac = np.stack([oracle.g1_mul(p, oracle.ints_to_limbs([3 + i], 4)[0]) for i, p in enumerate(pts)])
    got = g1_sum(jac)
    total = sum(k * (3 + i) for i, k in enumerate(oracle.limbs_to_ints(ks))) % pyref.R_MOD
    assert oracle.affine_to_ints(oracle.g1_to_affine(got)) == pyref.g1_mul(pyref.G1_GEN, total)
    # normalised output: Z is the Montgomery one
    one = oracle.fq_to_mont(oracle.ints_to_limbs([1], 6))[0]
    assert (got[12:] == one).all()
    # identity handling: P + (-P), and the empty sum
    neg = jac[0].copy()
    y = oracle.limbs_to_ints(oracle.fq_from_mont(neg[6:12]))[0]
    neg[6:12] = oracle.fq_to_mont(oracle.ints_to_limbs([(-y) % pyref.Q_MOD], 6))[0]
    z = g1_sum(np.stack([jac[0], neg]))
    assert not z[12:].any() and (z[:6] == one).all() and (z[6:12] == one).all()
    assert not g1_sum(np.empty((0, 18), dtype=np.uint64))[12:].any()


def test_header_is_plain_c99(tmp_path):
    """the boundary is a C ABI: include/gemini_hip.h compiles as C99 with -pedantic and links against the library"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "gemini_hip.h"\nint main(void) { return gm_abi_version() == 1 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                           "-L", os.path.join(root, "gemini_amd"), "-lgemini_hip", "-Wl,-rpath," + os.path.join(root, "gemini_amd"), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0  # gm_abi_version needs no GPU

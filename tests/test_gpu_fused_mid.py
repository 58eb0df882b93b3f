"""The mid-size calls of a batch as the levels of ONE block-sorted pass (msm.hip: MsmMulti::block, LevelGeom, GM_MSM_FUSE_MID=1; off by
default because it measures neutral, profiles/r5_fused_mid_probe.txt): the same points as the calls apart, with the key's tables,
with a prefix table set, without tables, forwards and against a stream view (CommitterKey::batch_commit, src/kzg/time.rs:98-107)."""
import os

import numpy as np
import pytest

from tests.util import jac_to_affine_ints

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


@pytest.mark.parametrize("tables", [True, False])
def test_fused_mid_levels_same_commitments(gm, oracle, tables):
    import ctypes as C

    from gemini_amd.fr import FrVec
    from gemini_amd.kzg import g1_generator_mont
    from gemini_amd.msm import G1Bases

    lib = gm.capi.load()
    gm.capi.check(lib.gm_set_auto_tables(C.c_int(int(tables)), C.c_size_t(0)))
    try:
        rng = np.random.default_rng(77)
        tau = rng.integers(0, 2**62, size=4, dtype=np.uint64)
        key = G1Bases.srs(g1_generator_mont(), tau, (1 << 18) + 3)
        # a folding tree from 2^18 down to 2 with ragged lengths, plus two more mid-size calls: 2 tiny groups, mid groups
        sizes = [(1 << 18) - 1, (1 << 17) + 5, 1 << 16, (1 << 15) - 3, 1 << 14, 9000, 1 << 13, 4000, 1000, 17, 2, 1, 20000, 70000]
        vecs = [FrVec.from_host(rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)) for n in sizes]
        os.environ["GM_MSM_FUSE_MID"] = "0"
        apart = key.msm_vec_batch(vecs, sizes)
        os.environ["GM_MSM_FUSE_MID"] = "1"
        fused = key.msm_vec_batch(vecs, sizes)
        for a, f, n in zip(apart, fused, sizes):
            assert jac_to_affine_ints(oracle, a) == jac_to_affine_ints(oracle, f), n
        # and one call at a time
        for v, f in list(zip(vecs, fused))[:6]:
            assert jac_to_affine_ints(oracle, key.msm_vec(v)) == jac_to_affine_ints(oracle, f)
        for v in vecs:
            v.free()
        key.free()
    finally:
        os.environ.pop("GM_MSM_FUSE_MID", None)
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(0)))


def test_fused_mid_levels_in_a_whole_proof(gm, oracle):
    """snark -i 16 and the elastic prover (reversed walks of the stream view) with the fused pass on: the same proof bytes"""
    from gemini_amd import snark
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream

    n = 1 << 16
    r1cs = dummy_r1cs(424242, n)
    ck = CommitterKey.new(n, 3, oracle.ints_to_limbs([oracle.limbs_to_ints(oracle.random_fr(808, 1))[0]], 4)[0])
    stream = R1csStream(r1cs)
    try:
        os.environ["GM_MSM_FUSE_MID"] = "0"
        want = snark.Proof.new_time(r1cs, ck, native=True).serialize_compressed()
        os.environ["GM_MSM_FUSE_MID"] = "1"
        assert snark.Proof.new_time(r1cs, ck, native=True).serialize_compressed() == want
        assert snark.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 1 << 20, native=True).serialize_compressed() == want
    finally:
        os.environ.pop("GM_MSM_FUSE_MID", None)
        stream.free()
        r1cs.free()
        ck.powers_of_g.free()

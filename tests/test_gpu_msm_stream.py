"""GPU parity: the streaming MSM over HOST-resident pairs (gm_g1_msm_stream_*: the device form of ChunkedPippenger,
src/kzg/msm/stream_pippenger.rs:209-272, and msm_chunks, src/kzg/space.rs:22-55) vs the oracle and vs the one-call
MSM.  The sum must not depend on where the stream is cut or on how the blocks are pushed."""
import threading

import numpy as np
import pytest

from tests.util import assert_same_point, jac_to_affine_ints, rand_bases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _push_ragged(st, bases, sc, cuts):
    lo = 0
    for hi in list(cuts) + [len(sc)]:
        st.add(None if bases is None else bases[lo:hi], sc[lo:hi])
        lo = hi


@pytest.mark.parametrize("chunk", [1, 64, 1000, 1024, 4096, 1 << 20])
def test_stream_equals_oracle_wherever_it_is_cut(gm, oracle, chunk):
    from gemini_amd.msm import HostMsmStream

    n = 4113
    bases = rand_bases(oracle, 901, n)
    sc = oracle.random_fr(902, n)
    exp = oracle.msm_pippenger(bases, sc)
    st = HostMsmStream(chunk)
    try:
        # one block; ragged blocks (a lone pair, a block ending exactly on a slot boundary, empty blocks)
        if chunk >= 64:
            st.add(bases, sc)
            assert_same_point(oracle, st.finalize(), exp)
        _push_ragged(st, bases, sc, [1, 1, 8, 1024, 1024, 2048, 3000])
        got = st.finalize()
        assert_same_point(oracle, got, exp)
        assert (got == gm.VariableBaseMSM.msm_bigint(bases, sc)).all()  # same normalised limbs as the one-call MSM
        # the stream starts over after finalize; an empty stream is the identity
        assert jac_to_affine_ints(oracle, st.finalize()) is None
        st.add(bases[:10], sc[:10])
        assert_same_point(oracle, st.finalize(), oracle.msm_pippenger(bases[:10], sc[:10]))
    finally:
        st.free()


def test_stream_montgomery_scalars_and_flagged_records(gm, oracle, pyref):
    """scalars as ark-ff Fr (Montgomery) and 104-byte G1Affine records with the infinity flag at byte 96"""
    from gemini_amd.msm import HostMsmStream

    n = 700
    bases = rand_bases(oracle, 911, n)
    sc = oracle.random_fr(912, n)
    rec = np.zeros((n, 13), dtype=np.uint64)
    rec[:, :12] = bases
    rec[5, 12] = 1  # flagged identity: coordinates are ignored
    rec[6, :12] = 0  # all-zero record = identity as well
    eff = bases.copy()
    eff[5] = 0
    eff[6] = 0
    exp = oracle.msm_pippenger(eff, sc)
    st = HostMsmStream(256, mont=True, base_words=13)
    try:
        _push_ragged(st, rec, oracle.fr_to_mont(sc), [100, 356, 357])
        assert_same_point(oracle, st.finalize(), exp)
    finally:
        st.free()


@pytest.mark.parametrize("reversed_", [False, True])
def test_stream_over_registered_bases(gm, oracle, reversed_):
    """scalars only, against a resident key: forward from `offset`, or the big-endian view walking down from it"""
    from gemini_amd.msm import HostMsmStream

    n, m = 3000, 2500
    bases = rand_bases(oracle, 921, n)
    sc = oracle.random_fr(922, m)
    reg = gm.G1Bases.register(bases)
    offset = n - 7 if reversed_ else 13
    try:
        exp = reg.msm_bigint(sc, offset=offset, reversed_=reversed_)
        sel = bases[offset - m + 1: offset + 1][::-1] if reversed_ else bases[offset: offset + m]
        assert_same_point(oracle, exp, oracle.msm_pippenger(np.ascontiguousarray(sel), sc))
        st = HostMsmStream(512, bases=reg, offset=offset, reversed_=reversed_)
        try:
            for _ in range(2):  # finalize rewinds the base cursor
                _push_ragged(st, None, sc, [3, 512, 2000])
                assert (st.finalize() == exp).all()
            # running off the registered bases is an error, not a wrap-around
            with pytest.raises(gm.capi.GeminiHipError):
                st.add(None, oracle.random_fr(923, n))
                st.finalize()
            st.finalize()
        finally:
            st.free()
    finally:
        reg.free()


def test_stream_pinned_buffers_and_chunked_pippenger_blocks(gm, oracle):
    from gemini_amd.msm import HostMsmStream, pinned_empty

    n = 5000
    bases = rand_bases(oracle, 931, n)
    sc = oracle.random_fr(932, n)
    exp = oracle.msm_pippenger(bases, sc)
    pb = pinned_empty((n, 12))
    ps = pinned_empty((n, 4))
    pb[:] = bases
    ps[:] = sc
    st = HostMsmStream(1024)
    try:
        st.add(pb, ps)
        assert_same_point(oracle, st.finalize(), exp)
    finally:
        st.free()
    p = gm.ChunkedPippenger.with_size(777)
    p.add(bases[0], sc[0])
    p.add_pairs(bases[1:4000], sc[1:4000])
    for b, s in zip(bases[4000:], sc[4000:]):
        p.add(b, s)
    assert_same_point(oracle, p.finalize(), exp)
    assert jac_to_affine_ints(oracle, gm.ChunkedPippenger(8).finalize()) is None


def test_stream_rejects_scalars_outside_fr(gm, oracle):
    from gemini_amd.msm import HostMsmStream

    bases = rand_bases(oracle, 941, 300)
    sc = oracle.random_fr(942, 300)
    sc[123, 3] |= np.uint64(1 << 63)  # >= 2^255: not an Fr element
    st = HostMsmStream(128)
    try:
        with pytest.raises(gm.capi.GeminiHipError):
            st.add(bases, sc)
            st.finalize()
        st.finalize()  # the stream is usable again
        st.add(bases[:100], sc[:100])
        assert_same_point(oracle, st.finalize(), oracle.msm_pippenger(bases[:100], sc[:100]))
    finally:
        st.free()


def test_two_streams_from_two_threads(gm, oracle):
    from gemini_amd.msm import HostMsmStream

    n = 6000
    data = [(rand_bases(oracle, 951 + t, n), oracle.random_fr(961 + t, n)) for t in range(2)]
    exp = [oracle.msm_pippenger(b, s) for b, s in data]
    out = [None, None]

    def run(t):
        st = HostMsmStream(512)
        try:
            for _ in range(3):
                _push_ragged(st, data[t][0], data[t][1], [100, 2000, 2001])
                out[t] = st.finalize()
        finally:
            st.free()

    th = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for t in range(2):
        assert_same_point(oracle, out[t], exp[t])


def test_stream_full_size_property(gm, oracle):
    """2^22 + 5 pairs through 2^20-pair slots (five flushes, copy under compute) == the one-call MSM of the same pairs,
    both with the bases in the stream and against the resident key (src/kzg/space.rs:41-53 composition)"""
    from gemini_amd.kzg import g1_generator_mont
    from gemini_amd.msm import HostMsmStream

    n = (1 << 22) + 5
    rng = np.random.default_rng(77)
    ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    ks[:, 3] &= np.uint64((1 << 60) - 1)
    reg = gm.G1Bases.fixed_base(g1_generator_mont(), ks)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) & np.uint64((1 << 62) - 1)
    try:
        exp = reg.msm_bigint(sc)
        host_bases = reg.download()
        st = HostMsmStream(1 << 20)
        try:
            st.add(host_bases, sc)
            assert (st.finalize() == exp).all()
        finally:
            st.free()
        st = HostMsmStream(1 << 20, bases=reg)
        try:
            st.add(None, sc)
            assert (st.finalize() == exp).all()
        finally:
            st.free()
        assert (gm.msm_chunks(host_bases, oracle.fr_to_mont(sc)) == exp).all()
        # the one-shot entry with host pointers (gm_g1_msm) streams from 2^22 pairs on
        assert (gm.VariableBaseMSM.msm_bigint(host_bases, sc) == exp).all()
    finally:
        reg.free()


def test_host_resident_key_and_polynomial_streams(gm, oracle):
    """time == space commitment (src/kzg/tests.rs:16-29) with the streams in HOST memory: a host-resident key in stream
    order (HostCommitterKeyStream) and a host coefficient stream against the resident key (CommitterKeyStream.commit of a
    numpy array >= 2^22 elements goes through the device slots instead of being uploaded whole)"""
    from gemini_amd.fr import powers, fr_from_int
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream, HostCommitterKeyStream

    for m, key_len in (((1 << 22) + 3, (1 << 22) + 10), (5000, 6000)):
        ck = CommitterKey.new(key_len, 3, oracle.random_fr(2301, 1)[0])
        poly = powers(fr_from_int(oracle.limbs_to_ints(oracle.random_fr(2302, 1))[0]), m)
        exp = ck.commit(poly)
        be = np.ascontiguousarray(poly.to_host()[::-1])  # big-endian coefficient stream, on the host
        poly.free()
        assert (CommitterKeyStream.from_committer_key(ck).commit(be) == exp).all()
        # flat (4n,) uint64 input is the same polynomial (accepted by FrVec.from_host / HostMsmStream.add as well)
        assert (CommitterKeyStream.from_committer_key(ck).commit(be.reshape(-1)) == exp).all()
        key_be = np.ascontiguousarray(ck.powers_of_g.download()[::-1])  # Reverse(powers_of_g), on the host
        assert (HostCommitterKeyStream(key_be, 3, chunk=1 << 19).commit(be) == exp).all()
        ck.powers_of_g.free()

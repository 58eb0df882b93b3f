"""GPU parity: G1 MSM through the C ABI vs the CPU restatement of the reference algorithm
(oracle/gemini_oracle.c, itself pinned in test_oracle_pin.py).  Bit-exact after normalisation."""
import ctypes as C

import numpy as np
import pytest

from tests.util import assert_same_point, dot_ints, is_normalised, jac_to_affine_ints, rand_bases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _set_window(gm, c):
    gm.capi.check(gm.capi.load().gm_set_msm_window(C.c_int(c)))


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 100, 257, 1000, 4113])
def test_msm_bigint_vs_oracle(gm, oracle, n):
    """src/kzg/msm/variable_base.rs:179-215 (Pippenger == naive) with the oracle as the naive side."""
    bases = rand_bases(oracle, 1000 + n, n)
    sc = oracle.random_fr(2000 + n, n)
    got = gm.VariableBaseMSM.msm_bigint(bases, sc)
    assert is_normalised(oracle, got)
    assert_same_point(oracle, got, oracle.msm_pippenger(bases, sc))


def test_msm_special_scalars_and_points(gm, oracle, pyref):
    n = 200
    bases = rand_bases(oracle, 7, n)
    sc = oracle.random_fr(8, n)
    r = pyref.R_MOD
    special = oracle.ints_to_limbs([0, 1, r - 1, 2, (1 << 254), (1 << 16) - 1, 1 << 16, (1 << 15), r - (1 << 15)], 4)
    sc[: len(special)] = special
    bases[20] = 0  # identity bases (CommitterKey::index_by creates them, src/kzg/time.rs:87)
    bases[21] = 0
    bases[31] = bases[30]  # equal points in different slots
    neg = oracle.affine_to_ints(bases[40])
    bases[41] = oracle.ints_to_affine((neg[0], (-neg[1]) % pyref.Q_MOD))  # P and -P
    sc[41] = sc[40]
    got = gm.VariableBaseMSM.msm_bigint(bases, sc)
    assert_same_point(oracle, got, oracle.msm_pippenger(bases, sc))
    # all-zero scalars and all-identity bases give the identity, encoded (R, R, 0)
    z = gm.VariableBaseMSM.msm_bigint(bases, np.zeros_like(sc))
    assert jac_to_affine_ints(oracle, z) is None and is_normalised(oracle, z)
    z = gm.VariableBaseMSM.msm_bigint(np.zeros_like(bases), sc)
    assert jac_to_affine_ints(oracle, z) is None
    # empty input
    z = gm.VariableBaseMSM.msm_bigint(bases[:0], sc[:0])
    assert jac_to_affine_ints(oracle, z) is None


@pytest.mark.parametrize("n", [64, 1000, 5000])
def test_msm_all_equal_scalars(gm, oracle, n):
    """dummy_r1cs makes every witness scalar equal (src/circuit.rs:349-365): one bucket per window."""
    bases = rand_bases(oracle, 11, n)
    e = oracle.random_fr(12, 1)[0]
    sc = np.tile(e, (n, 1))
    assert_same_point(oracle, gm.VariableBaseMSM.msm_bigint(bases, sc), oracle.msm_pippenger(bases, sc))


@pytest.mark.parametrize("n", [64, 1000, 5000])
def test_msm_all_equal_bases(gm, oracle, pyref, n):
    """the elastic example makes every base the generator (examples/snark.rs:59-63): P + P inside buckets."""
    g = oracle.g1_generator()
    bases = np.tile(g, (n, 1))
    sc = oracle.random_fr(13, n)
    got = gm.VariableBaseMSM.msm_bigint(bases, sc)
    total = sum(oracle.limbs_to_ints(sc)) % pyref.R_MOD
    assert jac_to_affine_ints(oracle, got) == pyref.g1_mul(pyref.G1_GEN, total)
    # and both degeneracies at once
    e = oracle.random_fr(14, 1)[0]
    got = gm.VariableBaseMSM.msm_bigint(bases, np.tile(e, (n, 1)))
    assert jac_to_affine_ints(oracle, got) == pyref.g1_mul(pyref.G1_GEN, oracle.limbs_to_ints(e)[0] * n % pyref.R_MOD)


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16, 17, 19, 20, 21])
def test_msm_window_independence(gm, oracle, c):
    """the result is a group element: it cannot depend on the window width"""
    n = 700
    bases = rand_bases(oracle, 21, n)
    sc = oracle.random_fr(22, n)
    sc[0] = oracle.ints_to_limbs([0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000000], 4)[0]
    exp = oracle.msm_pippenger(bases, sc)
    try:
        _set_window(gm, c)
        assert_same_point(oracle, gm.VariableBaseMSM.msm_bigint(bases, sc), exp)
    finally:
        _set_window(gm, 0)


def test_msm_rust_layout_and_unchecked(gm, oracle):
    """stride-104 `G1Affine` records with the infinity flag; msm / msm_unchecked length semantics."""
    n = 300
    bases = rand_bases(oracle, 31, n)
    sc = oracle.random_fr(32, n)
    rust = np.zeros((n, 13), dtype=np.uint64)
    rust[:, :12] = bases
    rust[5, 12] = 1  # infinity = true, coordinates are then meaningless
    rust[5, :12] = 0xDEADBEEF
    ref = bases.copy()
    ref[5] = 0
    exp = oracle.msm_pippenger(ref, sc)
    assert_same_point(oracle, gm.VariableBaseMSM.msm_bigint(rust, sc), exp)
    mont = oracle.fr_to_mont(sc)
    assert_same_point(oracle, gm.VariableBaseMSM.msm_unchecked(rust, mont), exp)
    # msm_unchecked truncates to the shorter side; msm reports Err(min_len)
    assert_same_point(oracle, gm.VariableBaseMSM.msm_unchecked(rust, mont[:250]), oracle.msm_pippenger(ref[:250], sc[:250]))
    res, err = gm.VariableBaseMSM.msm(rust, mont[:250])
    assert res is None and err == 250
    res, err = gm.VariableBaseMSM.msm(rust, mont)
    assert err is None
    assert_same_point(oracle, res, exp)


def test_registered_bases_offset_and_reverse(gm, oracle):
    """CommitterKey::commit uses a prefix of powers_of_g (src/kzg/time.rs:81-83); CommitterKeyStream
    walks Reverse(powers_of_g) after advance_by (src/kzg/space.rs:36-40,291-296)."""
    n = 500
    bases = rand_bases(oracle, 41, n)
    sc = oracle.random_fr(42, 200)
    reg = gm.G1Bases.register(bases)
    try:
        assert (reg.download() == bases).all()
        assert_same_point(oracle, reg.msm_bigint(sc), oracle.msm_pippenger(bases[:200], sc))
        assert_same_point(oracle, reg.msm_bigint(sc, offset=123), oracle.msm_pippenger(bases[123:323], sc))
        rev = bases[::-1].copy()
        skip = n - 200
        # stream element k <-> array index n-1-(skip+k)
        assert_same_point(oracle, reg.msm_bigint(sc, offset=n - 1 - skip, reversed_=True), oracle.msm_pippenger(rev[skip:], sc))
        with pytest.raises(gm.capi.GeminiHipError):
            reg.msm_bigint(sc, offset=400)
    finally:
        reg.free()


def test_chunked_and_hashmap_pippenger(gm, oracle, pyref):
    """time == space commitments (src/kzg/tests.rs:16-29) at the MSM level; stream_pippenger.rs:366-418"""
    n = 150
    bases = rand_bases(oracle, 51, n)
    sc = oracle.random_fr(52, n)
    full = oracle.msm_pippenger(bases, sc)
    for buf in [1 << 20, 64, 7]:
        p = gm.ChunkedPippenger(buf)
        for b, s in zip(bases, sc):
            p.add(b, s)
        assert_same_point(oracle, p.finalize(), full)
    dup = np.concatenate([bases, bases[:60]])
    sc2 = np.concatenate([sc, oracle.random_fr(53, 60)])
    mont = oracle.fr_to_mont(sc2)
    exp = oracle.hashmap_pippenger(dup, mont, 1 << 20)
    for cap in [1 << 20, 100, 16]:
        p = gm.HashMapPippenger(cap)
        for b, s in zip(dup, mont):
            p.add(b, s)
        assert_same_point(oracle, p.finalize(), exp)
    # msm_chunks alignment (src/kzg/space.rs:36-40)
    got = gm.msm_chunks(bases, oracle.fr_to_mont(sc[:100]))
    assert_same_point(oracle, got, oracle.msm_chunks(bases, sc[:100]))


def test_fixed_base_and_srs_generation(gm, oracle, pyref):
    """CommitterKey::new's powers_of_g (src/kzg/time.rs:51-59) built on device"""
    g = oracle.g1_generator()
    ks = oracle.random_fr(61, 300)
    ks[0] = 0
    ks[1] = oracle.ints_to_limbs([1], 4)[0]
    reg = gm.G1Bases.fixed_base(g, ks)
    try:
        assert (reg.download() == oracle.g1_fixed_base_mul(g, ks)).all()
    finally:
        reg.free()
    tau = oracle.random_fr(62, 1)[0]
    srs = gm.G1Bases.srs(g, tau, 257)
    try:
        t = oracle.limbs_to_ints(tau)[0]
        pw = oracle.ints_to_limbs([pow(t, i, pyref.R_MOD) for i in range(257)], 4)
        assert (srs.download() == oracle.g1_fixed_base_mul(g, pw)).all()
    finally:
        srs.free()


def test_msm_2_16_vs_oracle(gm, oracle):
    n = 1 << 16
    bases = rand_bases(oracle, 71, n)
    sc = oracle.random_fr(72, n)
    assert_same_point(oracle, gm.VariableBaseMSM.msm_bigint(bases, sc), oracle.msm_pippenger(bases, sc))


def test_msm_2_23_vs_oracle(gm, oracle):
    """2^23 - 3 pairs against the CPU Pippenger on the same inputs (about 10 s of CPU on the GPU box), both ways: the default
    for a resident key (tables) and the plain wide-window configuration (c = 19, 14 windows, 3.7 M buckets)"""
    import bench

    n = (1 << 23) - 3
    rng = np.random.default_rng(2323)
    reg = gm.G1Bases.fixed_base(oracle.g1_generator(), bench.uniform_fr(rng, n))
    try:
        sc = bench.uniform_fr(rng, n)
        exp = oracle.msm_pippenger(reg.download(), sc)
        # the library default for a resident key of this size: fixed-base tables (c = 20, 13 windows, one shared bucket set)
        assert reg.table_info()[0] == 20  # (22 from 2^23 points on)
        assert_same_point(oracle, reg.msm_bigint(sc), exp)
        # and the plain path of the same size
        gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1 << 62)))
        assert_same_point(oracle, reg.msm_bigint(sc), exp)
    finally:
        gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1 << 17)))
        reg.free()


def test_prefix_tables_serve_the_short_calls_of_a_big_key(gm, oracle):
    """A key of >= 2^23 points gets c = 22 tables, which lose below 2^22 pairs; calls of 2^17 .. 2^22 - 1 pairs that stay inside the
    first 2^22 points (the low levels of a folding tree) take a c = 20 table over that prefix.  Same group elements as the plain
    path, wherever the range lies: inside the prefix, ending at its last point, crossing it (plain path), walked backwards."""
    import bench
    from gemini_amd.fr import FrVec

    lib = gm.capi.load()
    n = 1 << 23
    rng = np.random.default_rng(2223)
    reg = gm.G1Bases.fixed_base(oracle.g1_generator(), bench.uniform_fr(rng, n))
    try:
        assert reg.table_info() == (22, (12 * n + 13 * (1 << 22) + 16 * (1 << 17)) * 96)
        P = 1 << 22
        S = 1 << 17  # ... and calls of 2^11 .. 2^17 - 1 pairs inside the first 2^17 points a c = 16 table (one bucket set, 16 final doublings)
        cases = [((1 << 17) + 5, 0, False), (1 << 18, 12345, False), (1 << 19, P - (1 << 19), False), (1 << 19, P - (1 << 19) + 1, False),
                 ((1 << 17) + 9, 1 << 20, True), (1 << 21, P - 1, True), (1 << 18, P, True), ((1 << 22) - 1, 0, False), ((1 << 17) - 1, 0, False),
                 (1 << 11, 0, False), ((1 << 11) - 1, 0, False), ((1 << 13) + 1, 777, False), (1 << 16, S - (1 << 16), False), (1 << 16, S - (1 << 16) + 1, False),
                 (1 << 14, S - 1, True), (1 << 14, S, True), (3000, 2999, True)]
        scs = [bench.uniform_fr(rng, m) for m, _, _ in cases]
        got = [reg.msm_bigint(sc, offset=o, reversed_=r) for sc, (m, o, r) in zip(scs, cases)]
        levels = [FrVec.from_host(oracle.fr_to_mont(bench.uniform_fr(rng, 1 << k))) for k in range(21, 9, -1)]
        got_batch = reg.msm_vec_batch(levels, [len(v) for v in levels])
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 62)))  # no tables of any kind
        for sc, (m, o, r), g in zip(scs, cases, got):
            assert (reg.msm_bigint(sc, offset=o, reversed_=r) == g).all(), (m, o, r)
        assert (reg.msm_vec_batch(levels, [len(v) for v in levels]) == got_batch).all()
        # what the library does when a device allocation fails twice: the prefix tables go, the main table stays, same results
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 17)))
        gm.capi.check(lib.gm_g1_release_spare_tables())
        assert reg.table_info() == (22, 12 * n * 96)
        for sc, (m, o, r), g in list(zip(scs, cases, got))[:4]:
            assert (reg.msm_bigint(sc, offset=o, reversed_=r) == g).all(), (m, o, r)
        for v in levels:
            v.free()
        # and one of them against the CPU Pippenger
        m, o, r = cases[0]
        assert_same_point(oracle, got[0], oracle.msm_pippenger(reg.download(o, m), scs[0]))
    finally:
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 17)))
        reg.free()


def test_table_sets_randomised_differential(gm, oracle):
    """Random (pairs, offset, direction) calls against a key with every kind of table set (main c = 22, prefix c = 20, small c = 16)
    and against the same key with no tables at all: 60 calls, sizes log-uniform in 2^9 .. 2^22.3, ranges that hug, touch and cross the
    prefix ends, bit-identical results; scalars uniform, sparse or all equal (the reference's own benchmark instance has equal ones)."""
    import bench

    lib = gm.capi.load()
    n = (1 << 23) + 11
    rng = np.random.default_rng(777)
    reg = gm.G1Bases.fixed_base(oracle.g1_generator(), bench.uniform_fr(rng, n))
    pool = bench.uniform_fr(rng, 1 << 23)
    try:
        assert reg.table_info()[0] == 22
        edges = [0, 1 << 17, 1 << 22, n]
        cases = []
        for it in range(60):
            m = int(2 ** rng.uniform(9, 22.3))
            e = edges[int(rng.integers(0, 4))]
            rev = bool(rng.integers(0, 2))
            slack = int(rng.integers(-3, 4)) if it % 2 else int(rng.integers(0, 1 << 16))
            if rev:  # indices o, o - 1, ..., o - m + 1
                o = min(max(e - 1 + slack, m - 1), n - 1) if it % 3 else int(rng.integers(m - 1, n))
            else:  # indices o .. o + m - 1
                o = min(max(e - m + slack, 0), n - m) if it % 3 else int(rng.integers(0, n - m + 1))
            cases.append((m, o, rev, it % 4))
        def scalars(m, kind, it):
            sc = pool[it * 4099 % (1 << 20):][:m].copy()
            if kind == 1:
                sc[::3] = 0
                sc[1::3, 1:] = 0
            elif kind == 2:
                sc[:] = sc[0]
            return sc
        got = [reg.msm_bigint(scalars(m, k, i), offset=o, reversed_=r) for i, (m, o, r, k) in enumerate(cases)]
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 62)))
        for i, ((m, o, r, k), g) in enumerate(zip(cases, got)):
            assert (reg.msm_bigint(scalars(m, k, i), offset=o, reversed_=r) == g).all(), (m, o, r, k)
    finally:
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 17)))
        reg.free()


def test_prefix_tables_of_a_key_too_long_for_tables_of_its_own(gm, oracle):
    """2^26 points and more cannot have whole-key tables (26-bit pair index in a table entry), but the folding levels of a proof --
    half of its pairs -- walk the first powers: such a key gets c = 22 over its first 2^25 points, c = 20 over the first 2^22 and
    c = 16 over the first 2^17.  Same group elements as the plain path inside, at the edge of and across each prefix."""
    import bench

    lib = gm.capi.load()
    n = (1 << 26) + 5
    rng = np.random.default_rng(2626)
    reg = gm.G1Bases.fixed_base(oracle.g1_generator(), bench.uniform_fr(rng, n))
    try:
        assert reg.table_info() == (0, (12 * (1 << 25) + 13 * (1 << 22) + 16 * (1 << 17)) * 96)
        Q = 1 << 25
        cases = [(1 << 22, 0, False), ((1 << 22) + 3, Q - (1 << 22) - 3, False), (1 << 22, Q - (1 << 22) + 1, False), (1 << 23, Q - 1, True),
                 (1 << 19, 5, False), (1 << 13, 1 << 16, True), (1 << 22, n - 1, True)]
        scs = [bench.uniform_fr(rng, m) for m, _, _ in cases]
        got = [reg.msm_bigint(sc, offset=o, reversed_=r) for sc, (m, o, r) in zip(scs, cases)]
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 62)))  # no tables of any kind
        for sc, (m, o, r), g in zip(scs, cases, got):
            assert (reg.msm_bigint(sc, offset=o, reversed_=r) == g).all(), (m, o, r)
    finally:
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 17)))
        reg.free()


def test_tables_are_the_default_for_a_resident_key(gm, oracle):
    """gm_set_auto_tables (on at gm_init): registering 2^17 .. 2^26 - 1 bases builds the fixed-base tables when they fit the
    budget; smaller keys, a tiny budget or the knob turned off leave the plain path.  Same group element either way."""
    lib = gm.capi.load()
    n = (1 << 17) + 3
    ks = oracle.random_fr(2901, n)
    sc = oracle.random_fr(2902, n)
    try:
        reg = gm.G1Bases.fixed_base(oracle.g1_generator(), ks)
        assert reg.table_info() == (20, (13 * n + 16 * (1 << 17)) * 96)  # + the c = 16 table of the small calls over the first 2^17 points
        with_tables = reg.msm_bigint(sc)
        assert_same_point(oracle, with_tables, oracle.msm_pippenger(reg.download(), sc))
        reg.free()
        small = gm.G1Bases.fixed_base(oracle.g1_generator(), ks[:5000])
        assert small.table_info() == (0, 0)
        small.free()
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(1 << 20)))  # a budget the tables do not fit
        reg = gm.G1Bases.fixed_base(oracle.g1_generator(), ks)
        assert reg.table_info() == (0, 0)
        reg.free()
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(0), C.c_size_t(0)))
        reg = gm.G1Bases.fixed_base(oracle.g1_generator(), ks)
        assert reg.table_info() == (0, 0)
        assert (reg.msm_bigint(sc) == with_tables).all()
        host = reg.download()
        reg.free()
        # points uploaded from the host: a registration may serve ONE MSM (VariableBaseMSM::msm), so it builds nothing;
        # a key that stays resident asks with precompute(-1), which follows the same budget rule
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(0)))
        up = gm.G1Bases.register(host)
        assert up.table_info() == (0, 0)
        assert (up.msm_bigint(sc) == with_tables).all()
        up.precompute(-1)
        assert up.table_info() == (20, (13 * n + 16 * (1 << 17)) * 96)
        up.precompute(-1)  # idempotent
        assert (up.msm_bigint(sc) == with_tables).all()
        up.free()
        from gemini_amd.kzg import CommitterKey

        ck = CommitterKey.from_powers(host, 3)
        assert ck.powers_of_g.table_info() == (20, (13 * n + 16 * (1 << 17)) * 96)
        ck.powers_of_g.free()
        tiny = gm.G1Bases.register(host[:4096])
        tiny.precompute(-1)
        assert tiny.table_info() == (0, 0)
        tiny.free()
    finally:
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(0)))


def test_msm_2_20_properties(gm, oracle, pyref):
    """BASELINE config 2 size.  Size-independent properties: additivity over a split of the pairs,
    linearity in the scalars, and the all-equal-scalar collapse sum_i e*P_i = e * sum_i P_i."""
    n = 1 << 20
    g = oracle.g1_generator()
    ks = oracle.random_fr(81, n)
    reg = gm.G1Bases.fixed_base(g, ks)
    try:
        a = oracle.random_fr(82, n)
        full = reg.msm_bigint(a)
        lo = reg.msm_bigint(a[: n // 2])
        hi = reg.msm_bigint(a[n // 2:], offset=n // 2)
        from gemini_amd.msm import g1_sum

        assert (g1_sum(np.stack([lo, hi])) == full).all()
        # discrete-log check: bases are k_i * G, so the MSM is (sum a_i k_i) * G
        ai = oracle.limbs_to_ints(a[: 1 << 12])
        ki = oracle.limbs_to_ints(ks[: 1 << 12])
        small = reg.msm_bigint(a[: 1 << 12])
        assert jac_to_affine_ints(oracle, small) == pyref.g1_mul(pyref.G1_GEN, sum(x * y for x, y in zip(ai, ki)) % pyref.R_MOD)
        # ... and on ALL 2^20 pairs: the whole result against (sum a_i k_i mod r) * G, no oracle MSM involved
        assert dot_ints(a[: 1 << 12], ks[: 1 << 12]) == sum(x * y for x, y in zip(ai, ki))
        assert jac_to_affine_ints(oracle, full) == pyref.g1_mul(pyref.G1_GEN, dot_ints(a, ks) % pyref.R_MOD)
        # all-equal scalars (the dummy_r1cs witness): e * sum P_i
        e = oracle.random_fr(83, 1)[0]
        ones = np.tile(oracle.ints_to_limbs([1], 4)[0], (n, 1))
        s1 = reg.msm_bigint(ones)
        se = reg.msm_bigint(np.tile(e, (n, 1)))
        exp = oracle.g1_mul(oracle.g1_to_affine(s1), e)
        assert_same_point(oracle, se, exp)
    finally:
        reg.free()


def test_fixed_base_tables_same_results(gm, oracle):
    """gm_g1_bases_precompute: MSMs through the shared-bucket / table path equal the plain path and the
    oracle, incl. offset / reversed addressing, all-equal scalars and identity bases."""
    n = 6000
    bases = rand_bases(oracle, 91, n)
    bases[17] = 0
    sc = oracle.random_fr(92, n)
    reg = gm.G1Bases.register(bases)
    gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1)))
    try:
        plain = reg.msm_bigint(sc)
        for c in (8, 13, 20):
            reg.precompute(c)
            got = reg.msm_bigint(sc)
            assert (got == plain).all(), c
            assert_same_point(oracle, got, oracle.msm_pippenger(bases, sc))
            assert_same_point(oracle, reg.msm_bigint(sc[:1000], offset=321), oracle.msm_pippenger(bases[321:1321], sc[:1000]))
            rev = bases[::-1].copy()
            assert_same_point(oracle, reg.msm_bigint(sc[:1000], offset=n - 1 - 500, reversed_=True), oracle.msm_pippenger(rev[500:1500], sc[:1000]))
            e = np.tile(oracle.random_fr(93, 1)[0], (n, 1))
            assert_same_point(oracle, reg.msm_bigint(e), oracle.msm_pippenger(bases, e))
    finally:
        gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1 << 17)))
        reg.free()


def test_msm_2_24_properties(gm, oracle, pyref):
    """`snark -i 24` size (BASELINE config 3: the witness commitment is a 2^24 - 1 pair MSM with all-equal
    scalars).  Size-independent properties only: additivity over a split, the all-equal collapse
    sum_i e*P_i = e * sum_i P_i, and the discrete-log identity on a prefix."""
    from gemini_amd.msm import g1_sum

    n = 1 << 24
    g = oracle.g1_generator()
    rng = np.random.default_rng(2424)
    import bench

    ks = bench.uniform_fr(rng, n)
    reg = gm.G1Bases.fixed_base(g, ks)
    try:
        a = bench.uniform_fr(rng, n)
        full = reg.msm_bigint(a)
        third = n // 3
        parts = [reg.msm_bigint(a[:third]), reg.msm_bigint(a[third:2 * third], offset=third), reg.msm_bigint(a[2 * third:], offset=2 * third)]
        assert (g1_sum(np.stack(parts)) == full).all()
        m = 1 << 11
        ai, ki = oracle.limbs_to_ints(a[:m]), oracle.limbs_to_ints(ks[:m])
        assert jac_to_affine_ints(oracle, reg.msm_bigint(a[:m])) == pyref.g1_mul(pyref.G1_GEN, sum(x * y for x, y in zip(ai, ki)) % pyref.R_MOD)
        # the WHOLE 2^24-pair result (tables path: the key was built by a constructor) and the same pairs on the plain path
        want = pyref.g1_mul(pyref.G1_GEN, dot_ints(a, ks) % pyref.R_MOD)
        assert jac_to_affine_ints(oracle, full) == want
        gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1 << 62)))
        try:
            assert (reg.msm_bigint(a) == full).all()
        finally:
            gm.capi.check(gm.capi.load().gm_set_msm_table_min(C.c_size_t(1 << 17)))
        e = oracle.random_fr(2425, 1)[0]
        ones = np.tile(oracle.ints_to_limbs([1], 4)[0], (n - 1, 1))
        s1 = reg.msm_bigint(ones)
        se = reg.msm_bigint(np.tile(e, (n - 1, 1)))  # the dummy_r1cs witness commitment shape
        assert_same_point(oracle, se, oracle.g1_mul(oracle.g1_to_affine(s1), e))
    finally:
        reg.free()


def test_batch_msm_equals_single_calls(gm, oracle):
    """gm_g1_msm_v_batch (pipelined enqueue/finish) returns exactly what k gm_g1_msm_v calls return,
    for mixed sizes including empty, and matches the oracle on one of them."""
    from gemini_amd.fr import FrVec

    n = 5000
    bases_h = oracle.g1_fixed_base_mul(oracle.g1_generator(), oracle.random_fr(31, n))
    b = gm.G1Bases.register(bases_h)
    sizes = [5000, 0, 1, 777, 4096, 5000, 33, 2]
    vecs = [FrVec.from_host(oracle.fr_to_mont(oracle.random_fr(40 + j, max(m, 1)))) for j, m in enumerate(sizes)]
    single = [b.msm_vec(v, n=m) for v, m in zip(vecs, sizes)]
    for _ in range(3):
        got = b.msm_vec_batch(vecs, sizes)
        assert all((got[j] == single[j]).all() for j in range(len(sizes)))
    exp = oracle.msm_pippenger(bases_h[:777], oracle.random_fr(43, 777))
    assert oracle.g1_jac_eq(got[3], exp)
    assert b.msm_vec_batch([], []).shape == (0, 18)
    for v in vecs:
        v.free()
    b.free()


def test_fused_small_calls_of_a_batch(gm, oracle):
    """the calls of <= 2^13 pairs of a batch (the tail of every folding tree, src/subprotocols/tensorcheck/mod.rs:124-133,
    src/kzg/time.rs:98-107) run as the levels of ONE fused pass (msm.hip: MsmMulti, k_digits_multi): twenty of them (more than one
    pass takes: the rest run on their own), between two larger calls, with all-equal scalars, zero scalars, identity bases, P and -P
    in one bucket, offsets and reversed walks -- every result equal to the single call's, and to the oracle's on two of them."""
    from gemini_amd.fr import FrVec

    n = 1 << 15
    bases_h = rand_bases(oracle, 4401, n)
    bases_h[5] = 0
    bases_h[9] = bases_h[8]
    b = gm.G1Bases.register(bases_h)
    rng = np.random.default_rng(44)
    sizes = [1 << 15, 8192, 8191, 4097, 1, 2, 3, 64, 65, 1000, 2048, 31, 32, 33, 5000, 7, 600, 8192, 129, 255, 256, 12, 1 << 14]
    hosts = [oracle.fr_to_mont(oracle.random_fr(4500 + j, m)) for j, m in enumerate(sizes)]
    hosts[3][:] = hosts[3][0]  # all-equal scalars: one bucket per window
    hosts[9][::2] = 0  # zero scalars
    vecs = [FrVec.from_host(h) for h in hosts]
    try:
        single = [b.msm_vec(v, n=m) for v, m in zip(vecs, sizes)]
        for _ in range(2):
            got = b.msm_vec_batch(vecs, sizes)
            assert all((got[j] == single[j]).all() for j in range(len(sizes))), [j for j in range(len(sizes)) if not (got[j] == single[j]).all()]
        for j in (2, 14):
            assert_same_point(oracle, got[j], oracle.msm_pippenger(bases_h[: sizes[j]], oracle.fr_from_mont(hosts[j])))
        offs = [int(rng.integers(0, n - m + 1)) for m in sizes]
        single_at = [b.msm_vec(v, n=m, offset=o) for v, m, o in zip(vecs, sizes, offs)]
        got = b.msm_vec_batch_at(vecs, sizes, offs)
        assert all((got[j] == single_at[j]).all() for j in range(len(sizes)))
        part = b.msm_vec_batch_at(vecs, sizes, offs, partial=True)
        assert all(oracle.g1_jac_eq(part[j], single_at[j]) for j in range(len(sizes)))
        roffs = [o + m - 1 for o, m in zip(offs, sizes)]
        single_rev = [b.msm_vec(v, n=m, offset=o, reversed_=True) for v, m, o in zip(vecs, sizes, roffs)]
        got = b.msm_vec_batch_at(vecs, sizes, roffs, reversed_=True)
        assert all((got[j] == single_rev[j]).all() for j in range(len(sizes)))
        with pytest.raises(gm.capi.GeminiHipError):
            b.msm_vec_batch_at(vecs[1:3], sizes[1:3], [n - 100, 0])  # a fused level that runs past the end of the key
    finally:
        for v in vecs:
            v.free()
        b.free()


def test_batch_randomised_differential(gm, oracle):
    """gm_g1_msm_v_batch_at against single calls on random batches: 2 .. 40 calls of log-uniform sizes 1 .. 2^15 (so that some batches
    have several fused groups of tiny calls, medium calls on the small lanes and big ones), random offsets, forward and reversed,
    normalised and partial; scalars uniform / all-equal / sparse per call."""
    from gemini_amd.fr import FrVec

    n = 1 << 15
    b = gm.G1Bases.register(rand_bases(oracle, 7701, n))
    rng = np.random.default_rng(7702)
    pool = oracle.fr_to_mont(oracle.random_fr(7703, n))
    try:
        for it in range(6):
            k = int(rng.integers(2, 41))
            sizes = [int(min(n, max(1, round(2 ** rng.uniform(0, 15))))) for _ in range(k)]
            if it == 0:
                sizes[:4] = [1, 8192, 8193, n]
            hosts = []
            for m in sizes:
                h = pool[int(rng.integers(0, n - m + 1)):][:m].copy()
                kind = int(rng.integers(0, 4))
                if kind == 1:
                    h[:] = h[0]
                elif kind == 2:
                    h[rng.random(m) < 0.7] = 0
                hosts.append(h)
            vecs = [FrVec.from_host(h) for h in hosts]
            rev = bool(it & 1)
            offs = [int(rng.integers(0, n - m + 1)) + (m - 1 if rev else 0) for m in sizes]
            single = [b.msm_vec(v, n=m, offset=o, reversed_=rev) for v, m, o in zip(vecs, sizes, offs)]
            got = b.msm_vec_batch_at(vecs, sizes, offs, reversed_=rev)
            bad = [j for j in range(k) if not (got[j] == single[j]).all()]
            assert not bad, (it, k, [sizes[j] for j in bad])
            part = b.msm_vec_batch_at(vecs, sizes, offs, reversed_=rev, partial=True)
            assert all(oracle.g1_jac_eq(part[j], single[j]) for j in range(k)), it
            for v in vecs:
                v.free()
    finally:
        b.free()


def test_segmented_srs_and_batch_with_offsets(gm, oracle, pyref):
    """gm_g1_srs_register_segments: ranges of tau^i g back to back in one handle (a rank's slices of
    CommitterKey::powers_of_g, src/kzg/time.rs:24-27) == the same ranges of the plain key;
    gm_g1_msm_v_batch_at: call j against bases[offsets[j] ...] == single calls with that offset, normalised or not"""
    from gemini_amd.fr import FrVec

    g = oracle.g1_generator()
    tau = oracle.random_fr(71, 1)[0]
    plain = gm.G1Bases.srs(g, tau, 1200)
    starts, counts = [512, 0, 1000, 7, 256], [300, 0, 200, 1, 128]
    seg = gm.G1Bases.srs_segments(g, tau, starts, counts)
    try:
        assert len(seg) == sum(counts)
        whole = plain.download()
        got = seg.download()
        at = 0
        for s0, c0 in zip(starts, counts):
            assert (got[at:at + c0] == whole[s0:s0 + c0]).all()
            at += c0
        offs = [0, 300, 500, 501, 300]
        ns = [300, 200, 1, 128, 57]
        vecs = [FrVec.from_host(oracle.fr_to_mont(oracle.random_fr(80 + j, m))) for j, m in enumerate(ns)]
        single = [seg.msm_vec(v, n=m, offset=o) for v, m, o in zip(vecs, ns, offs)]
        norm = seg.msm_vec_batch_at(vecs, ns, offs)
        assert all((norm[j] == single[j]).all() for j in range(len(ns)))
        part = seg.msm_vec_batch_at(vecs, ns, offs, partial=True)
        assert all(oracle.g1_jac_eq(part[j], single[j]) for j in range(len(ns)))
        # and against the plain key's own range: segment 2 = powers 1000 .. 1199
        assert (plain.msm_vec(vecs[1], n=200, offset=1000) == single[1]).all()
        # reversed: call j starts AT offsets[j] and walks down (the Reverse(powers_of_g) view of src/kzg/space.rs:95-125)
        roffs = [299, 499, 500, 628, 300]
        rsingle = [seg.msm_vec(v, n=m, offset=o, reversed_=True) for v, m, o in zip(vecs, ns, roffs)]
        rnorm = seg.msm_vec_batch_at(vecs, ns, roffs, reversed_=True)
        assert all((rnorm[j] == rsingle[j]).all() for j in range(len(ns)))
        rpart = seg.msm_vec_batch_at(vecs, ns, roffs, reversed_=True, partial=True)
        assert all(oracle.g1_jac_eq(rpart[j], rsingle[j]) for j in range(len(ns)))
        rev = oracle.fr_to_mont(oracle.random_fr(80, 300))[::-1].copy()
        rv = FrVec.from_host(rev)  # reversed walk over reversed scalars == the forward call
        assert (seg.msm_vec_batch_at([rv], [300], [299], reversed_=True)[0] == single[0]).all()
        rv.free()
        with pytest.raises(gm.capi.GeminiHipError):
            seg.msm_vec_batch_at(vecs[:1], [300], [298], reversed_=True)  # walks below base 0
        with pytest.raises(gm.capi.GeminiHipError):
            seg.msm_vec_batch_at(vecs[:1], [300], [400])  # runs past the end of the key
        for v in vecs:
            v.free()
    finally:
        seg.free()
        plain.free()


def _set_levels(gm, k):
    import ctypes as C

    rc = gm.capi.load().gm_set_msm_affine_levels(C.c_int(k))
    if rc != 0 and k != 0 and b"GM_EXPERIMENTS" in gm.capi.load().gm_last_error():
        pytest.skip("affine-level experiment not in this build (make EXTRA=-DGM_EXPERIMENTS)")
    gm.capi.check(rc)


@pytest.mark.parametrize("levels", [1, 2, 3, 5])
def test_affine_levels_same_results(gm, oracle, levels):
    """gm_set_msm_affine_levels: pairwise affine additions with one shared inversion per level in front of
    the XYZZ accumulation -- the result cannot depend on it.  Inputs hit every exceptional pair: repeated
    bases (P + P), P and -P in one bucket, identity bases, all-equal scalars (one bucket per window), zero
    scalars, and sizes around the lane count."""
    rng = np.random.default_rng(levels)
    cases = []
    n = 3000
    bases = rand_bases(oracle, 51, n)
    sc = oracle.random_fr(52, n)
    cases.append((bases, sc))
    b2 = bases.copy()
    b2[100:200] = bases[7]  # the same point many times
    b2[300] = 0  # identity
    b2[301] = 0
    s2 = sc.copy()
    s2[100:200] = sc[100]  # same scalar on the same point: P + P at every level
    s2[400:500] = 0
    neg = oracle.limbs_to_ints(sc[600:601])[0]
    s2[601] = oracle.ints_to_limbs([(0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001 - neg)], 4)[0]
    b2[601] = b2[600]  # k*P + (r-k)*P: digits cancel bucket-wise in places
    cases.append((b2, s2))
    s3 = np.tile(sc[5], (n, 1))  # the reference's benchmark instance: every pair in the same buckets
    cases.append((bases, s3))
    b4 = np.tile(bases[9], (n, 1))
    cases.append((b4, s3))  # one point, one scalar: only doublings
    cases.append((bases[:1], sc[:1]))
    cases.append((bases[:2], s3[:2]))
    try:
        for c in (0, 5):
            _set_window(gm, c)
            for bs, ss in cases:
                _set_levels(gm, 0)
                want = gm.VariableBaseMSM.msm_bigint(bs, ss)
                _set_levels(gm, levels)
                got = gm.VariableBaseMSM.msm_bigint(bs, ss)
                assert (got == want).all()
        _set_window(gm, 0)
        exp = oracle.msm_pippenger(cases[1][0], cases[1][1])
        assert_same_point(oracle, gm.VariableBaseMSM.msm_bigint(cases[1][0], cases[1][1]), exp)
    finally:
        _set_levels(gm, 0)
        _set_window(gm, 0)


def test_affine_levels_large(gm, oracle):
    """2^18 pairs, automatic and fixed level counts against the plain path, also with fixed-base tables"""
    n = 1 << 18
    ks = oracle.random_fr(61, n)
    b = gm.G1Bases.fixed_base(oracle.g1_generator(), ks)
    sc = oracle.random_fr(62, n)
    try:
        _set_levels(gm, 0)
        want = b.msm_bigint(sc)
        for lv in (1, 3, 4):
            _set_levels(gm, lv)
            assert (b.msm_bigint(sc) == want).all(), lv
        b.precompute(0)
        for lv in (0, 2):
            _set_levels(gm, lv)
            assert (b.msm_bigint(sc) == want).all(), ("tables", lv)
    finally:
        _set_levels(gm, 0)
        b.free()


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "few_values", "runs"])
def test_msm_skewed_digit_distributions_multi_block(gm, oracle, kind):
    """2^17 + 3 pairs (c = 16, a hundred sort blocks per pass) with scalars drawn from very few values, so whole
    waves hit the same sort bin (wave-aggregated LDS atomics), a bin's entries straddle many blocks of the
    staged scatter passes, and single buckets hold tens of thousands of entries -- against the CPU Pippenger."""
    n = (1 << 17) + 3
    rng = np.random.default_rng({"all_equal": 1, "two_values": 2, "few_values": 3, "runs": 4}[kind])
    ks = oracle.random_fr(91, n)
    reg = gm.G1Bases.fixed_base(oracle.g1_generator(), ks)
    vals = oracle.random_fr(92, 37)
    if kind == "all_equal":
        idx = np.zeros(n, dtype=np.int64)
    elif kind == "two_values":
        idx = rng.integers(0, 2, size=n)
    elif kind == "few_values":
        idx = rng.integers(0, 37, size=n)
    else:  # long runs of one value, then another: waves are uniform, blocks are not
        idx = (np.arange(n) // 3000) % 5
    sc = vals[idx]
    sc[::977] = 0  # holes
    try:
        got = reg.msm_bigint(sc)
        assert_same_point(oracle, got, oracle.msm_pippenger(reg.download(), sc))
    finally:
        reg.free()


def test_msm_rejects_scalars_that_are_not_fr_images(gm, oracle):
    """Canonical scalars (mont = 0) must be < r < 2^255 like every `BigInt` image of an Fr element.  With
    c * W = 256 a value with bit 255 set gives a top-window digit beyond the bucket range: the reference
    panics on the out-of-range bucket index (src/kzg/msm/variable_base.rs:133-146); the device clamps the
    digit (no out-of-bounds write) and the call fails with GM_EINVAL.  The next call is unaffected."""
    n = 300
    bases = gm.G1Bases.register(rand_bases(oracle, 91, n))
    sc = oracle.random_fr(92, n)
    good = bases.msm_bigint(sc)
    bad = sc.copy()
    bad[n // 2, 3] |= np.uint64(1 << 63)
    with pytest.raises(gm.capi.GeminiHipError) as ei:
        bases.msm_bigint(bad)
    assert ei.value.code == -1  # GM_EINVAL
    assert (bases.msm_bigint(sc) == good).all()
    assert_same_point(oracle, good, oracle.msm_pippenger(bases.download(), sc))
    bases.free()


def test_msm_window_group_split(gm, oracle):
    """gm_set_msm_split: a one-call MSM as two window groups on three streams gives the same group element
    (2^17 + 3 pairs, the smallest size that splits, and the degenerate all-equal-scalars instance)."""
    lib = gm.capi.load()
    n = (1 << 17) + 3
    bases = gm.G1Bases.fixed_base(oracle.g1_generator(), oracle.random_fr(95, n))
    sc = oracle.random_fr(96, n)
    eq = np.repeat(oracle.random_fr(97, 1), n, axis=0)
    plain = [bases.msm_bigint(sc), bases.msm_bigint(eq)]
    if lib.gm_set_msm_split(C.c_int(1)) != 0 and b"GM_EXPERIMENTS" in lib.gm_last_error():
        pytest.skip("window-group experiment not in this build (make EXTRA=-DGM_EXPERIMENTS)")
    try:
        split = [bases.msm_bigint(sc), bases.msm_bigint(eq)]
    finally:
        gm.capi.check(lib.gm_set_msm_split(C.c_int(0)))
    assert (split[0] == plain[0]).all() and (split[1] == plain[1]).all()
    assert_same_point(oracle, plain[0], oracle.msm_pippenger(bases.download(), sc))
    bases.free()


@pytest.mark.parametrize("n", [1, 33, 1000, 5000, (1 << 14) + 1, (1 << 17) + 3])
def test_msm_glv_same_results(gm, oracle, pyref, n):
    """gm_set_msm_glv: bases registered with their images under the curve endomorphism, scalars split as
    s = v1 + v2 lambda with |v1|, |v2| < 2^127 (half the windows, two digit strings per scalar).  Same group element as
    the plain path and as the CPU Pippenger, incl. the scalars that sit on the decomposition's edges."""
    lib = gm.capi.load()
    R = pyref.R_MOD
    lam = 0xAC45A4010001A40200000000FFFFFFFF
    host = rand_bases(oracle, 93, n)
    sc = oracle.random_fr(94, n)
    special = [0, 1, R - 1, lam, lam - 1, lam + 1, R - lam, lam // 2, lam // 2 + 1, ((lam + 1) // 2) * lam % R, ((lam + 1) // 2 + 1) * lam % R, 1 << 254, R // 2]
    sp = oracle.ints_to_limbs(special[: min(n, len(special))], 4)
    sc[: len(sp)] = sp
    plain = gm.G1Bases.register(host)
    if lib.gm_set_msm_glv(C.c_int(1)) != 0 and b"GM_EXPERIMENTS" in lib.gm_last_error():
        pytest.skip("GLV experiment not in this build (make EXTRA=-DGM_EXPERIMENTS)")
    try:
        glv = gm.G1Bases.register(host)
    finally:
        gm.capi.check(lib.gm_set_msm_glv(C.c_int(0)))
    a, b = plain.msm_bigint(sc), glv.msm_bigint(sc)
    assert (a == b).all()
    assert_same_point(oracle, b, oracle.msm_pippenger(host, sc))
    eq = np.repeat(oracle.random_fr(95, 1), n, axis=0)  # the benchmark instance's shape
    assert (plain.msm_bigint(eq) == glv.msm_bigint(eq)).all()
    # reversed / offset addressing (the stream key's view) goes through the same two arrays
    if n > 40:
        assert (plain.msm_bigint(sc[:30], offset=n - 5, reversed_=True) == glv.msm_bigint(sc[:30], offset=n - 5, reversed_=True)).all()
    plain.free()
    glv.free()


def test_msm_randomised_differential(gm, oracle):
    """seeded fuzz over the addressing and input forms of the MSM entry points -- sizes straddling the window-rule and
    sort-path thresholds, prefix / offset / reversed views of registered bases (CommitterKey::commit's prefix slice and
    CommitterKeyStream's Reverse + advance_by, src/kzg/space.rs:38-40,291-296), canonical and Montgomery scalars, sparse
    scalars (zeros, ones, small values: what real witnesses look like), identity bases, batches -- each against the CPU
    Pippenger on the same pairs"""
    from gemini_amd.fr import FrVec

    rng = np.random.default_rng(20240928)
    NB = 9000
    host = rand_bases(oracle, 96, NB)
    host[17] = 0  # identity records inside the key
    host[4099] = 0
    bases = gm.G1Bases.register(host)
    sizes = [1, 2, 7, 64, 255, 256, 257, 1023, 1025, 2047, 2049, 4097, 8191, 8193]
    for case in range(40):
        n = int(sizes[case % len(sizes)] if case < 28 else rng.integers(1, NB // 2))
        sc = oracle.random_fr(1000 + case, n)
        kind = case % 4
        if kind == 1:  # sparse: mostly 0 / 1 / small
            mask = rng.integers(0, 4, size=n)
            sc[mask == 0] = 0
            sc[mask == 1] = 0
            sc[mask == 1, 0] = 1
            small = mask == 2
            sc[small, 1:] = 0
            sc[small, 0] &= np.uint64(0xFFFF)
        reversed_ = bool(case & 1)
        off = int(rng.integers(n - 1, NB)) if reversed_ else int(rng.integers(0, NB - n + 1))
        idx = (off - np.arange(n)) if reversed_ else (off + np.arange(n))
        want = oracle.msm_pippenger(host[idx], sc)
        got = bases.msm_bigint(sc, offset=off, reversed_=reversed_)
        assert_same_point(oracle, got, want)
        assert is_normalised(oracle, got)
        if kind in (2, 3):  # the same through a resident Montgomery vector, and as a member of a batch
            v = FrVec.from_host(oracle.fr_to_mont(sc))
            assert (bases.msm_vec(v, offset=off, reversed_=reversed_) == got).all()
            if not reversed_ and off == 0:
                pass
            v.free()
    # batches of mixed sizes over the key's prefix (CommitterKey::batch_commit, src/kzg/time.rs:98-107)
    vecs, wants = [], []
    for j, n in enumerate([5000, 1, 300, 4096, 77, 2048, 9000, 13]):
        sc = oracle.random_fr(2000 + j, n)
        vecs.append(FrVec.from_host(oracle.fr_to_mont(sc)))
        wants.append(oracle.msm_pippenger(host[:n], sc))
    got = bases.msm_vec_batch(vecs, [len(v) for v in vecs])
    for g, w in zip(got, wants):
        assert_same_point(oracle, g, w)
    for v in vecs:
        v.free()
    bases.free()


def test_msm_beyond_one_call_is_the_sum_of_its_slices(gm, oracle):
    """2^26 + 2^25 + 3 pairs (the sizes of `snark -i 27/28` commitments): msm_run cuts the call at 2^26 pairs and both
    pieces take the widest window (c = 20, 13 windows), a path no oracle-sized test reaches.  Additivity pins it to the
    paths that are checked against the oracle: the same pairs as twelve slices of 2^23 (c = 19, checked at 2^23 - 3)."""
    import torch

    from gemini_amd.kzg import g1_generator_mont
    from gemini_amd.msm import g1_sum

    n = (1 << 26) + (1 << 25) + 3
    reg = gm.G1Bases.srs(g1_generator_mont(), oracle.random_fr(2601, 1)[0], n)  # tau^i * g: cheap to generate, distinct points
    try:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(2602)
        sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=gen)  # < 2^254 < r
        torch.cuda.synchronize()
        full = reg.msm_device(sc.data_ptr(), n, mont=False)
        step = 1 << 23
        parts = [reg.msm_device(sc[off:].data_ptr(), min(step, n - off), mont=False, offset=off, partial=True) for off in range(0, n, step)]
        assert (g1_sum(np.stack(parts)) == full).all()
        # and reversed addressing over the whole range (the stream view of the elastic prover)
        rev = reg.msm_device(sc.data_ptr(), n, mont=False, offset=n - 1, reversed_=True)
        parts = [reg.msm_device(sc[off:].data_ptr(), min(step, n - off), mont=False, offset=n - 1 - off, reversed_=True, partial=True) for off in range(0, n, step)]
        assert (g1_sum(np.stack(parts)) == rev).all()
        del sc
    finally:
        reg.free()

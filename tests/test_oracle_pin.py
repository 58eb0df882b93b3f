"""Pins oracle/gemini_oracle.c (the C restatement) against the independent Python big-int
statement oracle/pyref.py, and both against the naive definitions.  CPU only."""
import numpy as np
import pytest


def _rand_points(pyref, orc, seed, n):
    rng = pyref.SplitMix64(seed)
    ks = [rng.fr() for _ in range(n)]
    pts = [pyref.g1_mul(pyref.G1_GEN, k) for k in ks]
    arr = np.stack([orc.ints_to_affine(P) for P in pts])
    return pts, arr


def test_field_constants_and_mont(oracle, pyref):
    rng = pyref.SplitMix64(7)
    vals = [rng.fr() for _ in range(16)] + [0, 1, pyref.R_MOD - 1]
    m = oracle.fr_to_mont(oracle.ints_to_limbs(vals, 4))
    assert oracle.limbs_to_ints(m) == [pyref.fr_to_mont(v) for v in vals]
    assert oracle.limbs_to_ints(oracle.fr_from_mont(m)) == vals
    qs = [(rng.fr() * rng.fr()) % pyref.Q_MOD for _ in range(16)] + [0, 1, pyref.Q_MOD - 1]
    mq = oracle.fq_to_mont(oracle.ints_to_limbs(qs, 6))
    assert oracle.limbs_to_ints(mq) == [pyref.fq_to_mont(v) for v in qs]
    # products
    prod = oracle.fq_mul(mq, mq[::-1].copy())
    exp = [pyref.fq_to_mont(a * b % pyref.Q_MOD) for a, b in zip(qs, qs[::-1])]
    assert oracle.limbs_to_ints(prod) == exp
    prod = oracle.fr_mul(m, m[::-1].copy())
    exp = [pyref.fr_to_mont(a * b % pyref.R_MOD) for a, b in zip(vals, vals[::-1])]
    assert oracle.limbs_to_ints(prod) == exp


def test_generator_and_group_law(oracle, pyref):
    g = oracle.g1_generator()
    assert oracle.affine_to_ints(g) == pyref.G1_GEN
    assert oracle.g1_is_on_curve(g)
    # r * G = identity
    rlimbs = oracle.ints_to_limbs([pyref.R_MOD], 4)[0]
    assert not oracle.g1_mul(g, rlimbs)[12:].any()
    for k in [1, 2, 3, 5, 0xDEADBEEF, pyref.R_MOD - 1]:
        j = oracle.g1_mul(g, oracle.ints_to_limbs([k], 4)[0])
        assert oracle.affine_to_ints(oracle.g1_to_affine(j)) == pyref.g1_mul(pyref.G1_GEN, k)


def test_random_fr_stream_matches_pyref(oracle, pyref):
    rng = pyref.SplitMix64(0x47454D494E49)
    exp = [rng.fr() for _ in range(50)]
    assert oracle.limbs_to_ints(oracle.random_fr(0x47454D494E49, 50)) == exp


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 100])
def test_pippenger_vs_naive_vs_pyref(oracle, pyref, n):
    """src/kzg/msm/variable_base.rs:179-215 (test_var_base_msm): Pippenger == naive sum."""
    pts, arr = _rand_points(pyref, oracle, 100 + n, n)
    rng = pyref.SplitMix64(200 + n)
    sc = [rng.fr() for _ in range(n)]
    if n > 2:
        sc[0] = 0
        sc[1] = pyref.R_MOD - 1
        sc[2] = 1
    scl = oracle.ints_to_limbs(sc, 4)
    fast = oracle.msm_pippenger(arr, scl)
    naive = oracle.msm_naive(arr, scl)
    assert oracle.g1_jac_eq(fast, naive)
    assert oracle.affine_to_ints(oracle.g1_to_affine(fast)) == pyref.msm_naive(pts, sc)
    if n <= 33:
        assert pyref.pippenger(pts, sc) == pyref.msm_naive(pts, sc)


def test_signed_digits_recompose(oracle, pyref):
    """src/kzg/msm/variable_base.rs:63-93 (test_radix)"""
    rng = pyref.SplitMix64(5)
    for w in [3, 7, 13, 15, 16, 18]:
        for s in [rng.fr() for _ in range(8)] + [0, 1, pyref.R_MOD - 1]:
            d = oracle.signed_digits(oracle.ints_to_limbs([s], 4)[0], w)
            assert list(d) == pyref.signed_digits(s, w, 255)
            assert sum(int(x) << (w * i) for i, x in enumerate(d)) == s
            assert all(-(1 << (w - 1)) <= int(x) for x in d[:-1])


def test_degenerate_msm_inputs(oracle, pyref):
    """all-equal scalars (dummy_r1cs, src/circuit.rs:349-365), all-equal bases
    (examples/snark.rs:59-63), identity bases (src/kzg/time.rs:87), P + P and P - P."""
    pts, arr = _rand_points(pyref, oracle, 9, 40)
    rng = pyref.SplitMix64(10)
    e = rng.fr()
    sc = oracle.ints_to_limbs([e] * 40, 4)
    assert oracle.affine_to_ints(oracle.g1_to_affine(oracle.msm_pippenger(arr, sc))) == pyref.msm_naive(pts, [e] * 40)
    same = np.tile(arr[3], (40, 1))
    scr = [rng.fr() for _ in range(40)]
    got = oracle.affine_to_ints(oracle.g1_to_affine(oracle.msm_pippenger(same, oracle.ints_to_limbs(scr, 4))))
    assert got == pyref.g1_mul(pts[3], sum(scr) % pyref.R_MOD)
    arr2 = arr.copy()
    arr2[5] = 0
    arr2[6] = 0
    pts2 = list(pts)
    pts2[5] = pts2[6] = None
    got = oracle.affine_to_ints(oracle.g1_to_affine(oracle.msm_pippenger(arr2, oracle.ints_to_limbs(scr, 4))))
    assert got == pyref.msm_naive(pts2, scr)
    # P - P = identity
    two = np.stack([arr[0], arr[0]])
    got = oracle.msm_pippenger(two, oracle.ints_to_limbs([5, pyref.R_MOD - 5], 4))
    assert oracle.affine_to_ints(oracle.g1_to_affine(got)) is None


def test_chunked_and_hashmap_pippenger(oracle, pyref):
    """src/kzg/msm/stream_pippenger.rs:366-418 style: streamed == one-shot."""
    pts, arr = _rand_points(pyref, oracle, 11, 70)
    rng = pyref.SplitMix64(12)
    sc = [rng.fr() for _ in range(70)]
    scl = oracle.ints_to_limbs(sc, 4)
    full = oracle.msm_pippenger(arr, scl)
    for buf in [1, 7, 32, 70, 100]:
        assert oracle.g1_jac_eq(oracle.chunked_pippenger(arr, scl, buf), full)
    # hash map: duplicate bases merge their scalars
    dup = np.concatenate([arr, arr[:30]])
    sc2 = sc + [rng.fr() for _ in range(30)]
    exp = pyref.msm_naive(pts + pts[:30], sc2)
    for cap in [8, 64, 1000]:
        got = oracle.hashmap_pippenger(dup, oracle.fr_to_mont(oracle.ints_to_limbs(sc2, 4)), cap)
        assert oracle.affine_to_ints(oracle.g1_to_affine(got)) == exp
    # msm_chunks aligns by skipping the first len(bases)-len(scalars) bases (src/kzg/space.rs:36-40)
    got = oracle.msm_chunks(arr, scl[:50])
    assert oracle.affine_to_ints(oracle.g1_to_affine(got)) == pyref.msm_naive(pts[20:], sc[:50])


def test_fixed_base_mul(oracle, pyref):
    rng = pyref.SplitMix64(13)
    ks = [rng.fr() for _ in range(10)] + [0, 1, 255, 256]
    out = oracle.g1_fixed_base_mul(oracle.g1_generator(), oracle.ints_to_limbs(ks, 4))
    for k, a in zip(ks, out):
        assert oracle.affine_to_ints(a) == pyref.g1_mul(pyref.G1_GEN, k)


@pytest.mark.parametrize("nf,ng", [(2, 2), (3, 3), (30, 30), (93, 16), (16, 93), (17, 1), (64, 64)])
def test_sumcheck_time_prover_vs_pyref(oracle, pyref, nf, ng):
    """shapes from src/subprotocols/sumcheck/tests.rs:46,118-119,205 and time_prover.rs:141-159"""
    rng = pyref.SplitMix64(1000 + nf * 131 + ng)
    f = [rng.fr() for _ in range(nf)]
    g = [rng.fr() for _ in range(ng)]
    tw = rng.fr()
    P = pyref.TimeProver(f, g, tw)
    O = oracle.TimeProver(oracle.fr_to_mont(oracle.ints_to_limbs(f, 4)), oracle.fr_to_mont(oracle.ints_to_limbs(g, 4)),
                          oracle.fr_to_mont(oracle.ints_to_limbs([tw], 4))[0])
    assert P.tot_rounds == O.tot_rounds == pyref.ceil_log2(max(nf, ng))
    # claim: sum_i f_i g_i tw^i
    claim = sum(a * b * pow(tw, i, pyref.R_MOD) for i, (a, b) in enumerate(zip(f, g))) % pyref.R_MOD
    vm_p = vm_o = None
    while True:
        mp = P.next_message(vm_p)
        mo = O.next_message(vm_o)
        if mp is None:
            assert mo is None
            break
        got = tuple(oracle.limbs_to_ints(oracle.fr_from_mont(np.stack(mo))))
        assert got == mp
        # verifier identity (subclaim.rs:91-94): q(x) = a + b x + (claim - a) x^2... check q(0)+q(1)-b = claim
        a, b = mp
        r = rng.fr()
        c = (claim - a) % pyref.R_MOD
        claim = (a + b * r + c * r * r) % pyref.R_MOD
        vm_p = r
        vm_o = oracle.fr_to_mont(oracle.ints_to_limbs([r], 4))[0]
    ffp = P.final_foldings()
    ffo = oracle.limbs_to_ints(oracle.fr_from_mont(np.stack(O.final_foldings())))
    assert tuple(ffo) == ffp
    assert ffp[0] * ffp[1] % pyref.R_MOD == claim


def test_field_vector_helpers_vs_pyref(oracle, pyref):
    rng = pyref.SplitMix64(77)
    M = lambda v: oracle.fr_to_mont(oracle.ints_to_limbs(v, 4))
    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    f = [rng.fr() for _ in range(37)]
    g = [rng.fr() for _ in range(37)]
    x = rng.fr()
    assert I(oracle.fold_polynomial(M(f), M([x])[0])) == pyref.fold_polynomial(f, x)
    assert I(oracle.powers(M([x])[0], 20)) == pyref.powers(x, 20)
    assert I(oracle.powers2(M([x])[0], 9)) == pyref.powers2(x, 9)
    assert I(oracle.tensor(M(f[:6]))) == pyref.tensor(f[:6])
    assert I(oracle.evaluate_le(M(f), M([x])[0])) == [pyref.evaluate_le(f, x)]
    assert I(oracle.hadamard(M(f), M(g))) == pyref.hadamard(f, g)
    assert I(oracle.ip(M(f), M(g))) == [pyref.ip(f, g)]
    lc = oracle.linear_combination([M(f), M(g[:20]), M(f[:5])], M([x, 3, 5]))
    assert I(lc) == pyref.linear_combination([f, g[:20], f[:5]], [x, 3, 5])
    z = pyref.vanishing_polynomial([x, 5, 9])
    q, rem = oracle.poly_div_monic(M(f), M(z))
    qp, rp = pyref.poly_divmod(f, z)
    assert I(q) == qp and I(rem) == rp


def test_c_backed_new_time_equals_the_python_restatement(oracle, pyref):
    """oracle/snark_c.py (every O(n) pass a C call; the CPU baseline of bench.py's time_prover leg) against
    oracle/snark_ref.py (Python integers) on dummy_r1cs: byte-identical proofs, hence identical challenges"""
    from oracle import snark_c, snark_ref as sr, wire_ref as W

    for logn in (3, 4, 7):
        n = 1 << logn
        e, tau = 777 + logn, 31337 * (logn + 1)
        srs = sr.srs(tau, 2 * n + 1)
        a = sr.snark_new_time(sr.dummy_r1cs(e, n), srs)
        b = snark_c.new_time_dummy(e, n, srs)
        assert W.snark_proof(a, True) == W.snark_proof(b, True) and W.snark_proof(a, False) == W.snark_proof(b, False)

"""GPU parity: sumcheck TimeProver (fused fold + message kernel) and the misc.rs vector helpers
vs the CPU restatement.  Field arithmetic is exact, so every message must match bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _mont(orc, ints):
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4))


def _run_both(gm, orc, f, g, tw, seed):
    O = orc.TimeProver(f, g, tw)
    G = gm.TimeProver(f, g, tw)
    try:
        assert G.rounds() == O.tot_rounds
        ch = orc.fr_to_mont(orc.random_fr(seed, O.tot_rounds + 1))
        vm = None
        k = 0
        while True:
            mo = O.next_message(vm)
            mg = G.next_message(vm)
            if mo is None:
                assert mg is None
                break
            assert G.final_foldings() is None or G.round() == G.rounds()
            assert (mg[0] == mo[0]).all() and (mg[1] == mo[1]).all(), f"round {k}"
            vm = ch[k]
            k += 1
        assert k == O.tot_rounds
        fo, go = O.final_foldings()
        fg, gg = G.final_foldings()
        assert (fo == fg).all() and (go == gg).all()
        # a further call keeps returning None (time_prover.rs test_trivial_prover)
        assert G.next_message(None) is None
    finally:
        G.free()


@pytest.mark.parametrize("nf,ng", [(2, 2), (3, 3), (30, 30), (93, 16), (16, 93), (17, 1), (1025, 1025), (4096, 4096), (5000, 777)])
def test_time_prover_messages(gm, oracle, nf, ng):
    """shapes from src/subprotocols/sumcheck/tests.rs:46,118-119,205 + odd / unequal lengths"""
    f = oracle.fr_to_mont(oracle.random_fr(100 + nf, nf))
    g = oracle.fr_to_mont(oracle.random_fr(200 + ng, ng))
    tw = oracle.fr_to_mont(oracle.random_fr(300 + nf + ng, 1))[0]
    _run_both(gm, oracle, f, g, tw, 400 + nf)


def test_time_prover_dummy_r1cs_shape(gm, oracle):
    """the benchmark instance: z_a = z_b = [1; n] with twist alpha (src/snark/time_prover.rs:52)"""
    n = 1 << 14
    one = _mont(oracle, [1])[0]
    f = np.tile(one, (n, 1))
    tw = oracle.fr_to_mont(oracle.random_fr(5, 1))[0]
    _run_both(gm, oracle, f, f.copy(), tw, 6)


def test_time_prover_2_18(gm, oracle):
    n = 1 << 18
    f = oracle.fr_to_mont(oracle.random_fr(7, n))
    g = oracle.fr_to_mont(oracle.random_fr(8, n))
    tw = oracle.fr_to_mont(oracle.random_fr(9, 1))[0]
    _run_both(gm, oracle, f, g, tw, 10)


def test_time_prover_2_21_many_pairs_per_thread(gm, oracle):
    """2^20 pairs in the first round over 2^17 threads: every thread accumulates 8 (then 4, 2) products per inner product before
    its one Montgomery reduction (k_sc_round<.., LAZY>: 17-limb unreduced accumulators, src/misc.rs:235-266 `ip_unsafe` is the
    reference's CPU form of the same thing) -- bit for bit against the CPU restatement, every round"""
    n = (1 << 21) - 5
    f = oracle.fr_to_mont(oracle.random_fr(17, n))
    g = oracle.fr_to_mont(oracle.random_fr(18, n))
    f[:64] = oracle.fr_to_mont(oracle.ints_to_limbs([0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000000] * 64, 4))  # r - 1: the largest products
    g[:64] = f[:64]
    tw = oracle.fr_to_mont(oracle.random_fr(19, 1))[0]
    _run_both(gm, oracle, f, g, tw, 20)


def test_sumcheck_verifier_identity_2_20(gm, oracle, pyref):
    """size-independent property at 2^20: each round's quadratic q(x) = a + b x + (claim - a) x^2
    must satisfy q(rho) = next claim, ending at f0 * g0 (subclaim.rs:91-97)."""
    n = 1 << 20
    r = pyref.R_MOD
    f = oracle.fr_to_mont(oracle.random_fr(11, n))
    g = oracle.fr_to_mont(oracle.random_fr(12, n))
    twi = oracle.limbs_to_ints(oracle.random_fr(13, 1))[0]
    tw = _mont(oracle, [twi])[0]
    # claim = sum f_i g_i tw^i via the library's own hadamard / powers / ip (checked elsewhere vs oracle)
    pw = gm.powers(tw, n)
    fv = gm.FrVec.from_host(f)
    ft = gm.hadamard(fv, pw)
    gv = gm.FrVec.from_host(g)
    claim = gm.fr.fr_to_int(gm.ip(ft, gv))
    P = gm.TimeProver(fv, gv, tw)
    for v in (pw, fv, ft, gv):
        v.free()
    rho = oracle.limbs_to_ints(oracle.random_fr(14, 21))
    vm = None
    for k in range(20):
        a, b = P.next_message(vm)
        a, b = gm.fr.fr_to_int(a), gm.fr.fr_to_int(b)
        c = (claim - a) % r
        claim = (a + b * rho[k] + c * rho[k] * rho[k]) % r
        vm = gm.fr.fr_from_int(rho[k])
    assert P.next_message(vm) is None
    f0, g0 = P.final_foldings()
    assert gm.fr.fr_to_int(f0) * gm.fr.fr_to_int(g0) % r == claim
    P.free()


def test_vector_helpers_vs_oracle(gm, oracle):
    n = 3001
    f = oracle.fr_to_mont(oracle.random_fr(21, n))
    g = oracle.fr_to_mont(oracle.random_fr(22, n))
    x = oracle.fr_to_mont(oracle.random_fr(23, 3))
    assert (gm.fold_polynomial(f, x[0]).to_host() == oracle.fold_polynomial(f, x[0])).all()
    assert (gm.fold_polynomial(f[:1], x[0]).to_host() == oracle.fold_polynomial(f[:1], x[0])).all()
    assert (gm.powers(x[0], n).to_host() == oracle.powers(x[0], n)).all()
    assert (gm.hadamard(f, g).to_host() == oracle.hadamard(f, g)).all()
    assert (gm.ip(f, g) == oracle.ip(f, g)).all()
    ev = gm.evaluate_le(f, x)
    for k in range(3):
        assert (ev[k] == oracle.evaluate_le(f, x[k])).all()
    assert (gm.evaluate_le(f[:1], x[:1])[0] == f[0]).all()
    for k in [1, 2, 5, 12, 13]:
        rhos = oracle.fr_to_mont(oracle.random_fr(30 + k, k))
        assert (gm.tensor(rhos).to_host() == oracle.tensor(rhos)).all()
    lc = gm.linear_combination([f, g[:1000], f[:5]], x)
    assert (lc.to_host() == oracle.linear_combination([f, g[:1000], f[:5]], x)).all()
    with pytest.raises(gm.capi.GeminiHipError):
        gm.hadamard(f, g[:10])


def test_many_term_passes_vs_oracle(gm, oracle, pyref):
    """linear_combination of more terms than one launch takes (32 per pass: 70 ragged polynomials = three passes) and
    evaluate_le_batch of a folding-tree-shaped batch in its single launch (2^17 + 3 ... 1 coefficients, 1 to 512 blocks per job,
    an x / -x pair of points), both against the CPU restatement"""
    from gemini_amd.fr import FrVec, evaluate_le_batch

    rng = np.random.default_rng(321)
    lens = [int(v) for v in rng.integers(1, 5000, size=70)]
    lens[3], lens[40], lens[69] = 6000, 1, 5999
    polys = [oracle.fr_to_mont(oracle.random_fr(8800 + i, n)) for i, n in enumerate(lens)]
    ch = oracle.fr_to_mont(oracle.random_fr(8799, 70))
    assert (gm.linear_combination(polys, ch).to_host() == oracle.linear_combination(polys, ch)).all()
    # the trim of the result looks at the top 2^16 coefficients first and at the rest only if those are all zero
    padded = np.zeros((1000 + 70000, 4), dtype=np.uint64)
    padded[:1000] = polys[3][:1000]
    got = gm.linear_combination([padded, polys[40]], ch[:2]).to_host()
    assert len(got) == 1000 and (got == oracle.linear_combination([padded[:1000], polys[40]], ch[:2])).all()
    assert len(gm.linear_combination([np.zeros((70000, 4), dtype=np.uint64)], ch[:1])) == 0
    x = oracle.fr_to_mont(oracle.random_fr(8798, 1))[0]
    xi = oracle.limbs_to_ints(oracle.fr_from_mont(x.reshape(1, 4)))[0]
    pts = np.stack([x, oracle.fr_to_mont(oracle.ints_to_limbs([(pyref.R_MOD - xi) % pyref.R_MOD], 4))[0]])
    n, tree = (1 << 17) + 3, []
    while n >= 1:
        tree.append(oracle.fr_to_mont(oracle.random_fr(8700 + len(tree), n)))
        n //= 2
    vecs = [FrVec.from_host(t) for t in tree]
    got = evaluate_le_batch(vecs, pts)
    for j, t in enumerate(tree):
        for q in range(2):
            assert (got[j][q] == oracle.evaluate_le(t, pts[q])).all(), (j, q)
    for v in vecs:
        v.free()


def test_reference_known_answers_on_gpu(gm, oracle):
    """the reference's RNG-free tests, run through the device path:
    src/misc.rs:402-422 (linear_combination), src/subprotocols/tensorcheck/mod.rs:388-398 (fold)."""
    M = lambda v: _mont(oracle, v)
    got = gm.linear_combination([M([100, 101, 102, 103]), M([100, 100, 100, 100])], M([1, 10]))
    assert (got.to_host() == M([1100, 1101, 1102, 1103])).all()
    assert len(gm.linear_combination([], np.empty((0, 4), dtype=np.uint64))) == 0
    fold = gm.fold_polynomial(M([100, 101, 102, 103]), M([1])[0])
    assert (fold.to_host() == M([201, 205])).all()
    # trailing zeros are trimmed like DensePolynomial does
    got = gm.linear_combination([M([5, 7, 0, 0]), M([1, 0, 0, 0])], M([1, 1]))
    assert (got.to_host() == M([6, 7])).all()


def test_div_vanishing_vs_oracle(gm, oracle, pyref):
    """open_multi_points' quotient (src/kzg/time.rs:134-145) incl. the reference's known answer
    f(53) remainder check of src/kzg/space.rs:334-355"""
    from gemini_amd.fr import div_vanishing

    for n in [7, 64, 65, 1000, 70000]:
        f = oracle.fr_to_mont(oracle.random_fr(50 + n, n))
        pts_i = oracle.limbs_to_ints(oracle.random_fr(60 + n, 3))
        z = _mont(oracle, pyref.vanishing_polynomial(pts_i))
        q_exp, _ = oracle.poly_div_monic(f, z)
        q, _ = div_vanishing(f, _mont(oracle, pts_i))
        assert (q.to_host() == q_exp).all(), n
    f = _mont(oracle, [24, 7, 73, 3, 88, 80, 80])
    beta = 53
    pts = [beta * beta, beta, pyref.R_MOD - beta]
    q, _ = div_vanishing(f, _mont(oracle, pts))
    q_exp, rem = oracle.poly_div_monic(f, _mont(oracle, pyref.vanishing_polynomial(pts)))
    assert (q.to_host() == q_exp).all()
    assert oracle.limbs_to_ints(oracle.fr_from_mont(oracle.evaluate_le(rem, _mont(oracle, [beta])[0]))) == [1807299544171]


def test_prove_and_prove_batch_vs_restatement(gm, oracle, pyref):
    """Sumcheck::prove (proof.rs:36-66) and prove_batch (:69-122; shapes of sumcheck/tests.rs:141-200):
    messages, challenges and final foldings equal the Python restatement over the same transcript."""
    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    shapes = [(64, 64), (17, 5), (128, 128), (2, 2)]
    fs = [oracle.fr_to_mont(oracle.random_fr(700 + i, nf)) for i, (nf, _) in enumerate(shapes)]
    gs = [oracle.fr_to_mont(oracle.random_fr(800 + i, ng)) for i, (_, ng) in enumerate(shapes)]
    tws = oracle.fr_to_mont(oracle.random_fr(900, len(shapes)))
    # single prove, both the Python round loop and the in-library loop
    for native in (False, True):
        t = gm.Transcript()
        sc = gm.Sumcheck.new_time(t, fs[0], gs[0], tws[0], native=native)
        tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
        m, c, ff = pyref.sumcheck_prove(tr, pyref.TimeProver(I(fs[0]), I(gs[0]), I(tws[0])[0]))
        assert [(I(a)[0], I(b)[0]) for a, b in sc.messages] == m
        assert [I(x)[0] for x in sc.challenges] == c
        assert (I(sc.final_foldings[0][0])[0], I(sc.final_foldings[0][1])[0]) == ff
        assert I(t.get_challenge(b"next"))[0] == tr.get_challenge(b"next")
        t.free()
    # batch
    t = gm.Transcript()
    provers = [gm.TimeProver(f, g, tw) for f, g, tw in zip(fs, gs, tws)]
    sc = gm.Sumcheck.prove_batch(t, provers)
    tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    m, c, finals = pyref.sumcheck_prove_batch(tr, [pyref.TimeProver(I(f), I(g), I(tw)[0]) for f, g, tw in zip(fs, gs, tws)])
    assert sc.rounds == 8 == len(m)  # max rounds (7) + 1
    assert [(I(a)[0], I(b)[0]) for a, b in sc.messages] == m
    assert [I(x)[0] for x in sc.challenges] == c
    assert [(I(a)[0], I(b)[0]) for a, b in sc.final_foldings] == finals
    assert I(t.get_challenge(b"next"))[0] == tr.get_challenge(b"next")
    for p in provers:
        p.free()
    t.free()


def _msgs_equal(m1, m2):
    return all((a1 == a2).all() and (b1 == b2).all() for (a1, b1), (a2, b2) in zip(m1, m2)) and len(m1) == len(m2)


def test_messages_consistency_space_time(gm, oracle):
    """src/subprotocols/sumcheck/tests.rs:42-87: space prover (big-endian streams) == time prover, message by
    message and as whole proofs; also vs the CPU restatement."""
    n = 30  # DensePolynomial::rand(29)
    f = oracle.fr_to_mont(oracle.random_fr(41, n))
    g = oracle.fr_to_mont(oracle.random_fr(42, n))
    one = oracle.fr_to_mont(oracle.ints_to_limbs([1], 4))[0]
    tw = one
    tp = gm.TimeProver(f, g, tw)
    sp = gm.SpaceProver(f[::-1].copy(), g[::-1].copy(), tw)
    O = oracle.TimeProver(f, g, tw)
    for vm in (None, oracle.fr_to_mont(oracle.random_fr(43, 1))[0], one):
        ms, mt, mo = sp.next_message(vm), tp.next_message(vm), O.next_message(vm)
        assert (ms[0] == mt[0]).all() and (ms[1] == mt[1]).all()
        assert (ms[0] == mo[0]).all() and (ms[1] == mo[1]).all()
    tp.free()
    sp.free()
    for twist in (one, oracle.fr_to_mont(oracle.random_fr(44, 1))[0]):
        ts, tt = gm.Transcript(), gm.Transcript()
        space_proof = gm.Sumcheck.new_space(ts, f[::-1].copy(), g[::-1].copy(), twist)
        time_proof = gm.Sumcheck.new_time(tt, f, g, twist)
        assert _msgs_equal(space_proof.messages, time_proof.messages)
        assert all((x == y).all() for x, y in zip(space_proof.challenges, time_proof.challenges))
        assert (space_proof.final_foldings[0][0] == time_proof.final_foldings[0][0]).all()
        assert (space_proof.final_foldings[0][1] == time_proof.final_foldings[0][1]).all()
        ts.free()
        tt.free()


@pytest.mark.parametrize("nf,ng", [(1, 1), (2, 2), (5, 3), (1000, 1000), (1 << 13, (1 << 13) - 7), (700, 1 << 12)])
def test_borrowing_provers_leave_their_sources_alone(gm, oracle, nf, ng):
    """gm_sc_new_borrow / gm_sp_new_borrow: the same messages and foldings as the copying constructors, and the caller's
    vectors come back bit for bit after the last round (the second fold must not land in them)."""
    from gemini_amd.fr import FrVec

    f = oracle.fr_to_mont(oracle.random_fr(7100 + nf, nf))
    g = oracle.fr_to_mont(oracle.random_fr(7200 + ng, ng))
    tw = oracle.fr_to_mont(oracle.random_fr(7300, 1))[0]
    chal = oracle.fr_to_mont(oracle.random_fr(7400, 20))
    fv, gv = FrVec.from_host(f), FrVec.from_host(g)
    ref, bor = gm.TimeProver(f, g, tw), gm.TimeProver(fv, gv, tw, borrow=True)
    vm, k = None, 0
    while True:
        m1, m2 = ref.next_message(vm), bor.next_message(vm)
        assert (m1 is None) == (m2 is None)
        if m1 is None:
            break
        assert (m1[0] == m2[0]).all() and (m1[1] == m2[1]).all(), k
        vm = chal[k]
        k += 1
    f1, f2 = ref.final_foldings(), bor.final_foldings()
    assert (f1[0] == f2[0]).all() and (f1[1] == f2[1]).all()
    ref.free()
    bor.free()
    assert (fv.to_host() == f).all() and (gv.to_host() == g).all()
    # the space prover over the reversed streams, through its hand-off to a time prover after one round
    n = min(nf, ng)
    if n >= 4:
        fs, gs = FrVec.from_host(f[::-1].copy()), FrVec.from_host(g[::-1].copy())
        sp_ref, sp_bor = gm.SpaceProver(f[::-1].copy(), g[::-1].copy(), tw), gm.SpaceProver(fs, gs, tw, borrow=True)
        m1, m2 = sp_ref.next_message(None), sp_bor.next_message(None)
        assert (m1[0] == m2[0]).all() and (m1[1] == m2[1]).all()
        m1, m2 = sp_ref.next_message(chal[0]), sp_bor.next_message(chal[0])
        assert (m1[0] == m2[0]).all() and (m1[1] == m2[1]).all()
        t1, t2 = sp_ref.to_time_prover(), sp_bor.to_time_prover()
        sp_ref.free()
        sp_bor.free()
        m1, m2 = t1.next_message(chal[1]), t2.next_message(chal[1])
        assert (m1 is None) == (m2 is None)
        if m1 is not None:
            assert (m1[0] == m2[0]).all() and (m1[1] == m2[1]).all()
        t1.free()
        t2.free()
        assert (fs.to_host() == f[::-1]).all() and (gs.to_host() == g[::-1]).all()


def test_consistency_elastic(gm, oracle):
    """tests.rs:90-111: elastic == time.  With 30 coefficients the switch happens at the first fold;
    a 2^12-long instance with a lowered threshold exercises several space rounds before the hand-off."""
    import gemini_amd.sumcheck as S

    for n, thr in ((30, 22), (1 << 12, 8), (3000, 6)):
        f = oracle.fr_to_mont(oracle.random_fr(51 + n, n))
        g = oracle.fr_to_mont(oracle.random_fr(52 + n, n))
        tw = oracle.fr_to_mont(oracle.random_fr(53 + n, 1))[0]
        old = S.SPACE_TIME_THRESHOLD
        S.SPACE_TIME_THRESHOLD = thr
        try:
            te, tt = gm.Transcript(), gm.Transcript()
            elastic = gm.Sumcheck.new_elastic(te, f[::-1].copy(), g[::-1].copy(), tw)
            time_proof = gm.Sumcheck.new_time(tt, f, g, tw)
            assert _msgs_equal(elastic.messages, time_proof.messages), n
            assert (elastic.final_foldings[0][0] == time_proof.final_foldings[0][0]).all()
            assert (elastic.final_foldings[0][1] == time_proof.final_foldings[0][1]).all()
            te.free()
            tt.free()
        finally:
            S.SPACE_TIME_THRESHOLD = old


def test_space_prover_different_lengths(gm, oracle):
    """tests.rs:114-138: f of 93 and g of 16 coefficients, first message equal"""
    f = oracle.fr_to_mont(oracle.random_fr(61, 93))
    g = oracle.fr_to_mont(oracle.random_fr(62, 16))
    tw = oracle.fr_to_mont(oracle.ints_to_limbs([1], 4))[0]
    tp = gm.TimeProver(f, g, tw)
    sp = gm.SpaceProver(f[::-1].copy(), g[::-1].copy(), tw)
    ms, mt = sp.next_message(None), tp.next_message(None)
    assert (ms[0] == mt[0]).all() and (ms[1] == mt[1]).all()
    assert sp.rounds() == 4 and tp.rounds() == 7  # log2(min) vs log2(max), as in the reference
    tp.free()
    sp.free()


def test_herring_fmodule_prover(gm, oracle, pyref):
    """src/herring/time_prover.rs over FModule (module.rs:127-146): twist-free messages, twisted folds"""
    from gemini_amd.herring import FModuleTimeProver

    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    for nf, ng in ((64, 64), (33, 33), (100, 17)):
        f = oracle.fr_to_mont(oracle.random_fr(1100 + nf, nf))
        g = oracle.fr_to_mont(oracle.random_fr(1200 + ng, ng))
        tw = oracle.fr_to_mont(oracle.random_fr(1300, 1))[0]
        ch = oracle.fr_to_mont(oracle.random_fr(1400, 10))
        G = FModuleTimeProver(f, g, tw)
        P = pyref.HerringTimeProver("F", I(f), I(g), I(tw)[0])
        assert G.rounds() == P.tot_rounds
        vm_g = vm_p = None
        k = 0
        while True:
            mg, mp = G.next_message(vm_g), P.next_message(vm_p)
            if mp is None:
                assert mg is None
                break
            assert (I(mg[0])[0], I(mg[1])[0]) == mp, (nf, ng, k)
            vm_g, vm_p = ch[k], I(ch[k])[0]
            k += 1
        fg, fp = G.final_foldings(), P.final_foldings()
        assert (I(fg[0])[0], I(fg[1])[0]) == fp
        G.free()


def test_herring_g1module_prover(gm, oracle, pyref):
    """src/herring/time_prover.rs over G1Module (module.rs:81-102): messages are MSMs over the even/odd
    halves, fold is split_fold on G1 (P_even + (r*twist) P_odd)"""
    from gemini_amd.herring import G1ModuleTimeProver
    from tests.util import jac_to_affine_ints, rand_bases

    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    for n in (16, 11):
        pts_arr = rand_bases(oracle, 1500 + n, n)
        pts = [oracle.affine_to_ints(p) for p in pts_arr]
        g = oracle.fr_to_mont(oracle.random_fr(1600 + n, n))
        tw = oracle.fr_to_mont(oracle.random_fr(1700, 1))[0]
        ch = oracle.fr_to_mont(oracle.random_fr(1800, 6))
        G = G1ModuleTimeProver(pts_arr, g, tw)
        P = pyref.HerringTimeProver("G1", pts, I(g), I(tw)[0])
        assert G.rounds() == P.tot_rounds
        vm_g = vm_p = None
        k = 0
        while True:
            mg, mp = G.next_message(vm_g), P.next_message(vm_p)
            if mp is None:
                assert mg is None
                break
            assert (jac_to_affine_ints(oracle, mg[0]), jac_to_affine_ints(oracle, mg[1])) == mp, (n, k)
            vm_g, vm_p = ch[k], I(ch[k])[0]
            k += 1
        fg, fp = G.final_foldings(), P.final_foldings()
        assert (jac_to_affine_ints(oracle, fg[0]), I(fg[1])[0]) == fp
        G.free()


def test_sharded_prover_on_device(gm, oracle, pyref):
    """contiguous even-aligned shards, each with its twist origin tau^(2*first pair)
    (gm_sc_set_shard); the shard messages add up to the unsharded prover's message in every round while
    the shards stay pair-aligned, and the gathered tail continues identically."""
    N = 1 << 12
    f = oracle.fr_to_mont(oracle.random_fr(2001, N))
    g = oracle.fr_to_mont(oracle.random_fr(2002, N))
    tw = oracle.fr_to_mont(oracle.random_fr(2003, 1))[0]
    ch = oracle.fr_to_mont(oracle.random_fr(2004, 13))
    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    for world in (2, 4):
        per = N // world
        shards = [gm.TimeProver(f[r * per:(r + 1) * per], g[r * per:(r + 1) * per], tw) for r in range(world)]
        for r, s in enumerate(shards):
            s.set_shard(r * per // 2)
        ref = gm.TimeProver(f, g, tw)
        vm = None
        rounds_sharded = 0
        cur = per
        while cur >= 4:  # shards stay pair-aligned while their length is a multiple of 4 before a fold
            mr = ref.next_message(vm)
            ms = [s.next_message(vm) for s in shards]
            tot = [sum(I(m[i])[0] for m in ms) % pyref.R_MOD for i in range(2)]
            assert tot == [I(mr[0])[0], I(mr[1])[0]], (world, rounds_sharded)
            vm = ch[rounds_sharded]
            rounds_sharded += 1
            cur //= 2
        # hand-off: apply the pending fold shard-locally, gather, continue on one prover
        for s in shards:
            s.fold(vm)
        ref_m = ref.next_message(vm)
        states = [s.state() for s in shards]
        fs = np.concatenate([st[0] for st in states])
        gs = np.concatenate([st[1] for st in states])
        tail = gm.TimeProver(fs, gs, states[0][2])
        mt = tail.next_message(None)
        assert (mt[0] == ref_m[0]).all() and (mt[1] == ref_m[1]).all()
        k = rounds_sharded
        while True:
            vm = ch[k]
            k += 1
            a, b = ref.next_message(vm), tail.next_message(vm)
            if a is None:
                assert b is None
                break
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        fa, fb = ref.final_foldings(), tail.final_foldings()
        assert (fa[0] == fb[0]).all() and (fa[1] == fb[1]).all()
        for s in shards + [ref, tail]:
            s.free()


def test_device_sumchecks_are_accepted_by_the_reference_verifier(gm, oracle, pyref):
    """src/subprotocols/sumcheck/tests.rs:203-224 (test_sumcheck_correctness, d = 2^10) and :227-269
    (test_batch_sumcheck_correctness, d = 2^5 and 2^10): the verifier's Subclaim accepts the device prover's messages for
    the naively computed inner product <f . powers(twist), g> (oracle/verifier_ref.py restates subclaim.rs)."""
    from oracle import verifier_ref as V

    I = lambda a: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(a).reshape(-1, 4)))
    R = pyref.R_MOD

    def instance(seed, d):
        f, g = oracle.fr_to_mont(oracle.random_fr(seed, d + 1)), oracle.fr_to_mont(oracle.random_fr(seed + 1, d + 1))
        tw = oracle.fr_to_mont(oracle.random_fr(seed + 2, 1))[0]
        twi = I(tw)[0]
        asserted = sum(x * y % R * pow(twi, i, R) for i, (x, y) in enumerate(zip(I(f), I(g)))) % R
        return f, g, tw, asserted

    f, g, tw, asserted = instance(4100, 1 << 10)
    t = gm.Transcript()
    sc = gm.Sumcheck.new_time(t, f, g, tw)
    msgs = [(I(a)[0], I(b)[0]) for a, b in sc.messages]
    ff = (I(sc.final_foldings[0][0])[0], I(sc.final_foldings[0][1])[0])
    ch, _ = V.subclaim_new(pyref.GeminiTranscript(pyref.PROTOCOL_NAME), msgs, ff, asserted)
    assert ch == [I(x)[0] for x in sc.challenges]
    with pytest.raises(V.VerificationError):
        V.subclaim_new(pyref.GeminiTranscript(pyref.PROTOCOL_NAME), msgs, ff, (asserted + 1) % R)
    t.free()
    # batch of two provers of different lengths
    f1, g1, tw1, s1 = instance(4200, 1 << 5)
    f2, g2, tw2, s2 = instance(4300, 1 << 10)
    t = gm.Transcript()
    provers = [gm.TimeProver(f1, g1, tw1), gm.TimeProver(f2, g2, tw2)]
    sc = gm.Sumcheck.prove_batch(t, provers)
    msgs = [(I(a)[0], I(b)[0]) for a, b in sc.messages]
    finals = [(I(a)[0], I(b)[0]) for a, b in sc.final_foldings]
    V.subclaim_new_batch(pyref.GeminiTranscript(pyref.PROTOCOL_NAME), msgs, finals, [s1, s2])
    with pytest.raises(V.VerificationError):
        V.subclaim_new_batch(pyref.GeminiTranscript(pyref.PROTOCOL_NAME), msgs, finals, [s1, (s2 + 1) % R])
    for p in provers:
        p.free()
    t.free()


def test_fr_vector_ops_randomised_differential(gm, oracle, pyref):
    """seeded sweep over lengths (0, 1, odd, just around block and wave sizes, a few thousand) of every dense Fr pass of
    src/misc.rs and of the vanishing-polynomial division against the CPU restatement -- the ragged shapes the folding
    levels of a non-power-of-two instance produce"""
    from gemini_amd.fr import div_vanishing

    rng = np.random.default_rng(20240607)
    lengths = [0, 1, 2, 3, 5, 63, 64, 65, 127, 255, 256, 257, 1023, 1025, 4097] + [int(x) for x in rng.integers(2, 9000, size=12)]
    for it, n in enumerate(lengths):
        f = oracle.fr_to_mont(oracle.random_fr(7000 + it, max(n, 1)))[:n]
        m = int(rng.integers(0, n + 3)) if n else 0
        g = oracle.fr_to_mont(oracle.random_fr(7100 + it, max(m, 1)))[:m]
        x = oracle.fr_to_mont(oracle.random_fr(7200 + it, 4))
        if n:
            assert (gm.fold_polynomial(f, x[0]).to_host() == oracle.fold_polynomial(f, x[0])).all(), n
            ev = gm.evaluate_le(f, x[:3])
            assert all((ev[k] == oracle.evaluate_le(f, x[k])).all() for k in range(3)), n
            assert (gm.hadamard(f, f[::-1].copy()).to_host() == oracle.hadamard(f, f[::-1].copy())).all(), n
            assert (gm.ip(f, f[::-1].copy()) == oracle.ip(f, f[::-1].copy())).all(), n
        assert (gm.powers(x[1], n).to_host() == oracle.powers(x[1], n)).all(), n
        polys = [p for p in (f, g, f[: n // 3]) if len(p)]
        if polys:
            got = gm.linear_combination(polys, x[: len(polys)]).to_host()
            assert (got == oracle.linear_combination(polys, x[: len(polys)])).all(), (n, m)
        k = int(rng.integers(1, 4))
        if n > k:
            pts_i = oracle.limbs_to_ints(oracle.random_fr(7300 + it, k))
            q, _ = div_vanishing(f, _mont(oracle, pts_i))
            q_exp, _ = oracle.poly_div_monic(f, _mont(oracle, pyref.vanishing_polynomial(pts_i)))
            assert (q.to_host() == q_exp).all(), (n, k)

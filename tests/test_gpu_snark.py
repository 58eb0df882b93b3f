"""GPU parity for the callers of the hot path: KZG CommitterKey
(src/kzg/time.rs) and the snark time prover (src/snark/time_prover.rs) through the device path vs
the CPU restatement (oracle/snark_ref.py), transcript included."""
import numpy as np
import pytest

from tests.util import jac_to_affine_ints

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _M(orc, ints):
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4))


def test_committer_key_vs_oracle(gm, oracle, pyref):
    """src/kzg/time.rs tests (:162-211) + src/kzg/tests.rs:16-59 shape: commit / open /
    open_multi_points / batch_open_multi_points against the restatement."""
    from gemini_amd.kzg import CommitterKey
    from oracle import snark_ref as sr

    tau = oracle.limbs_to_ints(oracle.random_fr(1, 1))[0]
    ck = CommitterKey.new(200, 3, oracle.ints_to_limbs([tau], 4)[0])
    srs = sr.srs(tau, 201)
    assert len(ck.powers_of_g) == 201
    assert (ck.powers_of_g.download() == srs).all()
    poly = oracle.limbs_to_ints(oracle.random_fr(2, 101))
    pm = _M(oracle, poly)
    assert jac_to_affine_ints(oracle, ck.commit(pm)) == sr.commit(srs, poly)
    # trivial commitment test (:174-190): f = x + x^2, alpha = 0 -> evaluation 0
    ev, proof = ck.open(_M(oracle, [0, 1, 1]), _M(oracle, [0])[0])
    assert not ev.any()
    q, _ = pyref.poly_divmod([0, 1, 1], [0, 1])
    assert jac_to_affine_ints(oracle, proof) == sr.commit(srs, q)
    # open at a random point: evaluation == polynomial.evaluate (:193-211)
    a = oracle.limbs_to_ints(oracle.random_fr(3, 1))[0]
    ev, proof = ck.open(pm, _M(oracle, [a])[0])
    assert gm.fr.fr_to_int(ev) == pyref.evaluate_le(poly, a)
    q, _ = pyref.poly_divmod(poly, pyref.vanishing_polynomial([a]))
    assert jac_to_affine_ints(oracle, proof) == sr.commit(srs, q)
    pts = [a * a % pyref.R_MOD, a, (-a) % pyref.R_MOD]
    assert jac_to_affine_ints(oracle, ck.open_multi_points(pm, _M(oracle, pts))) == sr.open_multi_points(srs, poly, pts)
    polys = [oracle.limbs_to_ints(oracle.random_fr(10 + k, n)) for k, n in enumerate([101, 50, 25, 7])]
    chal = oracle.limbs_to_ints(oracle.random_fr(4, 1))[0]
    got = ck.batch_open_multi_points([_M(oracle, p) for p in polys], _M(oracle, pts), _M(oracle, [chal])[0])
    assert jac_to_affine_ints(oracle, got) == sr.batch_open_multi_points(srs, polys, pts, chal)
    batch = ck.batch_commit([_M(oracle, p) for p in polys])
    assert [jac_to_affine_ints(oracle, c) for c in batch] == [sr.commit(srs, p) for p in polys]


@pytest.mark.parametrize("logn", [3, 6, 9])
def test_snark_time_prover_dummy_r1cs(gm, oracle, pyref, logn):
    """examples/snark.rs:69-79 (time_snark_main) at small sizes: every proof element, i.e. every
    commitment and every sumcheck message, equals the restatement's -- which also fixes every
    Fiat-Shamir challenge in between."""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr

    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(100 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(200 + logn, 1))[0]
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])  # examples/snark.rs:75
    r1cs = dummy_r1cs(e, n)
    proof = Proof.new_time(r1cs, ck)
    exp = sr.snark_new_time(sr.dummy_r1cs(e, n), sr.srs(tau, 2 * n + 1))
    I = gm.fr.fr_to_int
    assert jac_to_affine_ints(oracle, proof.witness_commitment) == exp["witness_commitment"]
    assert I(proof.zc_alpha) == exp["zc_alpha"]
    for got, want in ((proof.first_sumcheck_msgs, exp["first_sumcheck_msgs"]), (proof.second_sumcheck_msgs, exp["second_sumcheck_msgs"])):
        assert [(I(a), I(b)) for a, b in got[0]] == want[0]
        assert (I(got[1][0][0]), I(got[1][0][1])) == want[1]
    tc, etc = proof.tensorcheck_proof, exp["tensorcheck_proof"]
    assert [jac_to_affine_ints(oracle, c) for c in tc.folded_polynomials_commitments] == etc["folded_polynomials_commitments"]
    assert [[I(x) for x in e2] for e2 in tc.folded_polynomials_evaluations] == etc["folded_polynomials_evaluations"]
    assert [[I(x) for x in e3] for e3 in tc.base_polynomials_evaluations] == etc["base_polynomials_evaluations"]
    assert jac_to_affine_ints(oracle, tc.evaluation_proof) == etc["evaluation_proof"]
    # proof shape of src/snark/mod.rs:76-82 at this size, and the compressed size examples/snark.rs:96 prints:
    # 48 + 32 + 2*(8 + 64 logn + 8 + 64) + (8 + 48 (logn-1)) + (8 + 64 (logn-1)) + 48 + (8 + 96)  (= 6056 at logn 24)
    assert len(proof.first_sumcheck_msgs[0]) == logn and len(tc.folded_polynomials_commitments) == logn - 1
    assert proof.compressed_size() == 48 + 32 + 2 * (8 + 64 * logn + 8 + 64) + (8 + 48 * (logn - 1)) + (8 + 64 * (logn - 1)) + 48 + (8 + 96)
    # wire formats: the bytes of the device proof equal the oracle-side serialisation of the restatement's proof in
    # both ark-serialize modes and both G1 framings, and deserialise back to an equal proof
    from oracle import wire_ref as W

    for compress in (True, False):
        for enc, mode in ((0, "arkworks"), (1, "zcash")):
            blob = proof.serialize(compress, enc)
            assert blob == W.snark_proof(exp, compress, mode)
            assert Proof.deserialize(blob, compress, enc, validate=(logn == 3)) == proof
    r1cs.free()


def test_snark_time_prover_general_r1cs(gm, oracle, pyref):
    """a non-diagonal instance: random sparse A, B and C chosen so that Az . Bz = Cz row by row"""
    from gemini_amd.circuit import R1cs, SparseMatrix
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr

    n = 32
    rng = pyref.SplitMix64(77)
    R = pyref.R_MOD
    z = [rng.fr() for _ in range(n)]
    mk = lambda: [[(rng.fr(), int(rng.next() % n)) for _ in range(1 + int(rng.next() % 3))] for _ in range(n)]
    a, b = mk(), mk()
    za, zb = sr.matvec(a, z), sr.matvec(b, z)
    c = [[(za[i] * zb[i] % R * pow(z[i], -1, R) % R, i)] for i in range(n)]
    inst = {"a": a, "b": b, "c": c, "z": z, "w": z[1:], "x": z[:1]}
    tau = rng.fr()
    exp = sr.snark_new_time(inst, sr.srs(tau, 2 * n + 1))
    M = lambda v: gm.fr.fr_from_int(v)
    dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]
    mats = [SparseMatrix.from_rows(dev(m), n) for m in (a, b, c)] + [SparseMatrix.from_rows(dev(m), n, transpose=True) for m in (a, b, c)]
    r1cs = R1cs(*mats, gm.FrVec.from_host(_M(oracle, z)), gm.FrVec.from_host(_M(oracle, z[1:])), gm.FrVec.from_host(_M(oracle, z[:1])))
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    proof = Proof.new_time(r1cs, ck)
    I = gm.fr.fr_to_int
    assert jac_to_affine_ints(oracle, proof.witness_commitment) == exp["witness_commitment"]
    assert [(I(x), I(y)) for x, y in proof.second_sumcheck_msgs[0]] == exp["second_sumcheck_msgs"][0]
    assert jac_to_affine_ints(oracle, proof.tensorcheck_proof.evaluation_proof) == exp["tensorcheck_proof"]["evaluation_proof"]
    # and the reference's verifier accepts it (src/snark/tests.rs:71)
    from oracle import verifier_ref as V
    from tests.util import snark_proof_to_ints

    V.snark_verify(snark_proof_to_ints(gm, oracle, proof), inst, V.VerifierKey.from_trapdoor(tau, 5))
    r1cs.free()


def test_committer_key_stream_consistency(gm, oracle, pyref):
    """src/kzg/tests.rs:16-59 (time == space commitments and openings), src/kzg/space.rs:313-387
    (open_multi_points incl. the 1807299544171 known answer), commit_folding / open_folding vs the
    time-side equivalents."""
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream, FoldedPolynomialTree
    from oracle import snark_ref as sr

    I = gm.fr.fr_to_int
    tau = oracle.limbs_to_ints(oracle.random_fr(71, 1))[0]
    time_ck = CommitterKey.new(200, 3, oracle.ints_to_limbs([tau], 4)[0])
    space_ck = CommitterKeyStream.from_committer_key(time_ck)
    srs = sr.srs(tau, 201)
    assert len(space_ck.powers_of_g) == len(time_ck.powers_of_g)  # test_srs
    # test_commitment_consistency (d = 15) and a longer one
    for d, seed in ((15, 72), (100, 73)):
        poly = oracle.fr_to_mont(oracle.random_fr(seed, d + 1))
        stream = poly[::-1].copy()  # Reverse(polynomial.coeffs())
        tc, scm = time_ck.commit(poly), space_ck.commit(stream)
        assert (tc == scm).all()
        # test_open_consistency
        alpha = oracle.fr_to_mont(oracle.random_fr(seed + 10, 1))[0]
        te, tp = time_ck.open(poly, alpha)
        se, sp = space_ck.open(stream, alpha, 1 << 20)
        assert (te == se).all() and (tp == sp).all()
        space_ck.min_device_chunk = 1  # cut literally (the default merges short flushes into one device MSM)
        se2, sp2 = space_ck.open(stream, alpha, 7)  # tiny buffer: many ChunkedPippenger flushes
        space_ck.min_device_chunk = space_ck.DEFAULT_MIN_DEVICE_CHUNK
        assert (te == se2).all() and (tp == sp2).all()
    # space.rs test_open_multi_points
    f_be = [80, 80, 88, 3, 73, 7, 24]
    stream = oracle.fr_to_mont(oracle.ints_to_limbs(f_be, 4))
    beta = 53
    M = lambda v: oracle.fr_to_mont(oracle.ints_to_limbs(v, 4))
    rem, _ = space_ck.open_multi_points(stream, M([beta * beta, beta, pyref.R_MOD - beta]), 1 << 20)
    assert pyref.evaluate_be([I(x) for x in rem], beta) == 1807299544171
    rem1, _ = space_ck.open_multi_points(stream, M([beta]), 1 << 20)
    assert len(rem1) == 1
    poly_i = oracle.limbs_to_ints(oracle.random_fr(74, 101))
    poly = M(poly_i)
    stream = poly[::-1].copy()
    b = oracle.limbs_to_ints(oracle.random_fr(75, 1))[0]
    _, proof_batch = space_ck.open_multi_points(stream, M([b]), 1 << 20)
    _, proof_single = space_ck.open(stream, M([b])[0], 1 << 20)
    assert (proof_batch == proof_single).all()
    pts = [b, pyref.R_MOD - b, b * b % pyref.R_MOD]
    rem, proof = space_ck.open_multi_points(stream, M(pts), 1 << 20)
    rem_i = [I(x) for x in rem]
    for p in pts:
        assert pyref.evaluate_be(rem_i, p) == pyref.evaluate_le(poly_i, p)
    assert (proof == time_ck.open_multi_points(poly, M(pts))).all()
    # commit_folding: level i = commitment of the i-fold folding
    chal_i = oracle.limbs_to_ints(oracle.random_fr(76, 6))
    tree = FoldedPolynomialTree(stream, M(chal_i))
    got = space_ck.commit_folding(tree, 1 << 20)
    cur = poly_i
    folds = []
    for c in chal_i:
        cur = pyref.fold_polynomial(cur, c)
        folds.append(cur)
    assert [jac_to_affine_ints(oracle, x) for x in got] == [sr.commit(srs, f) for f in folds]
    assert [jac_to_affine_ints(oracle, x) for x in space_ck.commit_folding(tree, 12)] == [sr.commit(srs, f) for f in folds]
    # open_folding: proof = commit(sum_i eta_i * quotient_i), remainders evaluate like the foldings
    etas = oracle.limbs_to_ints(oracle.random_fr(77, 6))
    rems, proof = space_ck.open_folding(tree, M(pts), M(etas), 1 << 20)
    z = pyref.vanishing_polynomial(pts)
    quots = [pyref.poly_divmod(f, z)[0] for f in folds]
    assert jac_to_affine_ints(oracle, proof) == sr.commit(srs, pyref.linear_combination(quots, etas))
    for f, r in zip(folds, rems):
        r_i = [I(x) for x in r]
        assert len(r_i) == 3
        for p in pts:
            assert pyref.evaluate_be(r_i, p) == pyref.evaluate_le(f, p)


@pytest.mark.parametrize("kind", ["dummy", "general"])
def test_snark_consistency_time_vs_elastic(gm, oracle, pyref, kind):
    """src/snark/tests.rs:14-57 (test_snark_consistency): assert_eq!(time_proof, space_proof) -- every
    commitment, sumcheck message, evaluation and the opening proof of Proof::new_elastic equal those of
    Proof::new_time (max_msm_buffer = 20 forces many Pippenger flushes, as in the reference test)."""
    import gemini_amd.sumcheck as S
    from gemini_amd.circuit import R1cs, R1csStream, SparseMatrix, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.snark import Proof

    rng = pyref.SplitMix64(4242)
    if kind == "dummy":
        n = 64
        r1cs = dummy_r1cs(rng.fr(), n)
    else:
        n = 16
        R = pyref.R_MOD
        z = [rng.fr() for _ in range(n)]
        mk = lambda: [[(rng.fr(), int(rng.next() % n)) for _ in range(1 + int(rng.next() % 3))] for _ in range(n)]
        a, b = mk(), mk()
        matvec = lambda rows: [sum(v * z[c] for v, c in row) % R for row in rows]
        za, zb = matvec(a), matvec(b)
        c = [[(za[i] * zb[i] % R * pow(z[i], -1, R) % R, i)] for i in range(n)]
        dev = lambda rows: [[(gm.fr.fr_from_int(v), col) for v, col in row] for row in rows]
        mats = [SparseMatrix.from_rows(dev(m), n) for m in (a, b, c)] + [SparseMatrix.from_rows(dev(m), n, transpose=True) for m in (a, b, c)]
        M = lambda v: oracle.fr_to_mont(oracle.ints_to_limbs(v, 4))
        r1cs = R1cs(*mats, gm.FrVec.from_host(M(z)), gm.FrVec.from_host(M(z[1:])), gm.FrVec.from_host(M(z[:1])))
    ck = CommitterKey.new(2 * n, 3, oracle.random_fr(4243, 1)[0])
    time_proof = Proof.new_time(r1cs, ck)
    stream = R1csStream(r1cs)
    old = S.SPACE_TIME_THRESHOLD
    S.SPACE_TIME_THRESHOLD = 3  # make the elastic provers spend rounds in the space prover at this size
    try:
        ck_stream = CommitterKeyStream.from_committer_key(ck)
        ck_stream.min_device_chunk = 1  # cut the streams literally every 20 / depth pairs
        space_proof = Proof.new_elastic(stream, ck_stream, 20)
        assert Proof.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 20).serialize_compressed() == space_proof.serialize_compressed()
    finally:
        S.SPACE_TIME_THRESHOLD = old
    eq = lambda x, y: bool((np.asarray(x) == np.asarray(y)).all())
    assert eq(time_proof.witness_commitment, space_proof.witness_commitment)
    assert eq(time_proof.zc_alpha, space_proof.zc_alpha)
    for tm, sm in ((time_proof.first_sumcheck_msgs, space_proof.first_sumcheck_msgs), (time_proof.second_sumcheck_msgs, space_proof.second_sumcheck_msgs)):
        assert len(tm[0]) == len(sm[0]) and all(eq(x[0], y[0]) and eq(x[1], y[1]) for x, y in zip(tm[0], sm[0]))
        assert eq(tm[1][0][0], sm[1][0][0]) and eq(tm[1][0][1], sm[1][0][1])
    tt, st = time_proof.tensorcheck_proof, space_proof.tensorcheck_proof
    assert len(tt.folded_polynomials_commitments) == len(st.folded_polynomials_commitments)
    assert all(eq(x, y) for x, y in zip(tt.folded_polynomials_commitments, st.folded_polynomials_commitments))
    assert all(eq(x, y) for x, y in zip(tt.folded_polynomials_evaluations, st.folded_polynomials_evaluations))
    assert all(eq(x, y) for x, y in zip(tt.base_polynomials_evaluations, st.base_polynomials_evaluations))
    assert eq(tt.evaluation_proof, st.evaluation_proof)
    stream.free()
    r1cs.free()


def test_sharded_committer_key_single_process(gm, oracle, pyref):
    """north star: the KZG key shards by powers across GPUs.  Here both shards live on the one GPU:
    each rank-local share (powers i = rank mod world) is generated from (tau^rank g, tau^world) and equals those powers
    of the full key; the
    partial commitments add up (gm_g1_sum) to the full key's commitment; with a world of one
    (gloo, single rank) Proof.new_time over the sharded key gives the byte-identical proof."""
    import os
    import socket

    import torch.distributed as dist

    from gemini_amd.circuit import dummy_r1cs
    from tests.stepwise.dist import ShardedCommitterKey
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.msm import g1_sum
    from gemini_amd.snark import Proof

    tau_l = oracle.random_fr(77, 1)[0]
    n = 1 << 10
    ck = CommitterKey.new(2 * n, 5, tau_l)
    full = ck.powers_of_g.download()
    shards = [ShardedCommitterKey.new(2 * n, 5, tau_l, r, 3) for r in range(3)]
    assert [len(s.powers_of_g) for s in shards] == [683, 683, 683] and sum(len(s.powers_of_g) for s in shards) == 2 * n + 1
    for s in shards:  # power i lives on rank i mod 3
        assert (s.powers_of_g.download() == full[s.global_indices()]).all()
    for m in (2 * n + 1, 1500, 700, 5, 0):
        poly = oracle.fr_to_mont(oracle.random_fr(100 + m, m)) if m else np.empty((0, 4), dtype=np.uint64)
        want = ck.commit(poly)
        got = g1_sum(np.stack([s.partial(poly) for s in shards]))
        assert (got == want).all(), m

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        one = ShardedCommitterKey.new(2 * n, 5, tau_l, 0, 1)
        r1cs = dummy_r1cs(12345, n)
        a = Proof.new_time(r1cs, ck).serialize_compressed()
        b = Proof.new_time(r1cs, one).serialize_compressed()
        assert a == b
    finally:
        dist.destroy_process_group()


def test_division_property_at_full_size(gm, oracle, pyref):
    """open_multi_points' quotient at n = 2^26 + 5 (more than 2^25 elements, i.e. > 2^19 blocked-scan chunks):
    f(beta) = q(beta) * prod (beta - p_j) + r(beta) with r the Newton-form remainder, at a random beta."""
    from gemini_amd.fr import FrVec, div_vanishing, evaluate_le, fr_from_int, fr_to_int, powers

    R = pyref.R_MOD
    n = (1 << 26) + 5
    x = oracle.limbs_to_ints(oracle.random_fr(4242, 5))
    f = powers(fr_from_int(x[0]), n)  # f_i = x0^i: a dense vector without a host upload
    pts = [x[1], x[1] * x[1] % R, (-x[1]) % R]
    q, rem = div_vanishing(f, np.stack([fr_from_int(p) for p in pts]))
    assert len(q) == n - 3
    beta = x[2]
    fb = fr_to_int(evaluate_le(f, fr_from_int(beta).reshape(1, 4))[0])
    assert fb == (pow(x[0] * beta % R, n, R) - 1) * pow(x[0] * beta - 1, -1, R) % R  # geometric sum: checks evaluate_le too
    qb = fr_to_int(evaluate_le(q, fr_from_int(beta).reshape(1, 4))[0])
    van, rb, basis = 1, 0, 1
    for p, r_j in zip(pts, rem):  # remainder in Newton form: sum_j r_j prod_{t<j} (x - p_t)
        rb = (rb + fr_to_int(r_j) * basis) % R
        basis = basis * (beta - p) % R
        van = van * (beta - p) % R
    assert fb == (qb * van + rb) % R
    # and the first remainder is f(p_0)
    assert fr_to_int(rem[0]) == (pow(x[0] * pts[0] % R, n, R) - 1) * pow(x[0] * pts[0] - 1, -1, R) % R
    f.free()
    q.free()


def test_sharded_stream_key_single_process(gm, oracle, pyref):
    """ShardedCommitterKeyStream on a world of one (gloo): the elastic proof over the sharded stream key equals
    the proof over the plain stream key; with three shards on the one GPU the per-rank stream partials
    (commit, open, open_multi_points) add up to the unsharded results."""
    import os
    import socket

    import torch.distributed as dist

    from tests.stepwise import dist as gd
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.msm import g1_sum
    from gemini_amd.snark import Proof

    tau_l = oracle.random_fr(78, 1)[0]
    n = 1 << 9
    ck = CommitterKey.new(2 * n, 5, tau_l)
    plain = CommitterKeyStream.from_committer_key(ck)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        # three shards in one process: replace the collective by "this rank only" and add the partials by hand
        shards = [gd.ShardedCommitterKeyStream.from_sharded_key(gd.ShardedCommitterKey.new(2 * n, 5, tau_l, r, 3)) for r in range(3)]
        poly = gm.FrVec.from_host(oracle.fr_to_mont(oracle.random_fr(79, 700)))
        alpha = oracle.fr_to_mont(oracle.random_fr(80, 1))[0]
        for sk in shards:
            sk.min_device_chunk = 64
        want_commit = plain.commit(poly)
        want_open = plain.open(poly, alpha, 1 << 10)
        # with world 1 the all-gather returns the rank's own partial: sum the three partials
        assert (g1_sum(np.stack([sk.commit(poly) for sk in shards])) == want_commit).all()
        got = [sk.open(poly, alpha, 100) for sk in shards]
        assert all((g[0] == want_open[0]).all() for g in got)
        assert (g1_sum(np.stack([g[1] for g in got])) == want_open[1]).all()
        one = gd.ShardedCommitterKeyStream.from_sharded_key(gd.ShardedCommitterKey.new(2 * n, 5, tau_l, 0, 1))
        r1cs = dummy_r1cs(4321, n)
        stream = R1csStream(r1cs)
        a = Proof.new_elastic(stream, plain, 1 << 20).serialize_compressed()
        b = Proof.new_elastic(stream, one, 1 << 20).serialize_compressed()
        assert a == b == Proof.new_time(r1cs, ck).serialize_compressed()
        stream.free()
        poly.free()
    finally:
        dist.destroy_process_group()


def test_snark_time_prover_full_size_closed_forms(gm, oracle, pyref):
    """BASELINE config 3 (`snark --time-prover -i 24`) at FULL size, through what the dummy instance fixes in
    closed form (src/circuit.rs:349-365: z = [e; n], w = [e; n-1], A = B = C = diag(1/e), so z_a = z_b = z_c =
    [1; n]):  commitment(w) = e (tau^(n-1) - 1)/(tau - 1) g;  zc(alpha) = (alpha^n - 1)/(alpha - 1) with alpha
    re-derived by the oracle's Merlin from that commitment;  the first message of the first sumcheck,
    a = sum_i alpha^(2i) = (alpha^n - 1)/(alpha^2 - 1), b = sum_i (1 + alpha) alpha^(2i) (time_prover.rs:105-118 on
    all-ones vectors).  Ties the 2^24-pair MSM, the transcript and the 2^24-element passes together."""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof

    R = pyref.R_MOD
    logn = 24
    n = 1 << logn
    e = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R
    tau = 0xFEDCBA9876543210FEDCBA9876543210FEDCBA98765432 % R
    r1cs = dummy_r1cs(e, n)
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    proof = Proof.new_time(r1cs, ck)
    inv = lambda v: pow(v % R, -1, R)
    k = e * (pow(tau, n - 1, R) - 1) % R * inv(tau - 1) % R
    want_cm = pyref.g1_mul(pyref.G1_GEN, k)
    assert jac_to_affine_ints(oracle, proof.witness_commitment) == want_cm
    tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    tr.append_message(b"witness", pyref.g1_serialize_uncompressed(want_cm))
    alpha = tr.get_challenge(b"alpha")
    I = gm.fr.fr_to_int
    assert I(proof.zc_alpha) == (pow(alpha, n, R) - 1) * inv(alpha - 1) % R
    a0 = (pow(alpha, n, R) - 1) * inv(alpha * alpha - 1) % R
    msgs = proof.first_sumcheck_msgs[0]
    assert len(msgs) == logn
    assert I(msgs[0][0]) == a0 and I(msgs[0][1]) == a0 * (1 + alpha) % R
    assert len(proof.tensorcheck_proof.folded_polynomials_commitments) == logn - 1
    assert proof.compressed_size() == 6056  # the size DESIGN.md derives for logN 24
    r1cs.free()
    ck.powers_of_g.free()  # 3.2 GB of powers + 38.7 GB of fixed-base tables


def test_matrix_tensor_known_answers(gm, oracle, pyref):
    """src/snark/streams.rs:104-222 (test_matrix_tensor_stream, test_matrix_tensor): the MatrixTensor stream
    M^T tensor(challenges) is a product with the transposed CSR matrix on the device.  Identities that hold for
    any r: diag(r) with the all-ones tensor gives [r; n]; with a random tensor the (big-endian) stream is
    [r t0 t1, r t1, r t0, r]; the 4x4 example gives [r^3 + r^2 + r + 1, r, r^2, 0]; the identity matrix
    returns tensor(challenges) itself."""
    from gemini_amd.circuit import SparseMatrix
    from gemini_amd.fr import FrVec, fr_from_int, fr_to_int, tensor

    R = pyref.R_MOD
    rng = pyref.SplitMix64(222)
    r, t0, t1 = rng.fr(), rng.fr(), rng.fr()
    M = lambda v: fr_from_int(v % R)
    ints = lambda vec: [fr_to_int(x) for x in vec.to_host()]
    diag = SparseMatrix.from_rows([[(M(r), i)] for i in range(4)], 4, transpose=True)
    ones = tensor(np.stack([M(1), M(1)]))
    assert ints(diag.mul(ones)) == [r] * 4
    tt = tensor(np.stack([M(t0), M(t1)]))
    assert ints(diag.mul(tt))[::-1] == [r * t0 * t1 % R, r * t1 % R, r * t0 % R, r]  # stream order = reversed vector
    # column-major example of test_matrix_tensor: column 0 holds rows 0..3, column 1 row 1, column 2 row 2, column 3 nothing
    rows = [[(M(1), 0)], [(M(1), 0), (M(1), 1)], [(M(1), 0), (M(1), 2)], [(M(1), 0)]]
    mt = SparseMatrix.from_rows(rows, 4, transpose=True)
    T = tensor(np.stack([M(r), M(r * r)]))
    assert ints(T) == [1, r, r * r % R, pow(r, 3, R)]
    assert ints(mt.mul(T)) == [(pow(r, 3, R) + r * r + r + 1) % R, r, r * r % R, 0]
    ch = [rng.fr() for _ in range(4)]
    ident = SparseMatrix.from_rows([[(M(1), i)] for i in range(16)], 16, transpose=True)
    Tc = tensor(np.stack([M(c) for c in ch]))
    assert ints(ident.mul(Tc)) == pyref.tensor(ch)
    for m in (diag, mt, ident):
        m.free()
    for v in (ones, tt, T, Tc):
        v.free()


def test_open_polynomials_no_longer_than_the_point_set(gm, oracle, pyref):
    """CommitterKey::open([c], x) returns c with the identity as proof (src/kzg/time.rs:112-131); the stream
    key's open / open_multi_points return the polynomial itself as remainder when it is shorter than the
    vanishing polynomial (src/kzg/space.rs:95-166).  The device division has no pass to run in that case and
    used to leave the remainders unwritten."""
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream

    I = gm.fr.fr_to_int
    M = lambda v: oracle.fr_to_mont(oracle.ints_to_limbs(v, 4))
    tau = oracle.limbs_to_ints(oracle.random_fr(171, 1))[0]
    time_ck = CommitterKey.new(16, 3, oracle.ints_to_limbs([tau], 4)[0])
    space_ck = CommitterKeyStream.from_committer_key(time_ck)
    x = 11
    ev, proof = time_ck.open(M([42]), M([x])[0])
    assert I(ev) == 42 and not np.asarray(proof).reshape(3, 6)[2].any()  # evaluation c, quotient 0 -> identity
    sev, sproof = space_ck.open(M([42]), M([x])[0], 1 << 20)
    assert I(sev) == 42 and (np.asarray(sproof) == np.asarray(proof)).all()
    # f(X) = 5 X^2 + 7 X + 9 (big-endian stream) against three points: remainder = f, quotient = 0
    pts = [3, 10, pyref.R_MOD - 10]
    rem, mp = space_ck.open_multi_points(M([5, 7, 9]), M(pts), 1 << 20)
    assert [I(r) for r in rem] == [5, 7, 9] and not np.asarray(mp).reshape(3, 6)[2].any()
    # ... and against two points: quotient 5, remainder f mod (X - 3)(X - 10) = (7 + 65) X + (9 - 150)
    rem2, mp2 = space_ck.open_multi_points(M([5, 7, 9]), M(pts[:2]), 1 << 20)
    assert [I(r) for r in rem2] == [72, (9 - 150) % pyref.R_MOD]
    assert (np.asarray(mp2) == np.asarray(time_ck.commit(M([5])))).all()
    assert (np.asarray(time_ck.open_multi_points(M([9, 7, 5]), M(pts[:2]))) == np.asarray(mp2)).all()


def test_snark_elastic_config4_shape_at_logn_22(gm, oracle, pyref):
    """BASELINE configs[3] (`examples/snark -i 28`, elastic prover) in its own shape at the largest size the suite
    budget allows: dummy_r1cs_stream(2^22), the GENERATOR-COPIES key of examples/snark.rs:59-63 (n + 1 copies of g),
    max_msm_buffer = 2^20 (:57) cut LITERALLY (min_device_chunk = 1: every ChunkedPippenger / HashMapPippenger flush of
    src/kzg/space.rs:105,139,205,244 is its own device MSM -- 2^20 / depth pairs in commit_folding), and the real
    SPACE_TIME_THRESHOLD = 22, so the space prover -> time prover hand-off of src/subprotocols/sumcheck/
    elastic_prover.rs:44-57 happens after the first round.  The proof must be byte-identical to Proof::new_time on the
    same key (`assert_eq!(time_proof, space_proof)`, src/snark/tests.rs:14-57) and to the default, merged-flush run."""
    from gemini_amd import sumcheck as S
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream, g1_generator_mont
    from gemini_amd.msm import G1Bases
    from gemini_amd.snark import Proof

    assert S.SPACE_TIME_THRESHOLD == 22
    logn = 22
    n = 1 << logn
    ones = np.zeros((n + 1, 4), dtype=np.uint64)
    ones[:, 0] = 1
    ck = CommitterKey(G1Bases.fixed_base(g1_generator_mont(), ones), 3)
    r1cs = dummy_r1cs(oracle.limbs_to_ints(oracle.random_fr(2228, 1))[0], n)
    time_bytes = Proof.new_time(r1cs, ck).serialize_compressed()
    stream = R1csStream(r1cs)
    literal = CommitterKeyStream.from_committer_key(ck, min_device_chunk=1)
    flushes = []
    inner = literal.powers_of_g.msm_vec

    def counting(*a, **k):
        flushes.append(k.get("n"))
        return inner(*a, **k)

    literal.powers_of_g.msm_vec = counting
    try:
        elastic = Proof.new_elastic(stream, literal, 1 << 20, native=False)  # (the Python-level flush counter sees the step-wise path only)
    finally:
        literal.powers_of_g.msm_vec = inner
    assert elastic.serialize_compressed() == time_bytes
    assert max(flushes) <= 1 << 20 and len(flushes) >= 100 and min(flushes) < 1 << 16, (len(flushes), max(flushes))  # the chunked paths really ran
    merged = Proof.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 1 << 20)
    assert merged.serialize_compressed() == time_bytes
    assert len(elastic.first_sumcheck_msgs[0]) == logn
    stream.free()
    r1cs.free()
    ck.powers_of_g.free()


def _elastic_closed_forms(gm, oracle, pyref, logn, native):
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream, g1_generator_mont
    from gemini_amd.msm import G1Bases
    from gemini_amd.snark import Proof

    R = pyref.R_MOD
    n = 1 << logn
    e = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F00112233445566778 % R
    ones = np.zeros((n + 1, 4), dtype=np.uint64)
    ones[:, 0] = 1
    ck = CommitterKey(G1Bases.fixed_base(g1_generator_mont(), ones), 3)
    del ones
    r1cs = dummy_r1cs(e, n)
    stream = R1csStream(r1cs)
    proof = Proof.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 1 << 20, native=native)
    inv = lambda v: pow(v % R, -1, R)
    want_cm = pyref.g1_mul(pyref.G1_GEN, e * (n - 1) % R)
    assert jac_to_affine_ints(oracle, proof.witness_commitment) == want_cm
    tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    tr.append_message(b"witness", pyref.g1_serialize_uncompressed(want_cm))
    alpha = tr.get_challenge(b"alpha")
    I = gm.fr.fr_to_int
    assert I(proof.zc_alpha) == (pow(alpha, n, R) - 1) * inv(alpha - 1) % R
    a0 = (pow(alpha, n, R) - 1) * inv(alpha * alpha - 1) % R
    msgs = proof.first_sumcheck_msgs[0]
    assert len(msgs) == logn
    assert I(msgs[0][0]) == a0 and I(msgs[0][1]) == a0 * (1 + alpha) % R
    assert len(proof.tensorcheck_proof.folded_polynomials_commitments) == logn - 1
    # ... and the WHOLE verifier of src/snark/verifier.rs:19-119 on this proof: the generator-copies key is the key of the
    # trapdoor tau = 1 (`powers_of_g2: vec![g2; 4]`, examples/snark.rs:63), and for dummy_r1cs the verifier's O(n) matrix
    # evaluations have an O(log n) closed form (oracle/verifier_ref.py::dummy_matrix_evaluations_closed_form, held equal to
    # the generic evaluation in tests/test_oracle_verifier.py) -- both sumcheck subclaims, the tensor relation of all
    # logn - 1 levels and the pairing check of the batched opening
    from oracle import verifier_ref as V
    from tests.util import snark_proof_to_ints

    V.snark_verify(snark_proof_to_ints(gm, oracle, proof), {"a": range(n), "x": [e]}, V.VerifierKey.from_trapdoor(1, 3),
                   m_of=V.dummy_matrix_evaluations_closed_form(e, n))
    size = proof.compressed_size()
    stream.free()
    r1cs.free()
    ck.powers_of_g.free()
    return size


def test_snark_elastic_config4_at_logn_26_closed_forms(gm, oracle, pyref):
    """BASELINE configs[3] two powers short of its own size: the elastic prover (Python-driven) on dummy_r1cs_stream(2^26) over
    the GENERATOR-COPIES key of examples/snark.rs:59-63, max_msm_buffer = 2^20 (:57), default flush merging.  Checked through
    what the instance fixes in closed form -- every base is g and w = [e; n - 1], so commitment(w) = e (n - 1) g; z_c = [1; n],
    so zc(alpha) = (alpha^n - 1) / (alpha - 1) with alpha re-derived by the oracle's Merlin from that commitment; the first
    sumcheck message on all-ones vectors -- and the space -> time hand-off after 4 rounds (SPACE_TIME_THRESHOLD = 22)."""
    _elastic_closed_forms(gm, oracle, pyref, 26, native=False)


def test_snark_elastic_config4_at_its_own_size_logn_28(gm, oracle, pyref):
    """BASELINE configs[3] AT ITS OWN SIZE: `examples/snark -i 28`, elastic prover (examples/snark.rs:54-66), on ONE MI355X
    (the config names 8; 288 GB hold it whole: 26 GB of key, 8.6 GB per vector) through the compiled driver
    gm_snark_new_elastic, the same closed forms as at 2^26 -- and the proof size the example prints (:96) follows from logn
    alone."""
    s28 = _elastic_closed_forms(gm, oracle, pyref, 28, native=True)
    assert s28 > 0


def test_stream_commit_crosses_the_merge_floor(gm, oracle):
    """a CommitterKeyStream that merges flushes shorter than 2^22 pairs (the default floor is 2^26) on a stream longer
    than the floor: 2^22 + 5 coefficients cross one chunk boundary of _msm_stream; time == stream (src/kzg/tests.rs:16-29)"""
    from gemini_amd.fr import FrVec, powers, fr_from_int
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream

    m = (1 << 22) + 5
    ck = CommitterKey.new(m, 3, oracle.random_fr(2229, 1)[0])
    assert CommitterKeyStream.from_committer_key(ck).min_device_chunk == CommitterKeyStream.DEFAULT_MIN_DEVICE_CHUNK == 1 << 26
    stream_ck = CommitterKeyStream.from_committer_key(ck, min_device_chunk=1 << 22)
    poly = powers(fr_from_int(oracle.limbs_to_ints(oracle.random_fr(2230, 1))[0]), m)  # dense, no host upload
    from gemini_amd.fr import reverse

    be = reverse(poly)
    assert (stream_ck.commit(be) == ck.commit(poly)).all()
    for v in (poly, be):
        v.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("logn", [3, 8, 12])
def test_device_proofs_are_accepted_by_the_reference_verifier(gm, oracle, pyref, logn):
    """src/snark/tests.rs:20-24, 42-46 (`proof.verify(&r1cs, &vk).is_ok()`): the proofs of the device provers, time and
    elastic, pass the reference's verification equations -- sumcheck subclaims, the tensor relation and the pairing check
    of the batched KZG opening against a key built from the trapdoor (oracle/verifier_ref.py, a restatement of the
    VERIFIER that never sees how the proof was made) -- and a proof with one altered element does not.  Unlike the
    comparison with the restated prover this holds at sizes the Python prover cannot reach."""
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr
    from oracle import verifier_ref as V
    from tests.util import snark_proof_to_ints

    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(3100 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(3200 + logn, 1))[0]
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    vk = V.VerifierKey.from_trapdoor(tau, 5)
    # the device key's G2 half is the verifier's (src/kzg/time.rs:29-40)
    assert ck.powers_of_g2 == vk.powers_of_g2[: len(ck.powers_of_g2)]
    r1cs = dummy_r1cs(e, n)
    inst = sr.dummy_r1cs(e, n)
    proof = Proof.new_time(r1cs, ck)
    ints = snark_proof_to_ints(gm, oracle, proof)
    V.snark_verify(ints, inst, vk)
    if logn <= 8:
        stream = R1csStream(r1cs)
        elastic = Proof.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 1 << 5)
        V.snark_verify(snark_proof_to_ints(gm, oracle, elastic), inst, vk)
        stream.free()
    bad = dict(ints)
    bad["tensorcheck_proof"] = dict(ints["tensorcheck_proof"])
    fe = [list(x) for x in ints["tensorcheck_proof"]["folded_polynomials_evaluations"]]
    fe[-1][1] = (fe[-1][1] + 1) % pyref.R_MOD
    bad["tensorcheck_proof"]["folded_polynomials_evaluations"] = fe
    with pytest.raises(V.VerificationError):
        V.snark_verify(bad, inst, vk)
    bad = dict(ints)
    bad["witness_commitment"] = pyref.g1_add(ints["witness_commitment"], ints["witness_commitment"])
    with pytest.raises(V.VerificationError):
        V.snark_verify(bad, inst, vk)
    r1cs.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("logn", [16, 20])
def test_device_proof_of_2p16_and_2p20_constraints_is_accepted_by_the_reference_verifier(gm, oracle, pyref, logn):
    """the O(n) verifier of the non-preprocessing SNARK on 2^16- and 2^20-constraint device proofs (BASELINE configs[2]
    shape, smaller; the restated PROVER stops at 2^9)"""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr
    from oracle import verifier_ref as V
    from tests.util import snark_proof_to_ints

    n = 1 << logn
    e, tau = 0x1D2C3B4A59687766554433221100FFEE % pyref.R_MOD, 0x0123456789ABCDEF0FEDCBA987654321 % pyref.R_MOD
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    r1cs = dummy_r1cs(e, n)
    proof = Proof.new_time(r1cs, ck)
    V.snark_verify(snark_proof_to_ints(gm, oracle, proof), sr.dummy_r1cs(e, n), V.VerifierKey.from_trapdoor(tau, 5))
    r1cs.free()
    ck.powers_of_g.free()


def test_stream_commit_default_settings_cross_the_default_floor(gm, oracle):
    """the DEFAULT CommitterKeyStream (flushes merged up to 2^26 pairs) on a stream longer than that: 2^26 + 5 coefficients
    cross one chunk boundary of _msm_stream with nothing overridden; time == stream (src/kzg/tests.rs:16-29)"""
    from gemini_amd.fr import fr_from_int, powers, reverse
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream

    m = (1 << 26) + 5
    ck = CommitterKey.new(m, 3, oracle.random_fr(2231, 1)[0])
    stream_ck = CommitterKeyStream.from_committer_key(ck)
    assert stream_ck.min_device_chunk == 1 << 26
    poly = powers(fr_from_int(oracle.limbs_to_ints(oracle.random_fr(2232, 1))[0]), m)  # dense, no host upload
    be = reverse(poly)
    assert (stream_ck.commit(be) == ck.commit(poly)).all()
    for v in (poly, be):
        v.free()
    ck.powers_of_g.free()


def test_device_openings_pass_the_pairing_checks(gm, oracle, pyref):
    """src/kzg/time.rs:193-211 (open at a random point, `vk.verify` accepts), src/kzg/tests.rs:43-59 (the same through the
    stream key) and src/kzg/tests.rs:62-102 (test_open_multipoints_correctness: 15 polynomials of degree 100, five points,
    batch_commit + batch_open_multi_points, `verify_multi_points` accepts) on the device committer, checked with the
    restated pairing verifier (oracle/verifier_ref.py)."""
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from oracle import verifier_ref as V

    I = gm.fr.fr_to_int
    R = pyref.R_MOD
    d = 100
    tau = oracle.limbs_to_ints(oracle.random_fr(5100, 1))[0]
    pts = oracle.limbs_to_ints(oracle.random_fr(5101, 5))
    polys = [oracle.limbs_to_ints(oracle.random_fr(5110 + k, d + 1)) for k in range(15)]
    ck = CommitterKey.new(d + 1, len(pts), oracle.ints_to_limbs([tau], 4)[0])
    vk = V.VerifierKey.from_trapdoor(tau, len(pts))
    A = lambda p: jac_to_affine_ints(oracle, p)  # noqa: E731
    dev = [_M(oracle, p) for p in polys]
    comms = [A(c) for c in ck.batch_commit(dev)]
    eta = oracle.limbs_to_ints(oracle.random_fr(5102, 1))[0] % (1 << 128)  # `u128::rand(..).into()`
    proof = A(ck.batch_open_multi_points(dev, _M(oracle, pts), _M(oracle, [eta])[0]))
    evals = [[pyref.evaluate_le(p, x) for x in pts] for p in polys]
    V.verify_multi_points(vk, comms, pts, evals, proof, eta)
    evals[7][3] = (evals[7][3] + 1) % R
    with pytest.raises(V.VerificationError):
        V.verify_multi_points(vk, comms, pts, evals, proof, eta)
    # single point, time key and stream key
    alpha = pts[0]
    ev, pr = ck.open(dev[0], _M(oracle, [alpha])[0])
    assert I(ev) == pyref.evaluate_le(polys[0], alpha)
    V.verify(vk, comms[0], alpha, I(ev), A(pr))
    with pytest.raises(V.VerificationError):
        V.verify(vk, comms[0], alpha, (I(ev) + 1) % R, A(pr))
    stream_ck = CommitterKeyStream.from_committer_key(ck)
    be = dev[0][::-1].copy()
    ev2, pr2 = stream_ck.open(be, _M(oracle, [alpha])[0], 16)
    assert I(ev2) == I(ev)
    V.verify(vk, A(stream_ck.commit(be)), alpha, I(ev2), A(pr2))
    ck.powers_of_g.free()


def test_baseline_config_proof_is_accepted_by_the_reference_verifier(gm, oracle, pyref):
    """BASELINE configs[2], `snark --time-prover -i 24` (examples/snark.rs:69-79): the device proof of dummy_r1cs(2^24) is
    ACCEPTED by the restated verifier of src/snark/verifier.rs -- both sumcheck subclaims, the tensor relation and the
    pairing check of the batched opening against the trapdoor key; the O(n) evaluation of the matrices at the powers of
    +-beta runs in the C restatement (oracle/verifier_ref.py::dummy_matrix_evaluations)."""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof
    from oracle import verifier_ref as V
    from tests.util import snark_proof_to_ints

    n = 1 << 24
    e, tau = 0x1D2C3B4A59687766554433221100FFEE % pyref.R_MOD, 0x0123456789ABCDEF0FEDCBA987654321 % pyref.R_MOD
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    r1cs = dummy_r1cs(e, n)
    proof = Proof.new_time(r1cs, ck)
    ints = snark_proof_to_ints(gm, oracle, proof)
    r1cs.free()
    ck.powers_of_g.free()
    V.snark_verify(ints, {"a": range(n), "x": [e]}, V.VerifierKey.from_trapdoor(tau, 5), m_of=V.dummy_matrix_evaluations(e, n))


@pytest.mark.parametrize("logn", [1, 3, 6, 9, 14])
def test_native_prover_equals_the_stepwise_one(gm, oracle, pyref, logn):
    """gm_snark_new_time (the orchestration of src/snark/time_prover.rs:19-117 compiled into the library,
    gemini_amd/csrc/snark.cpp) against the step-by-step driver of gemini_amd/snark.py: the same proof, byte for byte,
    on the dummy instance and on a general sparse one; and the restatement's proof at the sizes it reaches."""
    from gemini_amd.circuit import R1cs, SparseMatrix, dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr
    from oracle import wire_ref as W
    from tests.util import random_r1cs_instance

    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(6100 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(6200 + logn, 1))[0]
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    r1cs = dummy_r1cs(e, n)
    stepwise = Proof.new_time(r1cs, ck, native=False)
    native = Proof.new_time(r1cs, ck, native=True)
    assert native == stepwise
    for compress in (True, False):
        assert native.serialize(compress, 0) == stepwise.serialize(compress, 0)
    assert set(native.spans) == set(stepwise.spans)
    if logn <= 9 and logn >= 3:
        exp = sr.snark_new_time(sr.dummy_r1cs(e, n), sr.srs(tau, 2 * n + 1))
        assert native.serialize(True, 0) == W.snark_proof(exp, True, "arkworks")
    r1cs.free()
    if 3 <= logn <= 6:  # a general instance (distinct sparse A, B, C; public input of two elements)
        inst, _ = random_r1cs_instance(pyref, sr, n, 6300 + logn, nx=2)
        M = lambda v: gm.fr.fr_from_int(v)  # noqa: E731
        dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]  # noqa: E731
        mats = [SparseMatrix.from_rows(dev(inst[k]), n) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), n, transpose=True) for k in "abc"]
        g = R1cs(*mats, gm.FrVec.from_host(_M(oracle, inst["z"])), gm.FrVec.from_host(_M(oracle, inst["w"])), gm.FrVec.from_host(_M(oracle, inst["x"])))
        p1, p2 = Proof.new_time(g, ck, native=False), Proof.new_time(g, ck, native=True)
        assert p1 == p2 and p2.serialize(True, 0) == W.snark_proof(sr.snark_new_time(inst, sr.srs(tau, 2 * n + 1)), True, "arkworks")
        g.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("logn", [1, 3, 6, 9, 14, 20, 23])
def test_native_elastic_prover_equals_the_stepwise_one_and_the_time_prover(gm, oracle, pyref, logn):
    """gm_snark_new_elastic (src/snark/elastic_prover.rs:174-266 compiled into the library, gemini_amd/csrc/snark.cpp) against
    the step-by-step elastic driver of gemini_amd/snark.py and against Proof::new_time on the same key
    (`assert_eq!(time_proof, space_proof)`, src/snark/tests.rs:14-57), byte for byte: on the dummy instance, on a general
    sparse one, with merged flushes and with max_msm_buffer cut literally (min_device_chunk = 1), below and above
    SPACE_TIME_THRESHOLD = 22 rounds (2^23: the space prover runs the first two rounds, then hands over)."""
    from gemini_amd.circuit import R1cs, R1csStream, SparseMatrix, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr
    from tests.util import random_r1cs_instance

    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(6400 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(6500 + logn, 1))[0]
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    r1cs = dummy_r1cs(e, n)
    want = Proof.new_time(r1cs, ck, native=True).serialize_compressed()
    stream = R1csStream(r1cs)
    merged = CommitterKeyStream.from_committer_key(ck)
    native = Proof.new_elastic(stream, merged, 1 << 20, native=True)
    assert native.serialize_compressed() == want
    assert len(native.first_sumcheck_msgs[0]) == logn
    if logn <= 20:
        stepwise = Proof.new_elastic(stream, merged, 1 << 20, native=False)
        assert native == stepwise and native.serialize_uncompressed() == stepwise.serialize_uncompressed()
        literal = CommitterKeyStream.from_committer_key(ck, min_device_chunk=1)
        assert Proof.new_elastic(stream, literal, 1 << (10 if logn > 10 else 2), native=True).serialize_compressed() == want
    stream.free()
    r1cs.free()
    if 3 <= logn <= 6:  # a general instance (distinct sparse A, B, C; public input of two elements)
        inst, _ = random_r1cs_instance(pyref, sr, n, 6600 + logn, nx=2)
        M = lambda v: gm.fr.fr_from_int(v)  # noqa: E731
        dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]  # noqa: E731
        mats = [SparseMatrix.from_rows(dev(inst[k]), n) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), n, transpose=True) for k in "abc"]
        g = R1cs(*mats, gm.FrVec.from_host(_M(oracle, inst["z"])), gm.FrVec.from_host(_M(oracle, inst["w"])), gm.FrVec.from_host(_M(oracle, inst["x"])))
        gs = R1csStream(g)
        assert Proof.new_elastic(gs, merged, 1 << 20, native=True).serialize_compressed() == Proof.new_time(g, ck, native=False).serialize_compressed()
        gs.free()
        g.free()
    ck.powers_of_g.free()

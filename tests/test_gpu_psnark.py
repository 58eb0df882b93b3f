"""GPU parity for SURVEY.md section 8f rank 4: the preprocessing SNARK time prover
(src/psnark/time_prover.rs) with its entry-product / plookup builders, through the device path vs the
CPU restatement (oracle/psnark_ref.py), transcript included."""
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
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4)) if len(ints) else np.empty((0, 4), dtype=np.uint64)


def test_plookup_entryproduct_builders(gm, oracle, pyref):
    """lookup / alg_hash / plookup_set / plookup_subset / sorted / accumulated_product / right_rotation
    (plookup/time_prover.rs, entryproduct/time_prover.rs) incl. the reference's own relation tests
    (plookup/time_prover.rs:114-148 test_plookup_relation, :37-60 test_plookup_set_correct)."""
    from gemini_amd import fr as F
    from gemini_amd import psnark as ps
    from oracle import psnark_ref as pr

    R = pyref.R_MOD
    I = lambda v: oracle.limbs_to_ints(oracle.fr_from_mont(v.to_host() if hasattr(v, "to_host") else np.asarray(v).reshape(-1, 4)))
    for n, m, seed in ((6, 4, 1), (1, 1, 2), (64, 200, 3), (5000, 9000, 4), (300000, 100000, 5)):
        rng = np.random.default_rng(seed)
        set_ = oracle.limbs_to_ints(oracle.random_fr(10 * seed, n))
        index = rng.integers(0, n, size=m).astype(np.uint32)
        y, z, zeta = oracle.limbs_to_ints(oracle.random_fr(10 * seed + 1, 3))
        dset = F.FrVec.from_host(_M(oracle, set_))
        didx = F.IdxVec.from_host(index)
        sub = F.lookup(dset, didx)
        subset = pr.lookup(set_, index.tolist())
        assert I(sub) == subset
        assert I(F.alg_hash(dset, None, _M(oracle, [zeta])[0])) == pr.alg_hash(set_, range(n), zeta)
        assert I(F.alg_hash(sub, didx, _M(oracle, [zeta])[0])) == pr.alg_hash(subset, index.tolist(), zeta)
        assert I(F.plookup_set(dset, _M(oracle, [y])[0], _M(oracle, [z])[0])) == pr.plookup_set(set_, y, z)
        assert I(F.plookup_subset(dset, _M(oracle, [y])[0])) == pr.plookup_subset(set_, y)
        assert I(F.shift_monic(dset)) == pr.right_rotation(pr.monic(set_))
        acc = F.accumulated_product_monic(dset)
        assert I(acc) == pr.accumulated_product(pr.monic(set_))
        freq = ps.compute_frequency(n, index)
        assert freq.tolist() == pr.compute_frequency(n, index.tolist())
        ext = ps.extend_frequency(freq)
        assert ext.tolist() == pr.extend_frequency(freq.tolist())
        dext = F.IdxVec.from_host(ext)
        assert I(F.lookup(dset, dext)) == pr.sorted_(set_, freq.tolist())
        for zt in (zeta, 0):
            got = ps.plookup(sub, dset, didx, dext, _M(oracle, [y])[0], _M(oracle, [z])[0], _M(oracle, [zt])[0])
            want = pr.plookup(subset, set_, index.tolist(), y, z, zt)
            assert [I(v) for v in got] == want
            # the plookup relation: prod(sorted) = prod(set) prod(subset) (1+z)^|subset|
            prods = [I(F.accumulated_product_monic(v))[0] for v in got]
            assert prods[2] == prods[0] * prods[1] % R * pow(1 + z, m, R) % R
    # empty inputs
    e = F.FrVec.alloc(0)
    assert len(F.plookup_set(e, _M(oracle, [1])[0], _M(oracle, [2])[0])) == 0
    assert I(F.accumulated_product_monic(e)) == [1] and I(F.shift_monic(e)) == [1]


def _random_instance(pyref, sr, n, seed):
    from tests.util import random_r1cs_instance

    return random_r1cs_instance(pyref, sr, n, seed)


def _device_instance(gm, oracle, inst, n):
    from gemini_amd.circuit import R1cs, SparseMatrix

    M = lambda v: gm.fr.fr_from_int(v)
    dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]
    mats = [SparseMatrix.from_rows(dev(inst[k]), n) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), n, transpose=True) for k in "abc"]
    return R1cs(*mats, gm.FrVec.from_host(_M(oracle, inst["z"])), gm.FrVec.from_host(_M(oracle, inst["w"])), gm.FrVec.from_host(_M(oracle, inst["x"])))


def _check_proof(gm, oracle, proof, exp):
    I = gm.fr.fr_to_int
    A = lambda j: jac_to_affine_ints(oracle, j)
    assert A(proof.witness_commitment) == exp["witness_commitment"]
    assert I(proof.zc_alpha) == exp["zc_alpha"]
    for name in ("first_sumcheck_msgs", "second_sumcheck_msgs", "third_sumcheck_msgs"):
        msgs, finals = getattr(proof, name)
        assert [(I(x), I(y)) for x, y in msgs] == exp[name][0], name
        want = exp[name][1]
        want = [tuple(want)] if isinstance(want[0], int) else [tuple(f) for f in want]  # sumcheck_prove returns one pair
        assert [(I(x), I(y)) for x, y in finals] == want, name
    assert [A(c) for c in proof.r_star_commitments] == exp["r_star_commitments"]
    assert A(proof.z_star_commitment) == exp["z_star_commitment"]
    for k in ("set_r_ep", "subset_r_ep", "set_alpha_ep", "subset_alpha_ep", "set_z_ep", "subset_z_ep"):
        assert I(getattr(proof, k)) == exp[k], k
    for k in ("sorted_r_commitment", "sorted_alpha_commitment", "sorted_z_commitment", "ralpha_star_acc_mu_proof"):
        assert A(getattr(proof, k)) == exp[k], k
    assert [A(c) for c in proof.ep_msgs.acc_v_commitments] == exp["ep_msgs"]["acc_v_commitments"]
    assert [I(e) for e in proof.ep_msgs.claimed_sumchecks] == exp["ep_msgs"]["claimed_sumchecks"]
    assert [I(e) for e in proof.ralpha_star_acc_mu_evals] == exp["ralpha_star_acc_mu_evals"]
    assert [I(e) for e in proof.rstars_vals] == exp["rstars_vals"]
    tc, etc = proof.tensorcheck_proof, exp["tensorcheck_proof"]
    assert [A(c) for c in tc.folded_polynomials_commitments] == etc["folded_polynomials_commitments"]
    assert [[I(e) for e in e2] for e2 in tc.folded_polynomials_evaluations] == etc["folded_polynomials_evaluations"]
    assert [[I(e) for e in e3] for e3 in tc.base_polynomials_evaluations] == etc["base_polynomials_evaluations"]
    assert A(tc.evaluation_proof) == etc["evaluation_proof"]


@pytest.mark.parametrize("n,seed", [(8, 78), (16, 79)])
def test_psnark_time_prover_random_r1cs(gm, oracle, pyref, n, seed):
    """src/psnark/tests.rs shape (random circuit, ck of num_constraints * 100 + num_variables powers):
    every field of the proof against the restatement."""
    from gemini_amd import g2
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr
    from oracle import snark_ref as sr

    inst, tau = _random_instance(pyref, sr, n, seed)
    srs = sr.srs(tau, 12 * n + 1)
    g2p = pr.powers_of_g2(tau, 3)
    exp_index = pr.index(srs, inst)
    exp = pr.psnark_new_time(srs, g2p, inst, exp_index)

    r1cs = _device_instance(gm, oracle, inst, n)
    ck = CommitterKey.new(12 * n, 3, oracle.ints_to_limbs([tau], 4)[0])
    assert ck.powers_of_g2 == g2p  # two independent G2 implementations
    assert ck.powers_of_g2_bytes() == len(g2p).to_bytes(8, "little") + b"".join(pr.g2_serialize_uncompressed(p) for p in g2p)
    index = Proof.index(ck, r1cs)
    assert [jac_to_affine_ints(oracle, c) for c in index] == exp_index
    proof = Proof.new_time(ck, r1cs, index)
    _check_proof(gm, oracle, proof, exp)
    native = Proof.new_time(ck, r1cs, index, native=True)  # gm_psnark_new_time: the same sequence compiled into the library
    assert native == proof and native.serialize_uncompressed() == proof.serialize_uncompressed()
    # serialized size: fixed-size fields + the vectors (src/psnark/mod.rs:29-51)
    blob = proof.serialize_compressed()
    n_msgs = sum(len(getattr(proof, k)[0]) for k in ("first_sumcheck_msgs", "second_sumcheck_msgs", "third_sumcheck_msgs"))
    assert len(blob) == proof.compressed_size() and len(blob) > 48 * 10 + 64 * n_msgs
    # byte equality with the oracle-side serialisation of the restatement's proof, every mode; round trip
    from oracle import wire_ref as W

    for compress in (True, False):
        for enc, mode in ((0, "arkworks"), (1, "zcash")):
            b2 = proof.serialize(compress, enc)
            assert b2 == W.psnark_proof(exp, compress, mode)
            assert Proof.deserialize(b2, compress, enc, validate=False) == proof
    assert blob == W.psnark_proof(exp, True, "arkworks")
    r1cs.free()


def test_psnark_time_prover_dummy_r1cs(gm, oracle, pyref):
    """examples/psnark.rs:70-81 recipe at a small size: dummy_r1cs, ck of num_constraints + num_variables"""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr
    from oracle import snark_ref as sr

    n = 16
    e = 987654321987654321
    tau = 1234567890123456789012345
    srs = sr.srs(tau, 2 * n + 1)
    inst = sr.dummy_r1cs(e, n)
    g2p = pr.powers_of_g2(tau, 5)
    exp_index = pr.index(srs, inst)
    exp = pr.psnark_new_time(srs, g2p, inst, exp_index)
    r1cs = dummy_r1cs(e, n)
    ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    index = Proof.index(ck, r1cs)
    proof = Proof.new_time(ck, r1cs, index)
    _check_proof(gm, oracle, proof, exp)
    r1cs.free()


@pytest.mark.parametrize("kind", ["random", "dummy"])
def test_psnark_consistency_time_vs_elastic(gm, oracle, pyref, kind):
    """src/psnark/tests.rs:14-125 test_consistency: `assert!(elastic_proof == time_proof)`"""
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.psnark import Proof
    from oracle import snark_ref as sr

    n = 32
    if kind == "random":
        inst, tau = _random_instance(pyref, sr, n, 91)
        r1cs = _device_instance(gm, oracle, inst, n)
    else:
        r1cs, tau = dummy_r1cs(424242, n), 31337
    ck = CommitterKey.new(n * 100 + n, 3, oracle.ints_to_limbs([tau], 4)[0])
    ck_stream = CommitterKeyStream.from_committer_key(ck)
    index = Proof.index(ck, r1cs)
    time_proof = Proof.new_time(ck, r1cs, index)
    stream = R1csStream(r1cs)
    elastic_proof = Proof.new_elastic(ck_stream, stream, index, 1 << 20)
    a, b = time_proof.serialize_compressed(), elastic_proof.serialize_compressed()
    for f in ("witness_commitment", "z_star_commitment", "sorted_r_commitment", "ralpha_star_acc_mu_proof"):
        assert (getattr(time_proof, f) == getattr(elastic_proof, f)).all(), f
    I = gm.fr.fr_to_int
    assert [(I(x), I(y)) for x, y in time_proof.third_sumcheck_msgs[0]] == [(I(x), I(y)) for x, y in elastic_proof.third_sumcheck_msgs[0]]
    assert (time_proof.tensorcheck_proof.evaluation_proof == elastic_proof.tensorcheck_proof.evaluation_proof).all()
    assert a == b
    # gm_psnark_new_elastic: the same prover compiled into the library (one call per proof)
    native = Proof.new_elastic(ck_stream, stream, index, 1 << 20, native=True)
    assert native == time_proof and native.serialize_compressed() == a
    assert "ark_gemini::psnark::elastic_prover" in native.spans
    # a small MSM buffer changes the chunking only
    ck_stream.min_device_chunk = 1
    assert Proof.new_elastic(ck_stream, stream, index, 1 << 6).serialize_compressed() == a
    assert Proof.new_elastic(ck_stream, stream, index, 1 << 6, native=True).serialize_compressed() == a  # every flush literal, down to 64 / depth pairs
    assert Proof.new_elastic(ck_stream, stream, index, 1, native=True).serialize_compressed() == a
    stream.free()
    r1cs.free()


@pytest.mark.parametrize("n", [1000, (1 << 20) + 5])
def test_entry_product_relation(gm, oracle, pyref, n):
    """src/subprotocols/entryproduct/tests.rs:14-36: for any v, with g = accumulated_product(monic(v)) and psi,
    <(right_rotation(monic(v)) - psi) o powers(psi), g> = prod(v) - psi^(n+1).  Runs the reverse prefix-product
    scan, the shift, hadamard / powers / ip at n = 1000 (the reference's size) and beyond 2^20."""
    from gemini_amd import fr as F

    R = pyref.R_MOD
    rng = np.random.default_rng(n)
    host = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    host[:, 3] &= np.uint64((1 << 62) - 1)
    v = F.FrVec.from_host(host)  # arbitrary Montgomery residues
    chal = oracle.limbs_to_ints(oracle.random_fr(n % 97, 1))[0]
    acc = F.accumulated_product_monic(v)
    rrot = F.shift_monic(v)
    shifted = F.plookup_subset(rrot, F.fr_from_int((-chal) % R))  # every entry minus psi
    twist = F.powers(F.fr_from_int(chal), n + 1)
    had = F.hadamard(shifted, twist)
    lhs = F.fr_to_int(F.ip(had, acc))
    product = F.fr_to_int(F.element(acc, 0))
    assert lhs == (product - pow(chal, n + 1, R)) % R
    if n <= 1000:  # the product itself against Python integers
        want = 1
        for x in oracle.limbs_to_ints(oracle.fr_from_mont(host)):
            want = want * x % R
        assert product == want
    for x in (v, acc, rrot, shifted, twist, had):
        x.free()


def test_entry_product_consistency(gm, oracle, pyref):
    """entryproduct/tests.rs:38-54: EntryProduct::new_time and new_elastic produce the same messages (and,
    here, the same sumcheck transcripts) on v = [r; 1000]."""
    from gemini_amd import fr as F
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from tests.stepwise.psnark_steps import EntryProduct
    from gemini_amd.sumcheck import Sumcheck
    from gemini_amd.transcript import Transcript

    R = pyref.R_MOD
    n = 1000
    r = oracle.limbs_to_ints(oracle.random_fr(55, 1))[0]
    v = F.FrVec.alloc(n)
    v.fill(F.fr_from_int(r))
    product = F.fr_from_int(pow(r, n, R))
    ck = CommitterKey.new(n + 1, 1, oracle.random_fr(56, 1)[0])
    t1, t2 = Transcript(b"test"), Transcript(b"test")
    ep_time = EntryProduct.new_time(t1, ck, v, product)
    v_stream = F.reverse(v)
    ep_space = EntryProduct.new_elastic(t2, CommitterKeyStream.from_committer_key(ck), v_stream, product)
    assert (ep_time.msgs.acc_v_commitments[0] == ep_space.msgs.acc_v_commitments[0]).all()
    assert (ep_time.msgs.claimed_sumchecks[0] == ep_space.msgs.claimed_sumchecks[0]).all()
    assert (ep_time.chal == ep_space.chal).all()
    a = Sumcheck.prove(t1, ep_time.provers[0])
    b = Sumcheck.prove(t2, ep_space.provers[0])
    I = F.fr_to_int
    assert [(I(x), I(y)) for x, y in a.messages] == [(I(x), I(y)) for x, y in b.messages]
    # the claimed sumcheck is the first message's a + b? no: it is the inner product <acc o twist, rrot>: check it
    acc = F.accumulated_product_monic(v)
    rrot = F.shift_monic(v)
    tw = F.powers(ep_time.chal, len(acc))
    had = F.hadamard(acc, tw)
    assert I(F.ip(had, rrot)) == I(ep_time.msgs.claimed_sumchecks[0])
    for p in ep_time.provers + ep_space.provers:
        p.free()
    for x in (v, v_stream, acc, rrot, tw, had):
        x.free()


@pytest.mark.parametrize("n", [147, 1 << 21])
def test_evaluate_index_poly(gm, oracle, pyref, n):
    """src/misc.rs:424-437 (test_evaluate_index_poly): the `row` / `col` style vector [F::from(i)] evaluated on
    the device equals the closed form x (1 - x^(n-1)) / (1 - x)^2 - (n - 1) x^(n-1) x / (1 - x) (:394-399)."""
    from gemini_amd import fr as F
    from gemini_amd.psnark import _field_of_index

    R = pyref.R_MOD
    x = oracle.limbs_to_ints(oracle.random_fr(n % 89, 1))[0]
    idx = F.IdxVec.from_host(np.arange(n, dtype=np.uint32))
    vec = _field_of_index(idx)
    got = F.fr_to_int(F.evaluate_le(vec, F.fr_from_int(x).reshape(1, 4))[0])
    inv = lambda v: pow(v % R, -1, R)
    x1 = (1 - x) % R
    xn = pow(x, n - 1, R)
    want = (x * (1 - xn) % R * inv(x1 * x1) - (n - 1) * xn % R * x % R * inv(x1)) % R
    assert got == want
    assert F.fr_to_int(F.element(vec, n - 1)) == n - 1
    idx.free()
    vec.free()


@pytest.mark.parametrize("logn", [18, 20, 22, 26])
def test_psnark_config5_shape_time_equals_elastic(gm, oracle, pyref, logn):
    """BASELINE configs[4] (`examples/psnark -i 26`) in its own shape at a suite-sized instance: dummy_r1cs(2^logn) and
    the recipe of examples/psnark.rs:70-81 (key of num_constraints + num_variables powers, index from the key).  The
    elastic prover on the stream view of the same key, max_msm_buffer = 2^20 as in :52, must return the byte-identical
    proof (`assert!(elastic_proof == time_proof)`, src/psnark/tests.rs:124), and what the dummy instance fixes in
    closed form must hold on the device proof: z_a = z_b = z_c = [1; n] (src/circuit.rs:349-365), so
    zc(alpha) = (alpha^n - 1) / (alpha - 1) with alpha re-derived by the oracle's transcript from the proof's own
    witness commitment and the key's G2 powers.

    logn = 26 is BASELINE configs[4] AT ITS OWN SIZE, as the example runs it without --time-prover: Proof::new_elastic on
    dummy_r1cs_stream(2^26) over the 3 * 2^26 + 1-point stream key, max_msm_buffer 2^20 (examples/psnark.rs:54-68) -- through
    gm_psnark_new_elastic, and the compiled gm_psnark_new_time beside it (both ~5 s, ~250-270 GB of device memory in use at the
    peak: profiles/r5_prover_sweep.txt); the two proofs must be the same bytes."""
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr

    R = pyref.R_MOD
    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(2600 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(2700 + logn, 1))[0]
    # 3 n + 1 powers: the stream key of the elastic example (examples/psnark.rs:61); the streaming committer insists on
    # a key at least as long as every stream (src/kzg/space.rs:169-175), the time committer truncates silently
    ck = CommitterKey.new(3 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
    r1cs = dummy_r1cs(e, n)
    index = Proof.index(ck, r1cs)
    # (2^26: the step-by-step Python driver keeps its temporaries alive longer than the compiled prover -- ~150 GB more)
    time_proof = Proof.new_time(ck, r1cs, index, native=logn >= 24)
    stream = R1csStream(r1cs)
    if logn >= 24:  # what the library promised (gm_psnark_footprint) against what the proof then used
        before = gm.capi.mem_stats()
        promised = gm.capi.psnark_footprint(ck.powers_of_g.handle, n, n, 1)
        gm.capi.mem_reset_peak()
    # 2^18: the Python-driven elastic prover beside the compiled one; above: the compiled one (gm_psnark_new_elastic) alone
    if logn == 18:
        stepwise = Proof.new_elastic(CommitterKeyStream.from_committer_key(ck), stream, index, 1 << 20, native=False)
        assert stepwise == time_proof and stepwise.serialize_compressed() == time_proof.serialize_compressed()
    elastic_proof = Proof.new_elastic(CommitterKeyStream.from_committer_key(ck), stream, index, 1 << 20, native=True)
    assert elastic_proof == time_proof and elastic_proof.serialize_compressed() == time_proof.serialize_compressed()
    if logn >= 24:
        after = gm.capi.mem_stats()
        used = after["in_use_peak"] - before["in_use"]
        assert used <= promised["needed"], (used, promised)
        assert after["spare_table_releases"] == before["spare_table_releases"]  # it fitted as promised: nothing was dropped on the way
    if logn == 22:  # the flushes cut literally: 2^20-pair stream MSMs (and 2^20 / depth in commit_folding), sumchecks that start as space provers
        literal = Proof.new_elastic(CommitterKeyStream.from_committer_key(ck, min_device_chunk=1), stream, index, 1 << 20, native=True)
        assert literal.serialize_compressed() == time_proof.serialize_compressed()
    # closed forms
    I = gm.fr.fr_to_int
    tr = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    g2p = pr.powers_of_g2(tau, 5)
    w_aff = jac_to_affine_ints(oracle, time_proof.witness_commitment)
    # commitment(w) = e (tau^(n-1) - 1) / (tau - 1) g
    k = e * (pow(tau, n - 1, R) - 1) % R * pow(tau - 1, -1, R) % R
    assert w_aff == pyref.g1_mul(pyref.G1_GEN, k)
    # absorb order of src/psnark/time_prover.rs:80-87: witness, ck (G2 powers), instance (the index), then alpha
    G1 = pyref.g1_serialize_uncompressed
    tr.append_message(b"witness", G1(w_aff))
    tr.append_message(b"ck", len(g2p).to_bytes(8, "little") + b"".join(pr.g2_serialize_uncompressed(p) for p in g2p))
    tr.append_message(b"instance", len(index).to_bytes(8, "little") + b"".join(G1(jac_to_affine_ints(oracle, c)) for c in index))
    alpha = tr.get_challenge(b"alpha")
    assert I(time_proof.zc_alpha) == (pow(alpha, n, R) - 1) * pow(alpha - 1, -1, R) % R
    assert len(time_proof.first_sumcheck_msgs[0]) == logn
    if logn >= 26:
        # ... and BASELINE configs[4] at its own size through the reference's acceptance predicate (src/psnark/tests.rs:144): the
        # example's stream key of 3 n + 1 powers holds the 2 n + 2 a verifiable proof needs (the --time-prover key of
        # examples/psnark.rs:76 is one short: test_reference_example_key_is_one_power_short); the preprocessing verifier is O(log n)
        from oracle import verifier_ref as V
        from tests.util import psnark_proof_to_ints

        vk = V.VerifierKey.from_trapdoor(tau, 5)
        stub = {"x": [e], "z": range(n)}  # the verifier reads the public input and the number of variables only
        V.psnark_verify(psnark_proof_to_ints(gm, oracle, elastic_proof), stub, vk, [jac_to_affine_ints(oracle, c) for c in index], n)
    stream.free()
    r1cs.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("kind,n", [("random", 8), ("random", 64), ("random", 1024), ("dummy", 4096)])
def test_device_proofs_are_accepted_by_the_reference_verifier(gm, oracle, pyref, kind, n):
    """src/psnark/tests.rs:130-145 (test_psnark_correctness: `time_proof.verify(&r1cs, &vk, &index, num_non_zero).is_ok()`):
    the device prover's proofs pass the reference's verification equations (oracle/verifier_ref.py: three sumcheck
    subclaims, the plookup / entry-product relations at beta and -beta, two pairing checks) at sizes the restated prover
    does not reach, for time and elastic provers; an altered proof is rejected."""
    from gemini_amd.circuit import R1csStream, dummy_r1cs
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr
    from oracle import snark_ref as sr
    from oracle import verifier_ref as V
    from tests.util import psnark_proof_to_ints

    if kind == "random":
        inst, tau = _random_instance(pyref, sr, n, 500 + n)
        r1cs = _device_instance(gm, oracle, inst, n)
    else:
        e, tau = 987654321987654321, 1234567890123456789012345
        inst = sr.dummy_r1cs(e, n)
        r1cs = dummy_r1cs(e, n)
    jm = pr.sum_matrices(inst["a"], inst["b"], inst["c"], n)
    nnz = len(pr.joint_matrices(jm, inst["a"], inst["b"], inst["c"])[0])
    ck = CommitterKey.new(nnz + 2 * n, 3, oracle.ints_to_limbs([tau], 4)[0])  # examples/psnark.rs:62, tests.rs:137
    vk = V.VerifierKey.from_trapdoor(tau, 3)
    assert ck.powers_of_g2 == vk.powers_of_g2
    index = Proof.index(ck, r1cs)
    index_ints = [jac_to_affine_ints(oracle, c) for c in index]
    proof = Proof.new_time(ck, r1cs, index)
    ints = psnark_proof_to_ints(gm, oracle, proof)
    V.psnark_verify(ints, inst, vk, index_ints, nnz)
    if n <= 64:
        stream = R1csStream(r1cs)
        ck_stream = CommitterKeyStream.from_committer_key(ck)
        elastic = Proof.new_elastic(ck_stream, stream, index, 1 << 5)
        V.psnark_verify(psnark_proof_to_ints(gm, oracle, elastic), inst, vk, index_ints, nnz)
        stream.free()
    bad = dict(ints)
    bad["rstars_vals"] = [ints["rstars_vals"][0], (ints["rstars_vals"][1] + 1) % pyref.R_MOD]
    with pytest.raises(V.VerificationError):
        V.psnark_verify(bad, inst, vk, index_ints, nnz)
    bad = dict(ints)
    bad["sorted_z_commitment"] = pyref.g1_add(ints["sorted_z_commitment"], ints["sorted_z_commitment"])
    with pytest.raises(V.VerificationError):
        V.psnark_verify(bad, inst, vk, index_ints, nnz)
    r1cs.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("logn", [20, 22, 24])
def test_full_size_device_proof_is_accepted_by_the_reference_verifier(gm, oracle, pyref, logn):
    """examples/psnark.rs:70-81 at 2^20 / 2^22 / 2^24 constraints (2^26 -- BASELINE configs[4] ITSELF -- goes through the same
    verifier in test_psnark_config5_shape_time_equals_elastic[26], on the key that test has built anyway).  The preprocessing
    verifier is O(log n) -- it never touches the matrices -- so the device proof of a full-size instance is checked
    against the reference's acceptance predicate directly (three sumcheck subclaims, plookup / entry-product relations,
    two pairing checks over ~25 commitments each)."""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof
    from oracle import verifier_ref as V
    from tests.util import psnark_proof_to_ints

    n = 1 << logn
    e, tau = 0x1D2C3B4A59687766554433221100FFEE % pyref.R_MOD, 0x0123456789ABCDEF0FEDCBA987654321 % pyref.R_MOD
    r1cs = dummy_r1cs(e, n)
    # examples/psnark.rs:76 asks for max_degree 2n (2n + 1 powers); the accumulated products of the sorted vectors have
    # 2n + 2 coefficients, the commitment would silently drop the top one (src/kzg/time.rs:82) and the proof would not
    # verify -- in the reference as here (oracle: test_reference_example_key_is_one_power_short).  One more power:
    ck = CommitterKey.new(2 * n + 1, 5, oracle.ints_to_limbs([tau], 4)[0])
    index = Proof.index(ck, r1cs)
    # 2^24 and 2^26 through the prover compiled into the library (gm_psnark_new_time); no gm_pool_trim() beforehand any more: the
    # prover asks for its footprint up front and the pool gives cached blocks back on demand
    proof = Proof.new_time(ck, r1cs, index, native=logn >= 24)
    vk = V.VerifierKey.from_trapdoor(tau, 5)
    stub = {"x": [e], "z": range(n)}  # the verifier reads the public input and the number of variables only
    V.psnark_verify(psnark_proof_to_ints(gm, oracle, proof), stub, vk, [jac_to_affine_ints(oracle, c) for c in index], n)
    r1cs.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("logn", [3, 6, 10, 16, 20])
def test_native_psnark_prover_equals_the_stepwise_one(gm, oracle, pyref, logn):
    """gm_psnark_new_time (src/psnark/time_prover.rs:69-384 compiled into the library, gemini_amd/csrc/psnark.cpp) against the
    step-by-step driver of gemini_amd/psnark.py on dummy_r1cs(2^logn) with the example's key recipe plus one power
    (examples/psnark.rs:70-81): the same proof, byte for byte, in both serialisation modes; same span names."""
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof

    n = 1 << logn
    e = oracle.limbs_to_ints(oracle.random_fr(7100 + logn, 1))[0]
    tau = oracle.limbs_to_ints(oracle.random_fr(7200 + logn, 1))[0]
    r1cs = dummy_r1cs(e, n)
    ck = CommitterKey.new(2 * n + 1, 5, oracle.ints_to_limbs([tau], 4)[0])
    index = Proof.index(ck, r1cs)
    stepwise = Proof.new_time(ck, r1cs, index, native=False)
    native = Proof.new_time(ck, r1cs, index, native=True)
    assert native == stepwise
    for compress in (True, False):
        assert native.serialize(compress, 0) == stepwise.serialize(compress, 0)
    assert set(native.spans) == set(stepwise.spans)
    assert Proof.new_time(ck, r1cs, index, native=True).serialize_compressed() == native.serialize_compressed()  # no state left behind
    assert all((x == y).all() for x, y in zip(Proof.index(ck, r1cs, native=True), index))  # gm_psnark_preprocess on A = B = C
    assert Proof.new_time(ck, r1cs, index, native="preprocess").serialize_compressed() == native.serialize_compressed()
    r1cs.free()
    ck.powers_of_g.free()


@pytest.mark.parametrize("n", [8, 64, 512])
def test_native_psnark_prover_on_general_instances(gm, oracle, pyref, n):
    """gm_psnark_new_time on GENERAL sparse instances (distinct A, B, a diagonal C; joint support of several entries per row,
    so the lookup vectors and the three entry products are not the degenerate ones of the dummy instance): byte-equal to the
    step-by-step driver (tests/soak_psnark.py runs the same over a thousand random instances)"""
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr
    from oracle import snark_ref as sr

    inst, tau = _random_instance(pyref, sr, n, 7300 + n)
    r1cs = _device_instance(gm, oracle, inst, n)
    jm = pr.sum_matrices(inst["a"], inst["b"], inst["c"], n)
    nnz = len(pr.joint_matrices(jm, inst["a"], inst["b"], inst["c"])[0])
    ck = CommitterKey.new(nnz + 2 * n, 3, oracle.ints_to_limbs([tau], 4)[0])
    index = Proof.index(ck, r1cs)
    want = Proof.new_time(ck, r1cs, index, native=False).serialize_compressed()
    assert Proof.new_time(ck, r1cs, index, native=True).serialize_compressed() == want
    # the matrix-only part of the instance built inside the library (gm_psnark_preprocess / gm_psnark_index) instead of by
    # gemini_amd/psnark.py::joint_matrices: the same joint support, value vectors, frequencies -> the same index and proof
    from gemini_amd.psnark import _joint_device, _joint_native

    jn, jd = _joint_native(r1cs), _joint_device(r1cs)
    assert jn.rec.nnz == nnz == len(jd.row_index)
    index_native = Proof.index(ck, r1cs, native=True)
    assert all((x == y).all() for x, y in zip(index_native, index))
    assert Proof.new_time(ck, r1cs, index_native, native="preprocess").serialize_compressed() == want
    r1cs.free()
    ck.powers_of_g.free()

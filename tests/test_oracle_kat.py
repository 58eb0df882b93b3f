"""The reference's own RNG-free known-answer tests, run against the oracle (CPU).  These are the
only fixed values the reference's test-suite holds for this path; they pin
fold order, coefficient endianness and the multi-point quotient.  Merlin / Keccak are pinned
against merlin's published vector and hashlib."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def M(orc, ints):
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4))


def I(orc, a):
    return orc.limbs_to_ints(orc.fr_from_mont(np.asarray(a).reshape(-1, 4)))


def test_linear_combination_kat(oracle, pyref):
    """src/misc.rs:402-422"""
    polys = [[100, 101, 102, 103], [100, 100, 100, 100]]
    assert pyref.linear_combination(polys, [1, 10]) == [1100, 1101, 1102, 1103]
    assert pyref.linear_combination([], []) == []
    got = oracle.linear_combination([M(oracle, p) for p in polys], M(oracle, [1, 10]))
    assert I(oracle, got) == [1100, 1101, 1102, 1103]


def test_foldings_polynomial_kat(oracle, pyref):
    """src/subprotocols/tensorcheck/mod.rs:388-398: fold [100..103] with 1 -> first entry 201"""
    assert pyref.fold_polynomial([100, 101, 102, 103], 1) == [201, 205]
    assert I(oracle, oracle.fold_polynomial(M(oracle, [100, 101, 102, 103]), M(oracle, [1])[0])) == [201, 205]


def test_folded_polynomial_tree_kat(oracle, pyref):
    """src/subprotocols/sumcheck/streams.rs:233-288: coefficients [1,2,1,1], challenges [1,2]:
    level-1 foldings (1+2, 1+1), level-2 folding 2 + 2*(1+2); twelve ones folded with four unit
    challenges sum to 12."""
    l1 = pyref.fold_polynomial([1, 2, 1, 1], 1)
    # the stream is big-endian (Reverse): emitted order is high to low, values are the same set
    assert l1 == [3, 2]
    l2 = pyref.fold_polynomial(l1, 2)
    assert l2 == [(3 + 2 * 2) % pyref.R_MOD]
    # NB the reference's stream is big-endian, i.e. the slice [1,2,1,1] is x^3 + 2x^2 + x + 1;
    # its level-1 items are (1 + 1*2 = 3 from the top pair) and (1 + 1*1 = 2), level 2 = 2 + 2*3
    be = list(reversed([1, 2, 1, 1]))  # little-endian view of the same polynomial
    l1 = pyref.fold_polynomial(be, 1)
    assert l1 == [2, 3]
    assert pyref.fold_polynomial(l1, 2) == [2 + 2 * 3]
    cur = [1] * 12
    for _ in range(4):
        cur = pyref.fold_polynomial(cur, 1)
    assert cur == [12]
    cur = M(oracle, [1] * 12)
    for _ in range(4):
        cur = oracle.fold_polynomial(cur, M(oracle, [1])[0])
    assert I(oracle, cur) == [12]


def test_vanishing_polynomial_kat(pyref):
    """src/kzg/mod.rs:271-281"""
    z = pyref.vanishing_polynomial([10, 5, 13])
    for p in (10, 5, 13):
        assert pyref.evaluate_le(z, p) == 0


def test_open_multi_points_kat(oracle, pyref):
    """src/kzg/space.rs:334-355: f = 80x^6+80x^5+88x^4+3x^3+73x^2+7x+24 (given big-endian),
    points (beta^2, beta, -beta), beta = 53: evaluate_be(remainder, beta) == 1807299544171 == f(53)"""
    be = [80, 80, 88, 3, 73, 7, 24]
    le = list(reversed(be))
    beta = 53
    pts = [beta * beta % pyref.R_MOD, beta, (-beta) % pyref.R_MOD]
    z = pyref.vanishing_polynomial(pts)
    q, rem = pyref.poly_divmod(le, z)
    assert pyref.evaluate_le(rem, beta) == 1807299544171 == pyref.evaluate_be(be, beta)
    qo, remo = oracle.poly_div_monic(M(oracle, le), M(oracle, z))
    assert I(oracle, qo) == q and I(oracle, remo) == rem
    # single point: remainder has one element (space.rs:344-346)
    q1, r1 = pyref.poly_divmod(le, pyref.vanishing_polynomial([beta]))
    assert len(r1) == 1 and r1[0] == 1807299544171


def test_rounds_kat(oracle, pyref):
    """src/subprotocols/sumcheck/time_prover.rs:141-159: degree-1 needs 1 round, 17 coefficients 5"""
    assert pyref.TimeProver([1, 2], [3, 4], 1).tot_rounds == 1
    assert pyref.TimeProver(list(range(17)), [1, 2], 1).tot_rounds == 5
    assert oracle.TimeProver(M(oracle, list(range(17))), M(oracle, [1, 2]), M(oracle, [1])[0]).tot_rounds == 5


def test_keccak_and_merlin_vectors(pyref):
    for msg in [b"", b"abc", b"a" * 135, b"a" * 136, b"a" * 137, bytes(range(256)) * 3]:
        assert pyref.sha3_256(msg) == hashlib.sha3_256(msg).digest()
    # merlin 3.0.0 src/transcript.rs `equivalence_simple` published challenge
    t = pyref.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_golden_fixtures(oracle, pyref):
    """tests/golden/*.json were produced by tests/golden/make_golden.py (pyref, big-int); the C
    oracle must reproduce every one of them."""
    with open(os.path.join(GOLDEN, "msm_small.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        bases = np.stack([oracle.ints_to_affine(None if p is None else (int(p[0], 16), int(p[1], 16))) for p in case["bases"]]) if case["bases"] else np.empty((0, 12), dtype=np.uint64)
        sc = oracle.ints_to_limbs([int(s, 16) for s in case["scalars"]], 4)
        got = oracle.affine_to_ints(oracle.g1_to_affine(oracle.msm_pippenger(bases, sc)))
        exp = None if case["result"] is None else (int(case["result"][0], 16), int(case["result"][1], 16))
        assert got == exp, case["name"]
    with open(os.path.join(GOLDEN, "sumcheck_small.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        f = M(oracle, [int(x, 16) for x in case["f"]])
        g = M(oracle, [int(x, 16) for x in case["g"]])
        tw = M(oracle, [int(case["twist"], 16)])[0]
        P = oracle.TimeProver(f, g, tw)
        vm = None
        for rnd, (msg, ch) in enumerate(zip(case["messages"], case["challenges"])):
            a, b = P.next_message(vm)
            assert I(oracle, np.stack([a, b])) == [int(msg[0], 16), int(msg[1], 16)], (case["name"], rnd)
            vm = M(oracle, [int(ch, 16)])[0]
        assert P.next_message(vm) is None
        assert I(oracle, np.stack(P.final_foldings())) == [int(x, 16) for x in case["final_foldings"]]

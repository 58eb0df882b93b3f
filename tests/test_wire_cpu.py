"""Wire formats (gemini_amd/wire.py) on the CPU: G1 / Fr codecs against published constants and the oracle-side
serializer (oracle/wire_ref.py), whole-proof byte equality on synthetic proofs, and serialise -> deserialise
round trips in every mode.  The derive layouts are those of src/snark/mod.rs:75-82, src/psnark/mod.rs:29-51,
tensorcheck/mod.rs:110-121, sumcheck/prover.rs:9-14, kzg/mod.rs:107-112."""
import numpy as np
import pytest

from gemini_amd import wire
from gemini_amd.fr import fr_from_int
from oracle import pyref as P
from oracle import wire_ref as W

MODES = [(True, 0, "arkworks"), (False, 0, "arkworks"), (True, 1, "zcash"), (False, 1, "zcash")]


def _pts(n, seed=7):
    rng = P.SplitMix64(seed)
    return [P.g1_mul(P.G1_GEN, rng.fr()) for _ in range(n)]


def test_zcash_generator_known_answer():
    """the compressed / uncompressed encodings of the BLS12-381 G1 generator published with the zcash format
    (IETF draft-irtf-cfrg-pairing-friendly-curves, appendix C): pins byte order and flag positions of mode 1"""
    g = wire.g1_from_affine_ints(P.G1_GEN)
    comp = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    unc = bytes.fromhex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
    assert wire.g1_serialize(g, True, wire.G1Encoding.ZCASH) == comp
    assert wire.g1_serialize(g, False, wire.G1Encoding.ZCASH) == unc
    inf = wire.g1_from_affine_ints(None)
    assert wire.g1_serialize(inf, True, wire.G1Encoding.ZCASH) == bytes([0xC0]) + bytes(47)
    assert wire.g1_serialize(inf, False, wire.G1Encoding.ZCASH) == bytes([0x40]) + bytes(95)
    # -G: the sort flag (bit 5) appears
    assert wire.g1_serialize(wire.g1_from_affine_ints(P.g1_neg(P.G1_GEN)), True, wire.G1Encoding.ZCASH)[0] == 0xB7


@pytest.mark.parametrize("compress,enc,mode", MODES)
def test_g1_codec_round_trip_and_oracle_bytes(compress, enc, mode):
    for p in _pts(6) + [None, P.G1_GEN, P.g1_neg(P.G1_GEN)]:
        jac = wire.g1_from_affine_ints(p)
        b = wire.g1_serialize(jac, compress, enc)
        assert b == W.g1(p, compress, mode) and len(b) == (48 if compress else 96)
        back, pos = wire.g1_deserialize(b, 0, compress, enc, validate=True)
        assert pos == len(b) and np.array_equal(back, jac)


def test_arkworks_uncompressed_is_what_the_transcript_absorbs():
    """append_serializable uses serialize_uncompressed (src/transcript.rs:16-24): the wire module, the oracle's
    transcript framing and the library's gm_transcript_append_g1 must agree on those 96 bytes"""
    from gemini_amd.transcript import Transcript

    for p in _pts(3, seed=11) + [None]:
        jac = wire.g1_from_affine_ints(p)
        assert wire.g1_serialize(jac, False) == P.g1_serialize_uncompressed(p)
        for enc in (0, 1):
            a, b = Transcript(), Transcript()
            a.set_g1_encoding(enc)
            a.append_g1(b"commitment", jac)
            b.append_message(b"commitment", wire.g1_serialize(jac, False, enc))
            assert a.challenge_bytes(b"c", 32) == b.challenge_bytes(b"c", 32)
            a.free()
            b.free()


def test_deserialize_rejects_malformed_points():
    g = wire.g1_serialize(wire.g1_from_affine_ints(P.G1_GEN), True)
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(g[:-1], 0, True)  # short
    bad = bytearray(g)
    bad[47] |= 0xC0  # both flags
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(bytes(bad), 0, True)
    # x = 1: 1 + 4 = 5 is a quadratic non-residue mod q? find an x off the curve
    x = 1
    while pow((x**3 + 4) % P.Q_MOD, (P.Q_MOD - 1) // 2, P.Q_MOD) == 1:
        x += 1
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(x.to_bytes(48, "little"), 0, True)
    # on the curve but outside the prime-order subgroup (cofactor != 1): rejected only with validation
    x = 2
    while True:
        y = wire._sqrt_q((x**3 + 4) % P.Q_MOD)
        if y is not None and not wire._in_prime_order_subgroup((x, y)):
            break
        x += 1
    unc = x.to_bytes(48, "little") + y.to_bytes(48, "little")
    unc = unc[:-1] + bytes([unc[-1] | (0x80 if y > P.Q_MOD - y else 0)])
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(unc, 0, False, validate=True)
    wire.g1_deserialize(unc, 0, False, validate=False)
    with pytest.raises(wire.WireError):
        wire.fr_deserialize(P.R_MOD.to_bytes(32, "little"), 0)
    # non-canonical encodings ark-ec 0.4 accepts (recalled): rejected by default (one byte string per proof), accepted with strict = False
    gu = bytearray(wire.g1_serialize(wire.g1_from_affine_ints(P.G1_GEN), False))
    gu[-1] ^= 0x80  # the y-sign flag of an uncompressed point
    pt, _ = wire.g1_deserialize(bytes(gu), 0, False, strict=False)
    assert np.array_equal(pt, wire.g1_from_affine_ints(P.G1_GEN))
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(bytes(gu), 0, False)
    inf = bytearray(wire.g1_serialize(wire.g1_from_affine_ints(None), True))
    inf[0] = 7  # infinity flag over a non-zero x
    pt, _ = wire.g1_deserialize(bytes(inf), 0, True, strict=False)
    assert np.array_equal(pt, wire.g1_from_affine_ints(None))
    with pytest.raises(wire.WireError):
        wire.g1_deserialize(bytes(inf), 0, True)


def _synthetic_snark(rounds=5, seed=3):
    rng = P.SplitMix64(seed)
    pts = _pts(rounds + 2, seed)
    F = lambda: rng.fr()
    msgs = lambda: ([(F(), F()) for _ in range(rounds)], [(F(), F())])
    return {
        "witness_commitment": pts[0], "zc_alpha": F(), "first_sumcheck_msgs": msgs(), "second_sumcheck_msgs": msgs(),
        "tensorcheck_proof": {
            "folded_polynomials_commitments": pts[1:rounds] + [None],
            "folded_polynomials_evaluations": [[F(), F()] for _ in range(rounds)],
            "evaluation_proof": pts[rounds + 1],
            "base_polynomials_evaluations": [[F(), F(), F()]],
        },
    }


def _to_device_types(d):
    """oracle integers -> the array images the provers return"""
    from gemini_amd.snark import Proof
    from gemini_amd.tensorcheck import TensorcheckProof

    G, S = wire.g1_from_affine_ints, fr_from_int
    pm = lambda m: ([(S(a), S(b)) for a, b in m[0]], [(S(a), S(b)) for a, b in m[1]])
    tc = d["tensorcheck_proof"]
    t = TensorcheckProof([G(c) for c in tc["folded_polynomials_commitments"]],
                         [np.stack([S(e) for e in e2]) for e2 in tc["folded_polynomials_evaluations"]],
                         G(tc["evaluation_proof"]), [np.stack([S(e) for e in e3]) for e3 in tc["base_polynomials_evaluations"]])
    return Proof(G(d["witness_commitment"]), S(d["zc_alpha"]), pm(d["first_sumcheck_msgs"]), pm(d["second_sumcheck_msgs"]), t)


@pytest.mark.parametrize("compress,enc,mode", MODES)
def test_snark_proof_bytes_equal_the_oracle_serializer_and_round_trip(compress, enc, mode):
    d = _synthetic_snark()
    proof = _to_device_types(d)
    data = proof.serialize(compress, enc)
    assert data == W.snark_proof(d, compress, mode)
    back = type(proof).deserialize(data, compress, enc)
    assert back == proof and back.serialize(compress, enc) == data
    # size formula of examples/snark.rs:96 (compressed): 2 G1 + zc + 2 ProverMsgs + tensorcheck
    if compress:
        r = 5
        assert len(data) == 48 + 32 + 2 * (8 + 64 * r + 8 + 64) + (8 + 48 * r) + (8 + 64 * r) + 48 + (8 + 96)
    with pytest.raises(wire.WireError):
        type(proof).deserialize(data + b"\0", compress, enc)
    other = _to_device_types(_synthetic_snark(seed=4))
    assert other != proof


@pytest.mark.parametrize("compress,enc,mode", MODES)
def test_psnark_proof_bytes_equal_the_oracle_serializer_and_round_trip(compress, enc, mode):
    from gemini_amd.psnark import EntryProductMsgs, Proof

    rng = P.SplitMix64(21)
    pts = _pts(16, 21)
    F = lambda: rng.fr()
    msgs = lambda r: ([(F(), F()) for _ in range(r)], [(F(), F()) for _ in range(2)])
    base = _synthetic_snark(4, 22)
    d = {
        "witness_commitment": pts[0], "zc_alpha": F(), "first_sumcheck_msgs": msgs(3), "r_star_commitments": pts[1:4],
        "z_star_commitment": pts[4], "second_sumcheck_msgs": msgs(4),
        "set_r_ep": F(), "subset_r_ep": F(), "sorted_r_commitment": pts[5], "set_alpha_ep": F(), "subset_alpha_ep": F(),
        "sorted_alpha_commitment": pts[6], "set_z_ep": F(), "subset_z_ep": F(), "sorted_z_commitment": None,
        "ep_msgs": {"acc_v_commitments": pts[7:12], "claimed_sumchecks": [F() for _ in range(5)]},
        "ralpha_star_acc_mu_evals": [F() for _ in range(7)], "ralpha_star_acc_mu_proof": pts[12], "rstars_vals": [F(), F()],
        "third_sumcheck_msgs": msgs(5), "tensorcheck_proof": base["tensorcheck_proof"],
    }
    G, S = wire.g1_from_affine_ints, fr_from_int
    pm = lambda m: ([(S(a), S(b)) for a, b in m[0]], [(S(a), S(b)) for a, b in m[1]])
    kw = {}
    for k, v in d.items():
        if k.endswith("_msgs") and k != "ep_msgs":
            kw[k] = pm(v)
        elif k == "ep_msgs":
            kw[k] = EntryProductMsgs([G(c) for c in v["acc_v_commitments"]], [S(e) for e in v["claimed_sumchecks"]])
        elif k == "tensorcheck_proof":
            kw[k] = _to_device_types(base).tensorcheck_proof
        elif k in ("r_star_commitments",):
            kw[k] = [G(c) for c in v]
        elif k in ("ralpha_star_acc_mu_evals", "rstars_vals"):
            kw[k] = [S(e) for e in v]
        elif isinstance(v, int):
            kw[k] = S(v)
        else:
            kw[k] = G(v)
    proof = Proof(**kw)
    data = proof.serialize(compress, enc)
    assert data == W.psnark_proof(d, compress, mode)
    back = Proof.deserialize(data, compress, enc)
    assert back == proof and back.serialize(compress, enc) == data

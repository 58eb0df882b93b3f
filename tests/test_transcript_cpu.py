"""CPU: the product's C++ Merlin / GeminiTranscript (gemini_amd/csrc/transcript.cpp, host code
inside libgemini_hip.so) against merlin's published vector and the Python restatement."""
import numpy as np


def test_merlin_published_vector():
    import gemini_amd as gm

    t = gm.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    t.free()


def test_gemini_transcript_vs_pyref(oracle, pyref):
    import gemini_amd as gm
    from gemini_amd.fr import fr_from_int, fr_to_int

    a = gm.Transcript()
    b = pyref.GeminiTranscript(pyref.PROTOCOL_NAME)
    rng = pyref.SplitMix64(3)
    for rnd in range(40):
        x, y = rng.fr(), rng.fr()
        a.append_round_msg(b"evaluations", fr_from_int(x), fr_from_int(y))
        b.append_round_msg(b"evaluations", x, y)
        assert fr_to_int(a.get_challenge(b"challenge")) == b.get_challenge(b"challenge")
    # long messages cross the STROBE rate (166 bytes) several times
    blob = bytes(range(256)) * 5
    a.append_message(b"blob", blob)
    b.append_message(b"blob", blob)
    assert a.challenge_bytes(b"c", 400) == b.challenge_bytes(b"c", 400)
    # G1 framing, incl. the identity and both y-sign flags
    one = oracle.fq_to_mont(oracle.ints_to_limbs([1], 6))[0]
    for k in [1, 2, 3, 12345, pyref.R_MOD - 1]:
        P = pyref.g1_mul(pyref.G1_GEN, k)
        jac = np.concatenate([oracle.ints_to_affine(P), one])
        a.append_g1(b"commitment", jac)
        b.append_message(b"commitment", pyref.g1_serialize_uncompressed(P))
        assert fr_to_int(a.get_challenge(b"x")) == b.get_challenge(b"x")
    ident = np.concatenate([one, one, np.zeros(6, dtype=np.uint64)])
    a.append_g1(b"commitment", ident)
    b.append_message(b"commitment", pyref.g1_serialize_uncompressed(None))
    assert fr_to_int(a.get_challenge(b"x")) == b.get_challenge(b"x")
    # non-normalised Jacobian input is normalised before framing
    P = pyref.g1_mul(pyref.G1_GEN, 99)
    jac = oracle.g1_mul(oracle.g1_generator(), oracle.ints_to_limbs([99], 4)[0])
    a.append_g1(b"commitment", jac)
    b.append_message(b"commitment", pyref.g1_serialize_uncompressed(P))
    assert fr_to_int(a.get_challenge(b"x")) == b.get_challenge(b"x")
    a.free()

#!/usr/bin/env python3
"""Writes MOCK ref_*.json files -- produced by THIS repository's CPU restatement, NOT by the reference -- into a
scratch directory, so that the plumbing of tests/test_ref_vectors.py can be exercised where no Rust toolchain
exists:  GM_REFVECTORS_DIR=$(python tools/refvectors/mock_vectors.py /tmp/mockvec 4) pytest tests/test_ref_vectors.py
A pass with these files proves nothing about parity with arkworks; only files written by `cargo run` do."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402
from oracle import snark_ref as sr  # noqa: E402
from oracle import wire_ref as W  # noqa: E402


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mockvec"
    max_logn = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    os.makedirs(out_dir, exist_ok=True)
    hx = lambda v: hex(v)
    for crate, fname, mode in (("ark-test-curves", "ref_ark_test_curves.json", "arkworks"), ("ark-bls12-381", "ref_ark_bls12_381.json", "zcash")):
        saved = P.g1_serialize_uncompressed
        if mode == "zcash":
            P.g1_serialize_uncompressed = lambda p: W.g1(p, False, "zcash")
        rng = P.SplitMix64(0x4D4F434B)
        cases = []
        for logn in range(3, max_logn + 1):
            n = 1 << logn
            e, tau = rng.fr(), rng.fr()
            g = P.g1_mul(P.G1_GEN, rng.fr())
            inst = sr.dummy_r1cs(e, n)
            exp = sr.snark_new_time(inst, sr.srs(tau, 2 * n + 1, g))
            t = P.GeminiTranscript(P.PROTOCOL_NAME)
            t.append_message(b"witness", W.g1(exp["witness_commitment"], False, mode))
            alpha = t.get_challenge(b"alpha")
            raw = P.GeminiTranscript(P.PROTOCOL_NAME).challenge_bytes(b"raw", 64)
            raw_fr = P.GeminiTranscript(P.PROTOCOL_NAME).get_challenge(b"raw")
            from gemini_amd import g2 as G2

            el = sr.snark_new_time(inst, sr.srs(1, n + 1))  # generator copies = tau 1 (time == elastic proof, src/snark/tests.rs)
            cases.append({"logn": logn, "e": hx(e), "tau": hx(tau), "g": [hx(g[0]), hx(g[1])],
                          "g2_uncompressed": G2.serialize_uncompressed(G2.generator(), 1 if mode == "zcash" else 0).hex(),
                          "witness_commitment_uncompressed": W.g1(exp["witness_commitment"], False, mode).hex(),
                          "witness_commitment_compressed": W.g1(exp["witness_commitment"], True, mode).hex(),
                          "alpha_after_witness": hx(alpha), "raw_challenge_bytes": raw.hex(), "raw_challenge_as_fr": hx(raw_fr),
                          "proof_compressed": W.snark_proof(exp, True, mode).hex(), "proof_uncompressed": W.snark_proof(exp, False, mode).hex(), "verifies": True,
                          "elastic_generator_key": {"proof_compressed": W.snark_proof(el, True, mode).hex(), "proof_uncompressed": W.snark_proof(el, False, mode).hex()},
                          "psnark": None})
        P.g1_serialize_uncompressed = saved
        body = {"generator": "MOCK -- oracle/ of this repository, NOT the reference", "curve_crate": crate,
                "g1_generator_uncompressed": W.g1(P.G1_GEN, False, mode).hex(), "g1_generator_compressed": W.g1(P.G1_GEN, True, mode).hex(), "cases": cases}
        with open(os.path.join(out_dir, fname), "w") as fh:
            json.dump(body, fh)
    print(out_dir)


if __name__ == "__main__":
    main()

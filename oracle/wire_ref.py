"""TEST INFRASTRUCTURE (oracle side): ark-serialize byte images of the reference's proof types, written
independently of gemini_amd/wire.py from the integer-level proofs of oracle/snark_ref.py / psnark_ref.py.
Only tests/ import this.

Follows the derive(CanonicalSerialize) layouts of the reference -- src/kzg/mod.rs:107-112 (Commitment,
EvaluationProof: one G1), src/subprotocols/sumcheck/prover.rs:9-14 (RoundMsg(F, F), ProverMsgs(Vec<RoundMsg>,
Vec<[F; 2]>)), src/subprotocols/tensorcheck/mod.rs:110-121, src/subprotocols/entryproduct/mod.rs:48-52,
src/snark/mod.rs:75-82, src/psnark/mod.rs:29-51 -- and the framing rules of ark-serialize 0.4 (third-party crate,
Cargo.lock:152-154; restated, unverifiable in this image): Vec = u64 LE length + items, arrays and structs = their
items, Fp = canonical value little-endian.  G1, Compress::Yes / No:
  mode "arkworks" (ark-ec default, ark-test-curves): x LE [|| y LE], SWFlags in the top two bits of the last byte;
  mode "zcash" (ark-bls12-381): x BE [|| y BE], flags in the top three bits of the first byte.
Points are affine integer pairs or None (identity); scalars are integers.
"""
from . import pyref as P

Q = P.Q_MOD


def fr(v: int) -> bytes:
    return (v % P.R_MOD).to_bytes(32, "little")


def u64(n: int) -> bytes:
    return n.to_bytes(8, "little")


def g1(p, compress: bool, mode: str = "arkworks") -> bytes:
    larger = p is not None and p[1] > (Q - p[1]) % Q
    if mode == "arkworks":
        if p is None:
            body = bytes(48 if compress else 96)
        else:
            body = p[0].to_bytes(48, "little") + (b"" if compress else p[1].to_bytes(48, "little"))
        last = body[-1] | (0x40 if p is None else 0) | (0x80 if larger else 0)
        return body[:-1] + bytes([last])
    assert mode == "zcash"
    if p is None:
        body = bytes(48 if compress else 96)
    else:
        body = p[0].to_bytes(48, "big") + (b"" if compress else p[1].to_bytes(48, "big"))
    first = body[0] | (0x80 if compress else 0) | (0x40 if p is None else 0) | (0x20 if (compress and larger) else 0)
    return bytes([first]) + body[1:]


def prover_msgs(m) -> bytes:
    msgs, finals = m
    if len(finals) == 2 and isinstance(finals[0], int):  # a single prover's (f0, g0): Vec<[F; 2]> of length one
        finals = [finals]
    out = u64(len(msgs))
    for a, b in msgs:
        out += fr(a) + fr(b)
    out += u64(len(finals))
    for f0, g0 in finals:
        out += fr(f0) + fr(g0)
    return out


def tensorcheck(tc, compress, mode) -> bytes:
    out = u64(len(tc["folded_polynomials_commitments"]))
    for c in tc["folded_polynomials_commitments"]:
        out += g1(c, compress, mode)
    out += u64(len(tc["folded_polynomials_evaluations"]))
    for e2 in tc["folded_polynomials_evaluations"]:
        assert len(e2) == 2
        out += fr(e2[0]) + fr(e2[1])
    out += g1(tc["evaluation_proof"], compress, mode)
    out += u64(len(tc["base_polynomials_evaluations"]))
    for e3 in tc["base_polynomials_evaluations"]:
        assert len(e3) == 3
        out += fr(e3[0]) + fr(e3[1]) + fr(e3[2])
    return out


def snark_proof(p, compress: bool, mode: str = "arkworks") -> bytes:
    """src/snark/mod.rs:75-82"""
    return (g1(p["witness_commitment"], compress, mode) + fr(p["zc_alpha"]) + prover_msgs(p["first_sumcheck_msgs"])
            + prover_msgs(p["second_sumcheck_msgs"]) + tensorcheck(p["tensorcheck_proof"], compress, mode))


def psnark_proof(p, compress: bool, mode: str = "arkworks") -> bytes:
    """src/psnark/mod.rs:29-51"""
    G = lambda c: g1(c, compress, mode)
    out = G(p["witness_commitment"]) + fr(p["zc_alpha"]) + prover_msgs(p["first_sumcheck_msgs"])
    assert len(p["r_star_commitments"]) == 3
    for c in p["r_star_commitments"]:
        out += G(c)
    out += G(p["z_star_commitment"]) + prover_msgs(p["second_sumcheck_msgs"])
    for k in ("r", "alpha", "z"):
        out += fr(p[f"set_{k}_ep"]) + fr(p[f"subset_{k}_ep"]) + G(p[f"sorted_{k}_commitment"])
    ep = p["ep_msgs"]
    out += u64(len(ep["acc_v_commitments"]))
    for c in ep["acc_v_commitments"]:
        out += G(c)
    out += u64(len(ep["claimed_sumchecks"]))
    for e in ep["claimed_sumchecks"]:
        out += fr(e)
    out += u64(len(p["ralpha_star_acc_mu_evals"]))
    for e in p["ralpha_star_acc_mu_evals"]:
        out += fr(e)
    out += G(p["ralpha_star_acc_mu_proof"])
    assert len(p["rstars_vals"]) == 2
    out += fr(p["rstars_vals"][0]) + fr(p["rstars_vals"][1])
    out += prover_msgs(p["third_sumcheck_msgs"]) + tensorcheck(p["tensorcheck_proof"], compress, mode)
    return out

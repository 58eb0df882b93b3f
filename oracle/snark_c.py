"""TEST / BENCH INFRASTRUCTURE: `snark::Proof::new_time` (src/snark/time_prover.rs:19-117) on the dummy instance,
run end to end on the CPU through the C restatement (oracle/gemini_oracle.c) -- the same statement as
oracle/snark_ref.py, with every O(n) pass a C call instead of a Python-integer loop, so that it can serve as the
CPU baseline of the `time_prover` metric (bench.py, kind "port") at sizes where Python integers take minutes.

CPU execution model = the reference's: the MSMs run one OpenMP task per window (ark-ec's parallel grain,
src/kzg/msm/variable_base.rs:125-167), every field pass and both sumchecks are single-threaded loops
(src/subprotocols/sumcheck/time_prover.rs:83-123, src/misc.rs).  Only tests/ and bench.py's cpu_baseline leg import
this module."""
from __future__ import annotations

import time

import numpy as np

from . import oracle as orc
from . import pyref as P

R = P.R_MOD


def _m(v: int) -> np.ndarray:
    return orc.fr_to_mont(orc.ints_to_limbs([v % R], 4))[0]


def _i(x) -> int:
    return orc.limbs_to_ints(orc.fr_from_mont(np.asarray(x, dtype=np.uint64).reshape(1, 4)))[0]


def _commit(powers_of_g, poly_mont) -> tuple | None:
    n = min(len(powers_of_g), len(poly_mont))
    if n == 0:
        return None
    return orc.affine_to_ints(orc.g1_to_affine(orc.msm_pippenger(powers_of_g[:n], orc.fr_from_mont(poly_mont[:n]))))


def _sumcheck(tr, f, g, twist):
    """src/subprotocols/sumcheck/proof.rs:36-66"""
    pr = orc.TimeProver(f, g, twist)
    msgs, chs = [], []
    vm = None
    while True:
        m = pr.next_message(vm)
        if m is None:
            break
        a, b = _i(m[0]), _i(m[1])
        tr.append_round_msg(b"evaluations", a, b)
        ch = tr.get_challenge(b"challenge")
        msgs.append((a, b))
        chs.append(ch)
        vm = _m(ch)
    f0, g0 = pr.final_foldings()
    ff = (_i(f0), _i(g0))
    tr.append_fr(b"final-folding", ff[0])
    tr.append_fr(b"final-folding", ff[1])
    return msgs, chs, ff


def new_time_dummy(e: int, n: int, powers_of_g: np.ndarray) -> dict:
    """dummy_r1cs(e, n) (src/circuit.rs:349-365): z = [e; n], w = [e; n - 1], A = B = C = diag(1 / e).
    powers_of_g: (>= n, 12) affine Montgomery limbs.  Returns the proof as oracle/snark_ref.py does, plus spans."""
    spans = {}
    t_all = time.perf_counter()
    em, inv_e = _m(e), _m(pow(e, -1, R))
    z = np.tile(em, (n, 1))
    w = z[: n - 1]
    diag = np.tile(inv_e, (n, 1))
    z_a = orc.hadamard(diag, z)  # product_matrix_vector with a diagonal matrix, three times as in :32-34
    z_b = orc.hadamard(diag, z)
    z_c = orc.hadamard(diag, z)
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    t0 = time.perf_counter()
    witness_commitment = _commit(powers_of_g, w)
    spans["Commitment to w"] = time.perf_counter() - t0
    tr.append_message(b"witness", P.g1_serialize_uncompressed(witness_commitment))
    alpha = tr.get_challenge(b"alpha")
    zc_alpha = _i(orc.evaluate_le(z_c, _m(alpha)))
    tr.append_fr(b"zc(alpha)", zc_alpha)
    t0 = time.perf_counter()
    m1, ch1, ff1 = _sumcheck(tr, z_a, z_b, _m(alpha))
    spans["First sumcheck"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    b_ch = orc.tensor(np.stack([_m(c) for c in ch1]))
    c_ch = orc.powers(_m(alpha), len(b_ch))
    a_ch = orc.hadamard(b_ch, c_ch)
    eta = tr.get_challenge(b"eta")
    # abc_tensored[col] = sum_rows ... with diagonal matrices: inv_e * (a_ch + eta b_ch + eta^2 c_ch)   (:63-81)
    comb = orc.linear_combination([a_ch, b_ch, c_ch], np.stack([_m(1), _m(eta), _m(eta * eta % R)]))
    abc = np.zeros((n, 4), dtype=np.uint64)
    abc[: len(comb)] = orc.hadamard(diag[: len(comb)], comb)
    spans["tensor/powers/hadamard/abc_tensored"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    m2, ch2, ff2 = _sumcheck(tr, abc, z, _m(1))
    spans["Second sumcheck"] = time.perf_counter() - t0
    # ---- tensorcheck (src/subprotocols/tensorcheck/mod.rs:190-275): base [w], body [([abc, z], ch2)] ------------
    t0 = time.perf_counter()
    batch_challenge = tr.get_challenge(b"batch_challenge")
    bcs = orc.powers(_m(batch_challenge), n)
    batched = orc.linear_combination([abc, z], bcs)
    foldings = []
    cur = batched
    for ch in ch2[:-1]:
        cur = orc.fold_polynomial(cur, _m(ch))
        foldings.append(cur)
    commitments = [_commit(powers_of_g, f) for f in foldings]
    for c in commitments:
        tr.append_message(b"commitment", P.g1_serialize_uncompressed(c))
    eval_chal = tr.get_challenge(b"evaluation-chal")
    pts = [eval_chal * eval_chal % R, eval_chal, (-eval_chal) % R]
    base_evals = [[_i(orc.evaluate_le(w, _m(p))) for p in pts]]
    fold_evals = [[_i(orc.evaluate_le(f, _m(p))) for p in pts[1:]] for f in foldings]
    for e3 in base_evals:
        for v in e3:
            tr.append_fr(b"eval", v)
    for e2 in fold_evals:
        for v in e2:
            tr.append_fr(b"eval", v)
    open_chal = tr.get_challenge(b"open-chal")
    polys = [w] + foldings
    lin = orc.linear_combination(polys, orc.powers(_m(open_chal), len(polys)))
    van = np.stack([_m(c) for c in P.vanishing_polynomial(pts)])
    q, _ = orc.poly_div_monic(lin, van)
    evaluation_proof = _commit(powers_of_g, q)
    spans["Tensorcheck"] = time.perf_counter() - t0
    spans["ark_gemini::snark::time_prover"] = time.perf_counter() - t_all
    return {"witness_commitment": witness_commitment, "zc_alpha": zc_alpha, "first_sumcheck_msgs": (m1, ff1),
            "second_sumcheck_msgs": (m2, ff2),
            "tensorcheck_proof": {"folded_polynomials_commitments": commitments, "folded_polynomials_evaluations": fold_evals,
                                  "evaluation_proof": evaluation_proof, "base_polynomials_evaluations": base_evals},
            "spans": spans}

"""Time-bounded self-consistency soak of the non-preprocessing SNARK provers on RANDOM sparse R1CS instances.  NOT collected by
default (the file name); on the GPU box:

    SOAK_SECONDS=300 python -m pytest tests/soak_provers.py -q -s        # writes gpurun_out/soak_provers.json

Per case: a satisfied random instance (1-3 entries per row of A and B in distinct columns, a diagonal C, 1-3 public inputs) of
2^2 .. 2^12 constraints and a fresh key; then `Proof::new_time` step by step from Python, `gm_snark_new_time` (compiled driver),
`gm_snark_new_elastic` over the stream form of the same instance, and below 2^10 the step-by-step elastic driver -- four
independent orchestrations over the same kernels -- must serialise to the same bytes (`assert_eq!(time_proof, space_proof)`,
src/snark/tests.rs:14-57), and every fourth proof must be ACCEPTED by the restated reference verifier (sumcheck subclaims,
tensor relation, pairing check: oracle/verifier_ref.py, src/snark/verifier.rs)."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _M(orc, ints):
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4))


def test_soak_provers(gm, oracle, pyref):
    from gemini_amd.circuit import R1cs, R1csStream, SparseMatrix
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.snark import Proof
    from oracle import snark_ref as sr
    from oracle import verifier_ref as V
    from tests.util import random_r1cs_instance, snark_proof_to_ints

    budget = float(os.environ.get("SOAK_SECONDS", "30"))
    seed0 = int(os.environ.get("SOAK_SEED", "20241001"))
    t_end = time.time() + budget
    stats = {"cases": 0, "verified": 0, "by_logn": {}, "failures": []}
    case = 0
    M = lambda v: gm.fr.fr_from_int(v)  # noqa: E731
    dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]  # noqa: E731
    while time.time() < t_end:
        rng = np.random.default_rng(seed0 + case)
        logn = int(rng.integers(2, 13))
        n = 1 << logn
        nx = int(rng.integers(1, 4))
        inst, _ = random_r1cs_instance(pyref, sr, n, seed0 + 3 * case + 1, nx=nx)
        tau = oracle.limbs_to_ints(oracle.random_fr(seed0 + 3 * case + 2, 1))[0]
        ck = CommitterKey.new(2 * n, 5, oracle.ints_to_limbs([tau], 4)[0])
        mats = [SparseMatrix.from_rows(dev(inst[k]), n) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), n, transpose=True) for k in "abc"]
        g = R1cs(*mats, gm.FrVec.from_host(_M(oracle, inst["z"])), gm.FrVec.from_host(_M(oracle, inst["w"])), gm.FrVec.from_host(_M(oracle, inst["x"])))
        gs = R1csStream(g)
        merged = CommitterKeyStream.from_committer_key(ck)
        bad = []
        try:
            stepwise = Proof.new_time(g, ck, native=False)
            want = stepwise.serialize_compressed()
            if Proof.new_time(g, ck, native=True).serialize_compressed() != want:
                bad.append("native time != stepwise time")
            if Proof.new_elastic(gs, merged, 1 << 20, native=True).serialize_compressed() != want:
                bad.append("native elastic != time")
            if logn < 10:
                literal = CommitterKeyStream.from_committer_key(ck, min_device_chunk=1)
                if Proof.new_elastic(gs, literal, 1 << max(2, logn - 2)).serialize_compressed() != want:
                    bad.append("stepwise elastic (literal flushes) != time")
            if case % 4 == 0:
                try:
                    V.snark_verify(snark_proof_to_ints(gm, oracle, stepwise), inst, V.VerifierKey.from_trapdoor(tau, 5))
                    stats["verified"] += 1
                except Exception as exc:  # noqa: BLE001
                    bad.append(f"verifier rejected: {exc!r}")
        finally:
            gs.free()
            g.free()
            ck.powers_of_g.free()
        stats["cases"] += 1
        stats["by_logn"][str(logn)] = stats["by_logn"].get(str(logn), 0) + 1
        if bad:
            stats["failures"].append({"case": case, "seed": seed0, "logn": logn, "nx": nx, "what": bad})
            print("SOAK FAILURE", stats["failures"][-1], flush=True)
        case += 1
    stats["seconds"] = budget
    stats["seed"] = seed0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_provers.json", "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps({k: v for k, v in stats.items() if k != "failures"}), flush=True)
    assert not stats["failures"], stats["failures"][:5]

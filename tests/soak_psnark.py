"""Time-bounded self-consistency soak of the PREPROCESSING SNARK provers on random sparse R1CS instances.  NOT collected by
default (the file name); on the GPU box:

    SOAK_SECONDS=300 python -m pytest tests/soak_psnark.py -q -s         # writes gpurun_out/soak_psnark.json

Per case: a satisfied random instance of 2^2 .. 2^10 constraints (distinct sparse A, B, a diagonal C), a fresh key of
nnz + 2 n powers (examples/psnark.rs:62 plus what a verifiable proof needs) and its index commitments; `Proof::new_time` step by
step from Python, `gm_psnark_new_time` (compiled driver; also on the instance record that `gm_psnark_preprocess` builds inside the
library, with the index commitments of `gm_psnark_index`) and, below 2^7, `Proof::new_elastic` over the stream form must produce
the same bytes (src/psnark/tests.rs:56-124), and every fourth proof must be ACCEPTED by the restated reference verifier (three
sumcheck subclaims, plookup / entry-product relations, two pairing checks: oracle/verifier_ref.py, src/psnark/verifier.rs)."""
import json
import os
import time

import numpy as np
import pytest

from tests.util import jac_to_affine_ints, psnark_proof_to_ints, random_r1cs_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def test_soak_psnark(gm, oracle, pyref):
    from gemini_amd.circuit import R1cs, R1csStream, SparseMatrix
    from gemini_amd.kzg import CommitterKey, CommitterKeyStream
    from gemini_amd.psnark import Proof
    from oracle import psnark_ref as pr
    from oracle import snark_ref as sr
    from oracle import verifier_ref as V

    budget = float(os.environ.get("SOAK_SECONDS", "30"))
    seed0 = int(os.environ.get("SOAK_SEED", "20241002"))
    t_end = time.time() + budget
    stats = {"cases": 0, "verified": 0, "elastic": 0, "by_logn": {}, "failures": []}
    case = 0
    M = lambda v: gm.fr.fr_from_int(v)  # noqa: E731
    dev = lambda rows: [[(M(v), col) for v, col in row] for row in rows]  # noqa: E731
    mont = lambda ints: oracle.fr_to_mont(oracle.ints_to_limbs(ints, 4))  # noqa: E731
    while time.time() < t_end:
        rng = np.random.default_rng(seed0 + case)
        logn = int(rng.integers(2, 11))
        n = 1 << logn
        inst, _ = random_r1cs_instance(pyref, sr, n, seed0 + 3 * case + 1)
        tau = oracle.limbs_to_ints(oracle.random_fr(seed0 + 3 * case + 2, 1))[0]
        mats = [SparseMatrix.from_rows(dev(inst[k]), n) for k in "abc"] + [SparseMatrix.from_rows(dev(inst[k]), n, transpose=True) for k in "abc"]
        r1cs = R1cs(*mats, gm.FrVec.from_host(mont(inst["z"])), gm.FrVec.from_host(mont(inst["w"])), gm.FrVec.from_host(mont(inst["x"])))
        jm = pr.sum_matrices(inst["a"], inst["b"], inst["c"], n)
        nnz = len(pr.joint_matrices(jm, inst["a"], inst["b"], inst["c"])[0])
        ck = CommitterKey.new(nnz + 2 * n, 3, oracle.ints_to_limbs([tau], 4)[0])
        bad = []
        try:
            index = Proof.index(ck, r1cs)
            stepwise = Proof.new_time(ck, r1cs, index, native=False)
            want = stepwise.serialize_compressed()
            if Proof.new_time(ck, r1cs, index, native=True).serialize_compressed() != want:
                bad.append("native time != stepwise time")
            index_lib = Proof.index(ck, r1cs, native=True)  # gm_psnark_preprocess + gm_psnark_index
            if not all((x == y).all() for x, y in zip(index_lib, index)):
                bad.append("library index != index")
            if Proof.new_time(ck, r1cs, index_lib, native="preprocess").serialize_compressed() != want:
                bad.append("native time on the library-preprocessed instance != stepwise time")
            if logn < 7:
                stream = R1csStream(r1cs)
                ck_stream = CommitterKeyStream.from_committer_key(ck)
                if Proof.new_elastic(ck_stream, stream, index, 1 << max(2, logn - 1)).serialize_compressed() != want:
                    bad.append("elastic != time")
                stream.free()
                stats["elastic"] += 1
            if case % 4 == 0:
                try:
                    V.psnark_verify(psnark_proof_to_ints(gm, oracle, stepwise), inst, V.VerifierKey.from_trapdoor(tau, 3),
                                    [jac_to_affine_ints(oracle, c) for c in index], nnz)
                    stats["verified"] += 1
                except Exception as exc:  # noqa: BLE001
                    bad.append(f"verifier rejected: {exc!r}")
        finally:
            r1cs.free()
            ck.powers_of_g.free()
        stats["cases"] += 1
        stats["by_logn"][str(logn)] = stats["by_logn"].get(str(logn), 0) + 1
        if bad:
            stats["failures"].append({"case": case, "seed": seed0, "logn": logn, "nnz": nnz, "what": bad})
            print("SOAK FAILURE", stats["failures"][-1], flush=True)
        case += 1
    stats["seconds"] = budget
    stats["seed"] = seed0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_psnark.json", "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps({k: v for k, v in stats.items() if k != "failures"}), flush=True)
    assert not stats["failures"], stats["failures"][:5]

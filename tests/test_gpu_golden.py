"""GPU parity, directly against the committed fixtures: tests/golden/{msm_small,sumcheck_small}.json hold inputs and
expected outputs produced by the big-integer statement (tests/golden/make_golden.py: naive sum_i s_i P_i; the sumcheck
recurrence of src/subprotocols/sumcheck/time_prover.rs:75-123).  test_oracle_kat.py checks the C oracle against them; this
file checks the HIP path itself, through the C ABI, with nothing in between."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def test_msm_golden_vectors_on_device(gm, oracle):
    with open(os.path.join(GOLDEN, "msm_small.json")) as fh:
        cases = json.load(fh)["cases"]
    assert len(cases) >= 10
    for case in cases:
        if not case["bases"]:
            bases = np.empty((0, 12), dtype=np.uint64)
        else:
            bases = np.stack([oracle.ints_to_affine(None if p is None else (int(p[0], 16), int(p[1], 16))) for p in case["bases"]])
        sc = oracle.ints_to_limbs([int(s, 16) for s in case["scalars"]], 4)
        exp = None if case["result"] is None else (int(case["result"][0], 16), int(case["result"][1], 16))
        # the one-shot entry (bases and scalars from the host) ...
        got = gm.VariableBaseMSM.msm_bigint(bases, sc)
        assert oracle.affine_to_ints(oracle.g1_to_affine(got)) == exp, case["name"]
        # ... and the resident-key entry
        if len(bases):
            reg = gm.G1Bases.register(bases)
            try:
                assert oracle.affine_to_ints(oracle.g1_to_affine(reg.msm_bigint(sc))) == exp, case["name"]
            finally:
                reg.free()


def test_sumcheck_golden_vectors_on_device(gm, oracle):
    with open(os.path.join(GOLDEN, "sumcheck_small.json")) as fh:
        cases = json.load(fh)["cases"]
    M = lambda ints: oracle.fr_to_mont(oracle.ints_to_limbs(ints, 4))
    I = lambda limbs: oracle.limbs_to_ints(oracle.fr_from_mont(np.asarray(limbs, dtype=np.uint64).reshape(-1, 4)))
    for case in cases:
        f = M([int(x, 16) for x in case["f"]])
        g = M([int(x, 16) for x in case["g"]])
        tw = M([int(case["twist"], 16)])[0]
        P = gm.TimeProver(f, g, tw)
        try:
            vm = None
            for rnd, (msg, ch) in enumerate(zip(case["messages"], case["challenges"])):
                a, b = P.next_message(vm)
                assert I(np.stack([a, b])) == [int(msg[0], 16), int(msg[1], 16)], (case["name"], rnd)
                vm = M([int(ch, 16)])[0]
            assert P.next_message(vm) is None
            assert I(np.stack(P.final_foldings())) == [int(x, 16) for x in case["final_foldings"]], case["name"]
        finally:
            P.free()

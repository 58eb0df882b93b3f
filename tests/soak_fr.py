"""Time-bounded randomised differential soak of the Fr vector passes and the sumcheck time prover against the CPU restatement
(oracle/gemini_oracle.c).  NOT collected by default (the file name); on the GPU box:

    SOAK_SECONDS=300 python -m pytest tests/soak_fr.py -q -s            # writes gpurun_out/soak_fr.json

Lengths are drawn log-uniformly from 1 to 2^17 (ragged: the folding levels of instances that are not powers of two), every
result is compared element for element (field arithmetic is exact).  Passes: fold_polynomial, evaluate_le (one polynomial, and
the batched entry point at x / -x / x^2 that the tensor check uses), hadamard, ip, powers, linear_combination of unequal lengths,
division by the vanishing polynomial of one to three points, and TimeProver::next_message round by round with final foldings
(src/misc.rs, src/subprotocols/sumcheck/time_prover.rs:32-106, src/kzg/time.rs:124-160)."""
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


def _mont(orc, ints):
    return orc.fr_to_mont(orc.ints_to_limbs(ints, 4))


def test_soak_fr(gm, oracle, pyref):
    from gemini_amd.fr import FrVec, div_vanishing, evaluate_le_batch

    budget = float(os.environ.get("SOAK_SECONDS", "20"))
    seed0 = int(os.environ.get("SOAK_SEED", "20240930"))
    t_end = time.time() + budget
    stats = {"cases": 0, "elements": 0, "sumcheck_rounds": 0, "max_n": 0, "failures": []}
    case = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed0 + case)
        # every 16th case is LARGE (up to 2^21 elements): above 2^18 a thread of the sumcheck kernel owns several pairs, which is where
        # the unreduced 17-limb accumulation of round 5 (k_sc_round<.., LAZY>) differs from one product per thread (SOAK_MAX_LOG overrides)
        top = float(os.environ.get("SOAK_MAX_LOG", "21" if case % 16 == 15 else "17"))
        n = int(max(1, round(2 ** rng.uniform(0, top))))
        m = int(max(1, round(2 ** rng.uniform(0, top)))) if rng.integers(0, 2) else n
        f = oracle.fr_to_mont(oracle.random_fr(seed0 + 11 * case + 1, n))
        g = oracle.fr_to_mont(oracle.random_fr(seed0 + 11 * case + 2, m))
        x = oracle.fr_to_mont(oracle.random_fr(seed0 + 11 * case + 3, 4))
        bad = []

        def check(name, ok):
            if not ok:
                bad.append(name)

        check("fold", (gm.fold_polynomial(f, x[0]).to_host() == oracle.fold_polynomial(f, x[0])).all())
        ev = gm.evaluate_le(f, x[:3])
        check("evaluate_le", all((ev[k] == oracle.evaluate_le(f, x[k])).all() for k in range(3)))
        # the tensor check's points: beta^2, beta, -beta (the third is derived from the second inside the kernel)
        b = oracle.limbs_to_ints(oracle.fr_from_mont(x[1:2]))[0]
        pts = _mont(oracle, [b * b % pyref.R_MOD, b, (-b) % pyref.R_MOD])
        vf, vg = FrVec.from_host(f), FrVec.from_host(g)
        got = evaluate_le_batch([vf, vg, vf], pts)
        for pi, poly in enumerate((f, g, f)):
            check("evaluate_le_batch", all((got[pi, k] == oracle.evaluate_le(poly, pts[k])).all() for k in range(3)))
        vf.free()
        vg.free()
        k = min(n, m)
        check("hadamard", (gm.hadamard(f[:k], g[:k]).to_host() == oracle.hadamard(f[:k], g[:k])).all())
        check("ip", (gm.ip(f[:k], g[:k]) == oracle.ip(f[:k], g[:k])).all())
        check("powers", (gm.powers(x[2], n).to_host() == oracle.powers(x[2], n)).all())
        polys = [f, g, f[: max(1, n // 3)]]
        check("lincomb", (gm.linear_combination(polys, x[:3]).to_host() == oracle.linear_combination(polys, x[:3])).all())
        kd = int(rng.integers(1, 4))
        if n > kd:
            pts_i = oracle.limbs_to_ints(oracle.random_fr(seed0 + 11 * case + 4, kd))
            q, _ = div_vanishing(f, _mont(oracle, pts_i))
            q_exp, _ = oracle.poly_div_monic(f, _mont(oracle, pyref.vanishing_polynomial(pts_i)))
            check("div_vanishing", (q.to_host() == q_exp).all())
            q.free()
        if case % 4 == 0 and max(n, m) <= (1 << 15):  # the CPU prover is the slow side
            O = oracle.TimeProver(f, g, x[3])
            G = gm.TimeProver(f, g, x[3])
            try:
                ch = oracle.fr_to_mont(oracle.random_fr(seed0 + 11 * case + 5, O.tot_rounds + 1))
                vm, r = None, 0
                while True:
                    mo, mg = O.next_message(vm), G.next_message(vm)
                    if mo is None:
                        check("sumcheck end", mg is None)
                        break
                    check("sumcheck message", mg is not None and (mg[0] == mo[0]).all() and (mg[1] == mo[1]).all())
                    vm = ch[r]
                    r += 1
                fo, fg = O.final_foldings(), G.final_foldings()
                check("final foldings", (fo[0] == fg[0]).all() and (fo[1] == fg[1]).all())
                stats["sumcheck_rounds"] += r
            finally:
                G.free()
        stats["cases"] += 1
        stats["elements"] += n + m
        stats["max_n"] = max(stats["max_n"], n, m)
        if bad:
            stats["failures"].append({"case": case, "seed": seed0, "n": n, "m": m, "what": bad})
            print("SOAK FAILURE", stats["failures"][-1], flush=True)
        case += 1
    stats["seconds"] = budget
    stats["seed"] = seed0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_fr.json", "w") as fjs:
        json.dump(stats, fjs, indent=1)
    print(json.dumps({k: v for k, v in stats.items() if k != "failures"}), flush=True)
    assert not stats["failures"], stats["failures"][:5]

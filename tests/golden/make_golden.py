#!/usr/bin/env python3
"""Generates tests/golden/*.json from oracle/pyref.py (pure Python big-int arithmetic, no C, no GPU).

The reference (Rust, arithmetic in un-vendored crates) cannot run in this image, so these vectors
are NOT reference outputs; they are the independent big-int statement of the same definitions
(naive sum_i s_i*P_i; direct evaluation of the sumcheck round polynomials) that pins both the C
oracle and the HIP path.  Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref as P  # noqa: E402


def hx(v):
    return hex(v)


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def msm_cases():
    rng = P.SplitMix64(0x474F4C44)
    cases = []

    def add(name, bases, scalars):
        cases.append({"name": name, "bases": [pt(b) for b in bases], "scalars": [hx(s) for s in scalars], "result": pt(P.msm_naive(bases, scalars))})

    for n in [1, 2, 31, 32, 33]:
        bases = [P.g1_mul(P.G1_GEN, rng.fr()) for _ in range(n)]
        add(f"random_{n}", bases, [rng.fr() for _ in range(n)])
    bases = [P.g1_mul(P.G1_GEN, rng.fr()) for _ in range(12)]
    add("special_scalars", bases, [0, 1, P.R_MOD - 1, 2, 1 << 254, (1 << 15), (1 << 16) - 1, 1 << 16, P.R_MOD - (1 << 15), 3, 7, rng.fr()])
    e = rng.fr()
    add("all_equal_scalars", bases, [e] * 12)
    add("all_equal_bases", [P.G1_GEN] * 12, [rng.fr() for _ in range(12)])
    b2 = list(bases)
    b2[3] = None
    b2[4] = None
    add("identity_bases", b2, [rng.fr() for _ in range(12)])
    add("p_minus_p", [bases[0], P.g1_neg(bases[0])], [5, 5])
    add("p_plus_p", [bases[0], bases[0]], [5, 7])
    add("empty", [], [])
    return cases


def sumcheck_cases():
    rng = P.SplitMix64(0x53554D43)
    cases = []
    for nf, ng in [(2, 2), (3, 3), (30, 30), (93, 16), (17, 1), (64, 64)]:
        f = [rng.fr() for _ in range(nf)]
        g = [rng.fr() for _ in range(ng)]
        tw = rng.fr()
        pr = P.TimeProver(f, g, tw)
        msgs, chs = [], []
        vm = None
        while True:
            m = pr.next_message(vm)
            if m is None:
                break
            vm = rng.fr()
            msgs.append([hx(m[0]), hx(m[1])])
            chs.append(hx(vm))
        ff = pr.final_foldings()
        cases.append({"name": f"time_prover_{nf}_{ng}", "f": [hx(x) for x in f], "g": [hx(x) for x in g], "twist": hx(tw),
                      "messages": msgs, "challenges": chs, "final_foldings": [hx(ff[0]), hx(ff[1])]})
    return cases


def main():
    with open(os.path.join(HERE, "msm_small.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/pyref.py)", "cases": msm_cases()}, fh, indent=0)
    with open(os.path.join(HERE, "sumcheck_small.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/pyref.py)", "cases": sumcheck_cases()}, fh, indent=0)


if __name__ == "__main__":
    main()

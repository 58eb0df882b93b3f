#!/usr/bin/env python3
"""One big MSM as ONE call vs as TWO half-size calls on the two pipelined lanes (timing probe; the second half of the
batch reuses the first half's bases, so only the time is meaningful).  Dev tool (GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gemini_amd as gm  # noqa: E402
from gemini_amd.fr import FrVec  # noqa: E402
from gemini_amd.kzg import g1_generator_mont  # noqa: E402

gm.capi.init(0)
for logn in [int(a) for a in sys.argv[1:]] or [22, 24]:
    n = 1 << logn
    rng = np.random.default_rng(5)
    reg = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    v = FrVec.from_host(bench.uniform_fr(rng, n))
    h = n // 2
    a, b = FrVec.from_host(v.to_host()[:h]), FrVec.from_host(v.to_host()[h:])
    res = {}
    for mode in ("one", "two", "one", "two"):
        for rep in range(3):
            t0 = time.perf_counter()
            if mode == "one":
                reg.msm_vec(v, n=n)
            else:
                reg.msm_vec_batch([a, b], [h, n - h])
            dt = time.perf_counter() - t0
        res.setdefault(mode, []).append(round(dt * 1e3, 2))
    print(f"2^{logn} (tables c = {reg.table_info()[0]}): one call {res['one']} ms   two halves pipelined {res['two']} ms")
    for x in (v, a, b):
        x.free()
    reg.free()

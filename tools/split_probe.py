#!/usr/bin/env python3
"""One-call MSM with and without the window-group split (gm_set_msm_split) over sizes.  Dev tool (GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gemini_amd as gm  # noqa: E402
from gemini_amd.kzg import g1_generator_mont  # noqa: E402

gm.capi.init(0)
lib = gm.capi.load()
for logn in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    n = 1 << logn
    rng = np.random.default_rng(3)
    reg = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    d = torch.from_numpy(bench.uniform_fr(rng, n).view(np.int64)).cuda()
    torch.cuda.synchronize()
    res = {}
    for split in (0, 1, 0, 1):
        gm.capi.check(lib.gm_set_msm_split(C.c_int(split)))
        for _ in range(2):
            out = reg.msm_device(d.data_ptr(), n, mont=False)
        t0 = time.perf_counter()
        for _ in range(5):
            out = reg.msm_device(d.data_ptr(), n, mont=False)
        dt = (time.perf_counter() - t0) / 5
        res.setdefault(split, []).append((round(dt * 1e3, 3), out.tobytes()))
    assert res[0][0][1] == res[1][0][1]
    print(f"2^{logn}: unsplit {[r[0] for r in res[0]]} ms   split {[r[0] for r in res[1]]} ms")
    gm.capi.check(lib.gm_set_msm_split(C.c_int(0)))
    reg.free()

#!/usr/bin/env python3
"""Sweep MSM tuning knobs (window c, chunk L, lanes-per-output of the reduce) and print stage times.
Knobs that are env-only are read once per process, so this script re-execs itself per setting."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(logn, c, steps=6):
    import torch

    import gemini_amd as gm
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    gm.capi.init(0)
    lib = gm.capi.load()
    n = 1 << logn
    rng = np.random.default_rng(1)
    ks = bench.uniform_fr(rng, n)
    from gemini_amd.kzg import g1_generator_mont

    bases = gm.G1Bases.fixed_base(g1_generator_mont(), ks)
    sc = torch.from_numpy(bench.uniform_fr(rng, n).view(np.int64)).cuda()
    torch.cuda.synchronize()
    gm.capi.check(lib.gm_set_msm_window(C.c_int(c)))
    import time

    for _ in range(2):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
    t0 = time.perf_counter()
    for _ in range(steps):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    dt = (time.perf_counter() - t0) / steps
    ms = (C.c_double * 7)()
    cnt = (C.c_uint64 * 7)()
    gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
    names = ["hist", "scan", "scatter", "acc0", "merge", "reduce"]
    st = {k: round(ms[i] / max(cnt[i], 1), 3) for i, k in enumerate(names)}
    print(json.dumps({"logn": logn, "c": c, "env": {k: v for k, v in os.environ.items() if k.startswith("GM_MSM")}, "ms": round(dt * 1e3, 3), "Mpairs/s": round(n / dt / 1e6, 1), **st}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]))
    else:
        logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
        settings = []
        for c in (14, 15, 16, 17):
            settings.append((c, {}))
        for L in (32, 64, 256):
            settings.append((16, {"GM_MSM_L": str(L)}))
        for l1, l2 in ((4, 4), (6, 4), (6, 5), (6, 6), (5, 5)):
            settings.append((16, {"GM_MSM_LPO1": str(l1), "GM_MSM_LPO2": str(l2)}))
        for c, env in settings:
            e = dict(os.environ)
            e.update(env)
            subprocess.run([sys.executable, __file__, "one", str(logn), str(c)], env=e)

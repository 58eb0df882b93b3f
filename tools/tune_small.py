#!/usr/bin/env python3
"""Window-width sweep for small one-call MSMs (dev tool, GPU box): python tools/tune_small.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import gemini_amd as gm
from gemini_amd.kzg import g1_generator_mont
gm.capi.init(0); lib = gm.capi.load()
rng = np.random.default_rng(3)
N = 1 << 15
bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, N))
sc = torch.from_numpy(bench.uniform_fr(rng, N).view(np.int64)).cuda(); torch.cuda.synchronize()
for lg in range(3, 16):
    n = 1 << lg
    row = []
    for c in range(max(2, lg - 5), min(17, lg + 3)):
        gm.capi.check(lib.gm_set_msm_window(C.c_int(c)))
        for _ in range(2): bases.msm_device(sc.data_ptr(), n, mont=False)
        t0 = time.perf_counter()
        for _ in range(8): bases.msm_device(sc.data_ptr(), n, mont=False)
        row.append((c, (time.perf_counter() - t0) / 8 * 1e3))
    gm.capi.check(lib.gm_set_msm_window(C.c_int(0)))
    best = min(row, key=lambda r: r[1])
    print(f"2^{lg}: best c={best[0]} {best[1]:.3f} ms | " + " ".join(f"{c}:{t:.3f}" for c, t in row), flush=True)

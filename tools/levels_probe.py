#!/usr/bin/env python3
"""MSM time vs gm_set_msm_affine_levels at a few sizes (one-call MSMs, scalars resident)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemini_amd as gm  # noqa: E402
from gemini_amd.fr import FrVec  # noqa: E402

gm.capi.init()
lib = gm.capi.load()
rng = np.random.default_rng(1)
gx = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
gy = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
mont = lambda v: [(((v << 384) % q) >> (64 * i)) & (2**64 - 1) for i in range(6)]
g_aff = np.array(mont(gx) + mont(gy), dtype=np.uint64)


def rand(n):
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


for logn in [int(x) for x in (sys.argv[1:] or ["20", "22"])]:
    n = 1 << logn
    bases = gm.G1Bases.fixed_base(g_aff, rand(n))
    v = FrVec.from_host(rand(n))
    want = None
    for lv in ([0, int(os.environ["GM_PROBE_LEVELS"])] if "GM_PROBE_LEVELS" in os.environ else (0, 1, 2, 3, 4, 5, -1)):
        gm.capi.check(lib.gm_set_msm_affine_levels(C.c_int(lv)))
        r = bases.msm_vec(v)
        if want is None:
            want = r
        assert (r == want).all(), (logn, lv)
        gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            bases.msm_vec(v)
        dt = (time.perf_counter() - t0) / reps
        ms = (C.c_double * 7)()
        cnt = (C.c_uint64 * 7)()
        gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
        gm.capi.check(lib.gm_prof_enable(C.c_int(0)))
        st = [round(ms[i] / cnt[i], 3) if cnt[i] else 0 for i in range(6)]
        print(f"logn {logn} levels {lv:2d}: {dt*1e3:8.3f} ms  {n/dt/1e6:7.1f} Mscalar/s  sort {st[0]+st[2]:.3f} levels {st[1]:.3f} acc0 {st[3]:.3f} merge {st[4]:.3f} reduce {st[5]:.3f}", flush=True)
    gm.capi.check(lib.gm_set_msm_affine_levels(C.c_int(0)))
    v.free()
    bases.free()

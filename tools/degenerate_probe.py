#!/usr/bin/env python3
"""Stage times of one-call MSMs on the reference's benchmark instance shapes -- all scalars equal (the witness of
dummy_r1cs, src/circuit.rs:349-365) and all BASES equal (the generator-copies key of the elastic example,
examples/snark.rs:59-63: the second entry of every bucket run is a doubling) -- next to uniform inputs.
Dev tool (GPU box): python tools/degenerate_probe.py 24"""
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
names = ["digits_hist", "scan", "scatter", "acc0", "merge", "reduce", "sc_round"]
for logn in [int(a) for a in sys.argv[1:]] or [24]:
    n = (1 << logn) - 1
    rng = np.random.default_rng(1)
    bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    uni = bench.uniform_fr(rng, n)
    eq = np.repeat(bench.uniform_fr(rng, 1), n, axis=0)
    for label, sc in (("uniform", uni), ("all-equal", eq)):
        d = torch.from_numpy(sc.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for _ in range(2):
            bases.msm_device(d.data_ptr(), n, mont=False)
        gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
        t0 = time.perf_counter()
        for _ in range(3):
            bases.msm_device(d.data_ptr(), n, mont=False)
        dt = (time.perf_counter() - t0) / 3
        ms = (C.c_double * 7)()
        cnt = (C.c_uint64 * 7)()
        gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
        gm.capi.check(lib.gm_prof_enable(C.c_int(0)))
        print(f"2^{logn}-1 {label:10s} {dt * 1e3:8.2f} ms  " + "  ".join(f"{k}={ms[i] / cnt[i]:.2f}" for i, k in enumerate(names) if cnt[i]))
    bases.free()
    # every base the generator, uniform scalars; no tables (the example's key of 2^28 + 1 copies gets none either)
    gm.capi.check(lib.gm_set_auto_tables(C.c_int(0), C.c_size_t(0)))
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = 1
    copies = gm.G1Bases.fixed_base(g1_generator_mont(), ones)
    bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    for label, reg in (("uniform, no tables", bases), ("all bases equal", copies)):
        d = torch.from_numpy(uni.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for _ in range(2):
            reg.msm_device(d.data_ptr(), n, mont=False)
        gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
        t0 = time.perf_counter()
        for _ in range(3):
            reg.msm_device(d.data_ptr(), n, mont=False)
        dt = (time.perf_counter() - t0) / 3
        ms = (C.c_double * 7)()
        cnt = (C.c_uint64 * 7)()
        gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
        gm.capi.check(lib.gm_prof_enable(C.c_int(0)))
        print(f"2^{logn}-1 {label:18s} {dt * 1e3:8.2f} ms  " + "  ".join(f"{k}={ms[i] / cnt[i]:.2f}" for i, k in enumerate(names) if cnt[i]))
    gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(0)))
    bases.free()
    copies.free()

#!/usr/bin/env python3
"""Development probe (needs a library built with `make -C gemini_amd/csrc EXTRA=-DGM_ACC0_CYCLES`): the shader clock DURING
k_acc0 in the bench loop and the cycles a wave spends per entry, from clock64() / wall_clock64() readings inside the kernel.
Compare with tools/madd_cycles.hip (the statement alone: 27.8 k cycles per wave and entry at two waves per SIMD)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    import gemini_amd as gm
    from gemini_amd.kzg import g1_generator_mont

    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    gm.capi.init(0)
    lib = gm.capi.load()
    lib.gm_set_auto_tables(C.c_int(int(os.environ.get("TABLES", "0"))), C.c_size_t(0))
    os.environ.setdefault("GM_ACC0_PREFETCH", "0")  # the instrumented kernel is the plain one
    n = 1 << logn
    rng = np.random.default_rng(1)
    bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    sc = torch.from_numpy(bench.uniform_fr(rng, n).view(np.int64)).cuda()
    torch.cuda.synchronize()
    out = (C.c_uint64 * 3)()
    for _ in range(20):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    lib.gm_debug_acc0_cycles(out, C.c_int(1))
    steps = 50
    t0 = time.perf_counter()
    for _ in range(steps):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    dt = (time.perf_counter() - t0) / steps
    lib.gm_debug_acc0_cycles(out, C.c_int(1))
    cyc, ticks, waves = out[0], out[1], out[2]
    c, tb = bases.table_info()
    W = 13 if c == 20 else (12 if c == 22 else 16)
    entries_per_wave = n * W * steps / waves
    print(json.dumps({"logn": logn, "tables_c": c, "ms_per_msm": round(dt * 1e3, 3), "waves_per_launch": waves // steps,
                      "clock_MHz_during_k_acc0": round(cyc / ticks * 100.0, 1), "cycles_per_wave": round(cyc / waves),
                      "entries_per_wave": round(entries_per_wave, 2), "cycles_per_wave_and_entry": round(cyc / waves / entries_per_wave),
                      "wave_resident_us": round(ticks / waves / 100.0, 1)}))


if __name__ == "__main__":
    main()

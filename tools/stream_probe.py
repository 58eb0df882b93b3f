#!/usr/bin/env python3
"""Throughput of the streaming MSM over host-resident pairs (gm_g1_msm_stream_*), pageable vs page-locked host buffers,
pairs carrying their bases vs scalars only against the resident key.  Dev tool (GPU box): python tools/stream_probe.py 24"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import bench  # noqa: E402
import gemini_amd as gm  # noqa: E402
from gemini_amd.kzg import g1_generator_mont  # noqa: E402
from gemini_amd.msm import HostMsmStream, pinned_empty  # noqa: E402

gm.capi.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
rng = np.random.default_rng(5)
reg = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
sc = bench.uniform_fr(rng, n)
hb = reg.download()
pb, ps = pinned_empty((n, 12)), pinned_empty((n, 4))
pb[:] = hb
ps[:] = sc
d = torch.from_numpy(sc.view(np.int64)).cuda()
torch.cuda.synchronize()
reg.msm_device(d.data_ptr(), n, mont=False)
t0 = time.perf_counter()
ref = reg.msm_device(d.data_ptr(), n, mont=False)
dt = time.perf_counter() - t0
print(f"2^{logn} pairs; everything resident, one call: {dt * 1e3:8.2f} ms  {n / dt / 1e6:7.1f} Mpairs/s")
for chunk_log in (20, 22, 24):
    if chunk_log > logn:
        continue
    for label, kw, b, s in (("bases+scalars pageable", {}, hb, sc), ("bases+scalars pinned  ", {}, pb, ps),
                            ("scalars pageable, key resident", {"bases": reg}, None, sc), ("scalars pinned, key resident  ", {"bases": reg}, None, ps)):
        st = HostMsmStream(1 << chunk_log, **kw)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            st.add(b, s)
            out = st.finalize()
            best = min(best, time.perf_counter() - t0)
        assert (out == ref).all()
        st.free()
        gb = n * (32 + (96 if b is not None else 0)) / 1e9
        print(f"  chunk 2^{chunk_log}  {label}: {best * 1e3:8.2f} ms  {n / best / 1e6:7.1f} Mpairs/s  ({gb / best:5.1f} GB/s over PCIe)")

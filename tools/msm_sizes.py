#!/usr/bin/env python3
"""MSM wall time vs size (latency floor of the small MSMs that batch_commit issues)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemini_amd as gm
import bench
from gemini_amd.kzg import g1_generator_mont
gm.capi.init(0)
N = 1 << 22
rng = np.random.default_rng(5)
bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, N))
sc = torch.from_numpy(bench.uniform_fr(rng, N).view(np.int64)).cuda()
torch.cuda.synchronize()
for lg in range(1, 23):
    n = 1 << lg
    for _ in range(2):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    t0 = time.perf_counter(); k = 5
    for _ in range(k):
        bases.msm_device(sc.data_ptr(), n, mont=False)
    dt = (time.perf_counter() - t0) / k
    print(json.dumps({"logn": lg, "ms": round(dt * 1e3, 3), "Mpairs/s": round(n / dt / 1e6, 2)}), flush=True)

import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemini_amd as gm
import bench
from gemini_amd.kzg import g1_generator_mont
gm.capi.init(0); lib = gm.capi.load()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
rng = np.random.default_rng(1)
bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
sc = torch.from_numpy(bench.uniform_fr(rng, n).view(np.int64)).cuda(); torch.cuda.synchronize()
def run(tag):
    for _ in range(2): bases.msm_device(sc.data_ptr(), n, mont=False)
    gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
    t0 = time.perf_counter()
    for _ in range(6): bases.msm_device(sc.data_ptr(), n, mont=False)
    dt = (time.perf_counter() - t0) / 6
    ms = (C.c_double * 7)(); cnt = (C.c_uint64 * 7)()
    gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
    names = ["sort1", "scan", "sort2", "acc0", "merge", "reduce"]
    print(tag, round(dt*1e3,3), {k: round(ms[i]/max(cnt[i],1),3) for i,k in enumerate(names)}, flush=True)
run("plain c=16")
for c in ([int(x) for x in sys.argv[2:]] or [16, 18, 20, 21]):
    t0=time.perf_counter(); bases.precompute(c); tp=time.perf_counter()-t0
    run(f"tables c={c} (pre {tp:.2f}s)")

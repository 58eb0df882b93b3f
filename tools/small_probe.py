import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import gemini_amd as gm
from gemini_amd.kzg import g1_generator_mont
gm.capi.init(0); lib = gm.capi.load()
names = ["digits_hist", "scan", "scatter", "acc0", "merge", "reduce", "sc_round"]
rng = np.random.default_rng(1)
for logn in (3, 8, 10, 12, 14, 16):
    n = 1 << logn
    bases = gm.G1Bases.fixed_base(g1_generator_mont(), bench.uniform_fr(rng, n))
    d = torch.from_numpy(bench.uniform_fr(rng, n).view(np.int64)).cuda(); torch.cuda.synchronize()
    for _ in range(3): bases.msm_device(d.data_ptr(), n, mont=False)
    t0 = time.perf_counter()
    for _ in range(20): bases.msm_device(d.data_ptr(), n, mont=False)
    plain = (time.perf_counter() - t0) / 20
    gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
    for _ in range(10): bases.msm_device(d.data_ptr(), n, mont=False)
    ms = (C.c_double * 7)(); cnt = (C.c_uint64 * 7)()
    gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7))); gm.capi.check(lib.gm_prof_enable(C.c_int(0)))
    st = {k: ms[i] / cnt[i] for i, k in enumerate(names) if cnt[i]}
    print(f"2^{logn}: {plain*1e3:.3f} ms  stages sum {sum(st.values()):.3f}  " + " ".join(f"{k}={v:.3f}" for k, v in st.items()))
    bases.free()

#!/usr/bin/env python3
"""The folding-commitment batch in isolation: k MSMs of 2^top, 2^(top-1), ..., 2 pairs against one resident key through
gm_g1_msm_v_batch (CommitterKey::batch_commit, src/kzg/time.rs:98-107), and the same levels one call at a time.  Dev tool (GPU box).
  GM_MSM_SMALL_LANES=n python tools/batch_probe.py [top] [tables]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import ctypes as C

    import gemini_amd as gm
    from gemini_amd.fr import FrVec
    from gemini_amd.kzg import g1_generator_mont
    from gemini_amd.msm import G1Bases

    top = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    tables = len(sys.argv) > 2 and sys.argv[2] == "tables"
    gm.capi.init()
    lib = gm.capi.load()
    gm.capi.check(lib.gm_set_auto_tables(C.c_int(int(tables)), C.c_size_t(0)))
    rng = np.random.default_rng(5)
    tau = rng.integers(0, 2**62, size=4, dtype=np.uint64)
    key = G1Bases.srs(g1_generator_mont(), tau, (1 << top) + 1)
    vecs = []
    for lg in range(top, 0, -1):
        h = rng.integers(0, 2**62, size=(1 << lg, 4), dtype=np.uint64)
        vecs.append(FrVec.from_host(h))
    ns = [len(v) for v in vecs]
    for _ in range(2):
        key.msm_vec_batch(vecs, ns)
    ts = []
    for _ in range(5):
        time.sleep(0.02)  # an idle gap a kernel trace can be cut at (tools/batch_timeline.py)
        t0 = time.perf_counter()
        key.msm_vec_batch(vecs, ns)
        ts.append(time.perf_counter() - t0)
    if os.environ.get("PROBE_BATCH_ONLY") == "1":
        print(f"batch {min(ts) * 1e3:.3f} ms")
        return
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        for v in vecs:
            key.msm_vec(v)
        t1.append(time.perf_counter() - t0)
    per = []
    for v in vecs:
        key.msm_vec(v)
        t0 = time.perf_counter()
        key.msm_vec(v)
        per.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(f"top 2^{top} tables={tables} lanes={os.environ.get('GM_MSM_SMALL_LANES', 'default')}: batch {min(ts) * 1e3:.3f} ms, one by one {min(t1) * 1e3:.3f} ms; per level (ms, 2^{top} down): {per}")


if __name__ == "__main__":
    main()

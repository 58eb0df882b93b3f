#!/usr/bin/env python3
"""Which commitments of a preprocessing-SNARK proof are slow for their size?  Every MSM the prover issues is re-run alone
with the stage timers on and printed with a fingerprint of its scalar vector (distinct values / bit length of a sample).
Dev tool (GPU box): python tools/psnark_msm_probe.py 20"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import gemini_amd as gm  # noqa: E402
from gemini_amd import fr as F  # noqa: E402
from gemini_amd.circuit import dummy_r1cs  # noqa: E402
from gemini_amd.kzg import CommitterKey  # noqa: E402
from gemini_amd.msm import G1Bases  # noqa: E402
from gemini_amd.psnark import Proof  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gm.capi.init(0)
lib = gm.capi.load()
names = ["digits_hist", "scan", "scatter", "acc0", "merge", "reduce", "sc_round"]
rows = []
orig = G1Bases.msm_vec


def probe(self, vec, n, tag):
    if n < (1 << 16):
        return
    h = vec.to_host()[:n]
    ints = F.fr_into_bigint(h[:: max(1, n // 4096)][:4096])
    vals = [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in ints]
    head = F.fr_into_bigint(h[:64])
    hv = [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in head]
    for _ in range(2):
        orig(self, vec, n=n)
    gm.capi.check(lib.gm_prof_enable(C.c_int(1)))
    t0 = time.perf_counter()
    for _ in range(3):
        orig(self, vec, n=n)
    dt = (time.perf_counter() - t0) / 3
    ms = (C.c_double * 7)()
    cnt = (C.c_uint64 * 7)()
    gm.capi.check(lib.gm_prof_read(ms, cnt, C.c_int(7)))
    gm.capi.check(lib.gm_prof_enable(C.c_int(0)))
    st = "  ".join(f"{k}={ms[i] / cnt[i]:.2f}" for i, k in enumerate(names) if cnt[i])
    rows.append((dt * 1e3 / (n / 1e6), f"{tag:6s} n={n:9d} {dt * 1e3:8.2f} ms  {n / dt / 1e6:7.1f} M/s  sample: distinct {len(set(vals)):4d}/{len(vals)} "
                 f"bits<= {max(v.bit_length() for v in vals):3d} zeros {sum(v == 0 for v in vals):4d}  first64 distinct {len(set(hv)):2d}  | {st}"))
    print(rows[-1][1], flush=True)


def msm_vec(self, vec, n=None, **kw):
    probe(self, vec, len(vec) if n is None else n, "one")
    return orig(self, vec, n=n, **kw)


orig_batch = G1Bases.msm_vec_batch


def msm_vec_batch(self, vecs, ns, **kw):
    for v, n in zip(vecs, ns):
        probe(self, v, n, "batch")
    return orig_batch(self, vecs, ns, **kw)


n = 1 << logn
rng = np.random.default_rng(2022420)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD  # noqa: E731
r1cs = dummy_r1cs(rnd(), n)
tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
ck = CommitterKey.new(2 * n, 5, tau)
G1Bases.msm_vec = msm_vec
G1Bases.msm_vec_batch = msm_vec_batch
print("== index")
index = Proof.index(ck, r1cs)
print("== new_time")
proof = Proof.new_time(ck, r1cs, index)
print("== slowest per scalar")
for _, line in sorted(rows, reverse=True)[:12]:
    print(line)

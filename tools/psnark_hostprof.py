#!/usr/bin/env python3
"""cProfile of Proof.new_time (psnark) at a mid size: where does the HOST time go?  Dev tool (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import gemini_amd as gm  # noqa: E402
from gemini_amd.circuit import dummy_r1cs  # noqa: E402
from gemini_amd.kzg import CommitterKey  # noqa: E402

which = sys.argv[2] if len(sys.argv) > 2 else "psnark"
if which == "psnark":
    from gemini_amd.psnark import Proof  # noqa: E402
else:
    from gemini_amd.snark import Proof  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gm.capi.init(0)
n = 1 << logn
rng = np.random.default_rng(2022420)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD  # noqa: E731
r1cs = dummy_r1cs(rnd(), n)
tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
ck = CommitterKey.new(2 * n, 5, tau)
if which == "psnark":
    index = Proof.index(ck, r1cs)
    run = lambda: Proof.new_time(ck, r1cs, index)  # noqa: E731
else:
    run = lambda: Proof.new_time(r1cs, ck)  # noqa: E731
import warnings

warnings.simplefilter("ignore")
run()
t0 = time.perf_counter()
run()
print("wall", round(time.perf_counter() - t0, 4))
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)

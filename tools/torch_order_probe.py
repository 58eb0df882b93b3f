#!/usr/bin/env python3
"""Which load order of libgemini_hip.so and torch leaves torch with a GPU?  (gemini_amd/capi.py::_load_torch_runtime_first)
Dev tool (GPU box): GM_NO_TORCH_PRELOAD=1 python tools/torch_order_probe.py lib_first 1000"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import numpy as np
if mode == "torch_import_first":
    import torch
import gemini_amd as gm
gm.capi.init(0)
from gemini_amd.kzg import g1_generator_mont
one = np.array([5, 0, 0, 0], dtype=np.uint64)
reg = gm.G1Bases.srs(g1_generator_mont(), one, int(sys.argv[2]))
import torch
try:
    if mode == "generator":
        g = torch.Generator(device="cuda")
    x = torch.zeros(4).cuda()
    print(mode, sys.argv[2], "ok", x.device)
except Exception as e:
    print(mode, sys.argv[2], "FAILED", str(e).splitlines()[0])

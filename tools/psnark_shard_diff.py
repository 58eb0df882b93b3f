#!/usr/bin/env python3
"""debug aid: gm_psnark_new_time vs gm_psnark_new_time_sharded (world 1 or N over shm) field by field"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemini_amd as gm
from gemini_amd import collective
from gemini_amd.circuit import dummy_r1cs
from gemini_amd.kzg import CommitterKey
from gemini_amd.psnark import Proof
from gemini_amd.sharded import PsnarkShard, PsnarkShardKey, psnark_new_time_sharded

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
gm.capi.init(0)
collective.init_shm(rank, world, "/gm_diff_%s" % os.environ.get("MASTER_PORT", "0"))
n = 1 << logn
rng = np.random.default_rng(2022420)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD
r1cs = dummy_r1cs(rnd(), n)
tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
import warnings
warnings.filterwarnings("ignore")
shard = PsnarkShard(r1cs, tail_log=tail)
key = PsnarkShardKey(2 * n, shard.block, tail, tau)
idx_s = shard.index(key)
ps = psnark_new_time_sharded(shard, key, idx_s)
collective.finalize()
ck = CommitterKey.new(2 * n, 5, tau)
idx = Proof.index(ck, r1cs)
p = Proof.new_time(ck, r1cs, idx, native=True)
if rank == 0:
    print("block", shard.block, "longest", shard.longest, "segments", key.segments)
    print("index equal:", all((a == b).all() for a, b in zip(idx, idx_s)))
    def cmp(name, a, b):
        if isinstance(a, (list, tuple)):
            if len(a) != len(b):
                print("DIFF len", name, len(a), len(b)); return
            for i, (x, y) in enumerate(zip(a, b)):
                cmp(f"{name}[{i}]", x, y)
        elif isinstance(a, np.ndarray):
            if a.shape != b.shape or not (a == b).all():
                print("DIFF", name)
        elif hasattr(a, "__dict__"):
            for k in a.__dict__:
                if k != "spans": cmp(f"{name}.{k}", getattr(a, k), getattr(b, k))
        else:
            if a != b: print("DIFF", name, a, b)
    cmp("proof", p, ps)
    print("done")

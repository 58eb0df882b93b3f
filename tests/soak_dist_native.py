"""Sweep of the N-GPU provers compiled into the library (gemini_amd/csrc/sharded.cpp over dist.cpp): instance sizes, world sizes,
tail lengths, transports (shm / gloo hook), the block-diagonal and the general-matrix form of the dummy instance, random general
R1CS instances, the cyclic-key native provers (snark time / elastic, psnark time / elastic), and the block-sharded psnark (any world
size 1 .. 9, dummy / random general instances, the three key shapes; SOAK_ONLY=psnark: only those) -- every configuration compared
with the single-GPU proof of the same instance through its SHA-256.  NOT collected by default (the file name); all ranks share
the one GPU of the test box:

    SOAK_SECONDS=600 python -m pytest tests/soak_dist_native.py -q -s          # writes gpurun_out/soak_dist_native.json"""
import json
import os
import time

import numpy as np
import pytest

from tests.test_gpu_dist_native import _run

pytestmark = pytest.mark.gpu


def test_soak_dist_native():
    budget = float(os.environ.get("SOAK_SECONDS", "60"))
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "20260929")))
    t_end = time.time() + budget
    single = {}
    stats = {"configs": [], "failures": []}

    def one(key, extra, tool, logn):
        if key not in single:
            single[key] = _run(1, extra, tool, logn, transport=None)["proof_sha256"]
        return single[key]

    only_psnark = os.environ.get("SOAK_ONLY", "") == "psnark"
    while time.time() < t_end:
        kind = int(rng.integers(10, 16)) if only_psnark else int(rng.integers(0, 16))
        transport = "hook" if rng.integers(0, 4) == 0 else "shm"
        if kind >= 10:  # block-sharded psnark (gm_psnark_new_time_sharded): ANY world size, dummy or random general instance, time or elastic key
            logn = int(rng.integers(3, 13))
            world = int(rng.integers(1, 10))
            tail_log = int(rng.integers(2, 9))
            extra = []
            if kind >= 13:
                logn = min(logn, 9)
                extra += ["--random-r1cs", str(int(rng.integers(1, 1 << 30)))]
            elif rng.integers(0, 3) == 0:
                extra += ["--elastic"]
            elif rng.integers(0, 3) == 0:
                extra += ["--verifiable-key"]
            want = one(("psnark", logn, tuple(extra)), extra, "run_psnark.py", logn)
            cfg = {"kind": "psnark block", "logn": logn, "world": world, "tail_log": tail_log, "extra": extra, "transport": transport}
            got = _run(world, extra + ["--block-sharded", "--tail-log", str(tail_log)], "run_psnark.py", logn, transport=transport)["proof_sha256"]
        elif kind < 5:  # block-sharded, dummy instance (local or global columns)
            logn = int(rng.integers(8, 18))
            world = int([1, 2, 4, 8, 16][int(rng.integers(0, 5))])
            m_log = logn - world.bit_length() + 1
            if m_log < 3:
                continue
            tail_log = int(rng.integers(3, min(m_log, 11) + 1))
            extra = ["--block-sharded", "--tail-log", str(tail_log)] + (["--global-columns"] if rng.integers(0, 2) else [])
            want = one(("snark", logn), [], "run_snark.py", logn)
            cfg = {"kind": "block", "logn": logn, "world": world, "tail_log": tail_log, "global": "--global-columns" in extra, "transport": transport}
            got = _run(world, extra, "run_snark.py", logn, transport=transport)["proof_sha256"]
        elif kind < 7:  # block-sharded, random general R1CS
            logn = int(rng.integers(6, 12))
            world = int([1, 2, 4, 8][int(rng.integers(0, 4))])
            m_log = logn - world.bit_length() + 1
            if m_log < 3:
                continue
            tail_log = int(rng.integers(3, min(m_log, 8) + 1))
            seed = int(rng.integers(1, 1 << 30))
            want = one(("rand", logn, seed), ["--random-r1cs", str(seed)], "run_snark.py", logn)
            cfg = {"kind": "general", "logn": logn, "world": world, "tail_log": tail_log, "seed": seed, "transport": transport}
            got = _run(world, ["--random-r1cs", str(seed), "--block-sharded", "--tail-log", str(tail_log)], "run_snark.py", logn, transport=transport)["proof_sha256"]
        else:  # cyclic key under the native provers
            tool = "run_psnark.py" if kind == 9 else "run_snark.py"
            logn = int(rng.integers(4, 13 if tool == "run_psnark.py" else 16))
            world = int(rng.integers(2, 8))
            extra = ["--elastic"] if rng.integers(0, 2) else []
            want = one((tool, logn, tuple(extra)), extra, tool, logn)
            cfg = {"kind": "cyclic " + tool, "logn": logn, "world": world, "elastic": bool(extra), "transport": transport}
            got = _run(world, extra, tool, logn, transport=transport)["proof_sha256"]
        cfg["ok"] = bool(got == want)
        stats["configs"].append(cfg)
        if not cfg["ok"]:
            stats["failures"].append(cfg)
            print("SOAK FAILURE", cfg, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_dist_native.json", "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps({"configs": len(stats["configs"]), "failures": len(stats["failures"])}), flush=True)
    assert not stats["failures"], stats["failures"][:5]

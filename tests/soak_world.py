"""Sweep of the block-sharded prover (tests/stepwise/dist_prover.py) over instance sizes, world sizes and tail lengths, every
configuration compared with the single-GPU proof of the same instance through its SHA-256 and with the pure work model.  NOT
collected by default (the file name); all ranks share the one GPU of the test box over gloo:

    SOAK_SECONDS=600 python -m pytest tests/soak_world.py -q -s          # writes gpurun_out/soak_world.json"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, logn, extra):
    env = dict(os.environ, GM_BENCH_BACKEND="gloo", GM_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = [os.path.join(ROOT, "tools", "run_snark.py"), "-i", str(logn), "--repeat", "1"] + extra
    if world == 1:
        cmd = [sys.executable] + script
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port())] + script
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_soak_world():
    from tests.stepwise.dist_prover import fr_work

    budget = float(os.environ.get("SOAK_SECONDS", "60"))
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "20241003")))
    t_end = time.time() + budget
    single = {}
    stats = {"configs": [], "failures": []}
    while time.time() < t_end:
        logn = int(rng.integers(10, 19))
        world = int([2, 4, 8][int(rng.integers(0, 3))])
        m_log = logn - world.bit_length() + 1
        tail_log = int(rng.integers(3, min(m_log, 11) + 1))
        if logn not in single:
            single[logn] = _run(1, logn, [])["proof_sha256"]
        many = _run(world, logn, ["--block-sharded", "--tail-log", str(tail_log)])
        ok = many["proof_sha256"] == single[logn] and many["fr_work"] == fr_work(1 << logn, world, tail_log)
        cfg = {"logn": logn, "world": world, "tail_log": tail_log, "ok": bool(ok)}
        stats["configs"].append(cfg)
        if not ok:
            stats["failures"].append(cfg)
            print("SOAK FAILURE", cfg, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_world.json", "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps({"configs": len(stats["configs"]), "failures": len(stats["failures"])}), flush=True)
    assert not stats["failures"], stats["failures"][:5]

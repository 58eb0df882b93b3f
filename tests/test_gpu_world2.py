"""Two ranks over gloo sharing the one GPU of the test box (the N > 1 code path of tests/stepwise/dist.py with the real
device library on every rank): `snark --time-prover` and the elastic prover over the element-cyclic sharded KZG key
must produce the proof of the single-GPU run, byte for byte (compared through its SHA-256).  The driver's multi-GPU
bench launches the same entry points with the nccl backend, one rank per GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_SINGLE = {}


def _single(extra=()):
    """the single-GPU run of the same recipe, once per module"""
    key = tuple(extra)
    if key not in _SINGLE:
        _SINGLE[key] = _run(1, list(extra))
    return _SINGLE[key]


def _run(world, extra, tool="run_snark.py", logn=12):
    env = dict(os.environ, GM_BENCH_BACKEND="gloo", GM_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = [os.path.join(ROOT, "tools", tool), "-i", str(logn), "--repeat", "1"] + extra
    if world == 1:
        cmd = [sys.executable] + script
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port())] + script
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("extra", [[]], ids=["time"])  # (this file covers the step-wise Python composition, tests/stepwise: the cross-check, not the product)
def test_two_ranks_one_gpu_same_proof(extra):
    one = _single(extra)
    for world in (2, 3):
        many = _run(world, extra)
        assert many["n_gpus"] == world
        assert many["proof_sha256"] == one["proof_sha256"], (world, extra)



@pytest.mark.parametrize("extra", [["--elastic"]], ids=["elastic"])
def test_psnark_two_and_three_ranks_one_gpu_same_proof(extra):
    """BASELINE configs[4] is the 8-GPU preprocessing SNARK (examples/psnark.rs:54-81): `psnark` time and elastic provers over
    the element-cyclic sharded key on 2 and 3 ranks must produce the single-GPU proof byte for byte (src/psnark/tests.rs:14-125
    holds time == elastic on one key; here every rank count must agree with one GPU)."""
    one = _run(1, list(extra), tool="run_psnark.py", logn=10)
    for world in ((2,) if extra else (3,)):  # the compiled provers run 2 AND 3 ranks of both (tests/test_gpu_dist_native.py)
        many = _run(world, list(extra), tool="run_psnark.py", logn=10)
        assert many["n_gpus"] == world
        assert many["proof_sha256"] == one["proof_sha256"], (world, extra)


@pytest.mark.parametrize("tail_log", [6])
def test_block_sharded_prover_same_proof(tail_log):
    """tests/stepwise/dist_prover.py: the field arithmetic sharded as well (block-sharded vectors, per-level key slices, sumchecks
    through ShardedTimeProver, the opening through per-block carries): 1, 2, 4 (and 8) ranks on the one GPU of the test box must
    produce the single-GPU proof byte for byte.  tail_log 4 / 6 at 2^12 constraints: 6 / 4 sharded levels at 4 ranks."""
    from tests.stepwise.dist_prover import fr_work

    one = _single()
    for world in ((2,) if tail_log == 4 else (4,)):  # 8 ranks: blocks of 512 constraints, 5 sharded levels (1 / 2 / 4 / 8 of the compiled prover: test_gpu_dist_native.py)
        many = _run(world, ["--block-sharded", "--tail-log", str(tail_log)])
        assert many["n_gpus"] == world
        assert many["proof_sha256"] == one["proof_sha256"], (world, tail_log)
        # what rank 0's device passes read + wrote, counted as they ran, is what the pure model says (the CPU suite evaluates the
        # model at 2^24 constraints for 2, 4, 8 ranks: tests/test_multi_gpu_gloo.py)
        assert many["fr_work"] == fr_work(1 << 12, world, tail_log), (world, many["fr_work"], fr_work(1 << 12, world, tail_log))

"""The shared-memory transport of the library's collective layer (gemini_amd/csrc/dist.cpp: gm_dist_init_shm) with N real
processes and NO GPU: payloads from 8 bytes to several slots, thousands of back-to-back calls (the two-bank protocol), the
hook transport against it, and the error paths.  The sharded provers call exactly these all-gathers (tests/test_gpu_dist_native.py
runs them on the device)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, ctypes as C
    import numpy as np
    sys.path.insert(0, %r)
    os.environ["GM_NO_TORCH_PRELOAD"] = "1"
    from gemini_amd import capi, collective
    rank, world, name, slot = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    collective.init_shm(rank, world, name, slot)
    assert collective.info() == (rank, world, "shm")
    collective.selftest()
    rng = np.random.default_rng(7)
    sizes = [1, 8, 18, 1000] + [int(s) for s in rng.integers(1, 3 * max(slot, 64) // 8 + 5, size=12)]
    for it, words in enumerate(sizes):
        mine = (np.arange(words, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(rank * 1000003 + it))
        got = collective.allgather_host(mine)
        assert got.shape == (world, words)
        for r in range(world):
            want = (np.arange(words, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(r * 1000003 + it))
            assert (got[r] == want).all(), (rank, r, words)
    # back-to-back small calls: a fast rank must never overwrite a bank a slow peer still reads
    acc = 0
    for it in range(3000):
        got = collective.allgather_host(np.array([rank, it, acc & 0xffff], dtype=np.uint64))
        assert (got[:, 0] == np.arange(world)).all() and (got[:, 1] == it).all() and (got[:, 2] == (acc & 0xffff)).all(), (rank, it)
        acc += int(got.sum())
        if it %% 500 == rank:
            import time; time.sleep(0.01)
    st = collective.stats()
    assert st["collectives"] >= 3000 + len(sizes)
    collective.finalize()
    assert collective.info() == (0, 1, "none")
    print("ok", rank, acc)
""") % ROOT


def _spawn(world, slot):
    name = f"/gm_test_{os.getpid()}_{world}_{slot}"
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), name, str(slot)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    accs = {o.split()[-1] for o, _ in outs}
    assert len(accs) == 1  # every rank saw the same data throughout
    assert not os.path.exists("/dev/shm" + name)  # the last rank out unlinks the segment


@pytest.mark.parametrize("world,slot", [(2, 0), (3, 256), (4, 4096), (8, 64)])
def test_shm_allgather_between_processes(world, slot):
    _spawn(world, slot)


def test_single_rank_and_hook_and_errors():
    os.environ.setdefault("GM_NO_TORCH_PRELOAD", "1")
    from gemini_amd import capi, collective

    collective.finalize()
    assert collective.info() == (0, 1, "none")
    x = np.arange(18, dtype=np.uint64)
    assert (collective.allgather_host(x) == x[None]).all()  # no transport: a copy
    # a hook that plays three ranks (this process is rank 1 of 3)
    seen = []

    def gather(payload: bytes) -> bytes:
        seen.append(len(payload))
        return b"".join(bytes((b + r - 1) & 0xFF for b in payload) for r in range(3))

    collective.init_hook(1, 3, gather)
    assert collective.info() == (1, 3, "hook")
    got = collective.allgather_host(np.array([5, 6], dtype=np.uint64))
    assert got.shape == (3, 2) and (got[1] == [5, 6]).all() and seen == [16]
    with pytest.raises(capi.GeminiHipError):
        collective.selftest()  # the fake peers do not send the selftest's patterns: it must notice
    # a failing hook surfaces as an error code, not an exception through C frames
    collective.init_hook(0, 2, lambda payload: (_ for _ in ()).throw(RuntimeError("boom")))
    with pytest.raises(capi.GeminiHipError):
        collective.allgather_host(x)
    collective.finalize()
    lib = capi.load()
    import ctypes as C

    assert lib.gm_dist_init_shm(C.c_int(0), C.c_int(2), b"no-slash", C.c_size_t(0)) == -1
    assert lib.gm_dist_init_shm(C.c_int(2), C.c_int(2), b"/x", C.c_size_t(0)) == -1
    assert lib.gm_dist_init_hook(C.c_int(0), C.c_int(2), collective.ALLGATHER_FN(), None) == -1
    assert lib.gm_dist_init_rccl(C.c_int(0), C.c_int(1), (C.c_uint8 * 128)()) == -2  # needs gm_init: the communicator binds to its device

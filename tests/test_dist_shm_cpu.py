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


STALE_WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    os.environ["GM_NO_TORCH_PRELOAD"] = "1"
    from gemini_amd import collective
    rank, world, name, delay = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
    time.sleep(delay)
    collective.init_shm(rank, world, name, 256)
    for it in range(50):
        got = collective.allgather_host(np.array([rank * 7 + it, it], dtype=np.uint64))
        assert (got[:, 0] == np.arange(world) * 7 + it).all() and (got[:, 1] == it).all(), (rank, it, got)
    collective.finalize()
    print("ok", rank)
""") % ROOT


def _stale_segment(name, world, slot, seq_value):
    """What a crashed run leaves behind: a segment of the right size with a valid header (dist.cpp: ShmHeader = magic, world,
    slot_bytes, attached, seq[64], hello[64], ack[64]), counters at `seq_value`, slots full of old payload."""
    header = 8 * (4 + 3 * 64)
    total = ((header + 63) & ~63) + 2 * world * slot
    raw = np.zeros(total // 8, dtype=np.uint64)
    raw[0] = 0x474D44495354  # SHM_MAGIC
    raw[1], raw[2], raw[3] = world, slot, world
    raw[4:4 + 64] = seq_value
    raw[4 + 64:4 + 128] = 0xDEADBEEF  # old nonces, old echoes
    raw[4 + 128:4 + 192] = 0xDEADBEEF
    raw[(header + 63) // 64 * 8:] = 0x5A5A5A5A5A5A5A5A
    with open("/dev/shm" + name, "wb") as f:
        f.write(raw.tobytes())


@pytest.mark.parametrize("seq_value", [1, 2, 123456])
def test_a_stale_segment_of_a_crashed_run_is_never_trusted(seq_value):
    """bench.py reuses its segment name; a peer that opens the name BEFORE rank 0 has replaced the stale
    segment used to accept it (valid magic, world and slot size) and read old payload bytes as GM_OK while rank 0 waited for
    300 s.  Now a peer trusts a segment only once the rank 0 of THIS run has echoed the peer's fresh nonce, rank 0 poisons what it
    unlinks, and a counter that is not at the one or two values it can have is an error.  The peers start 1.5 s before rank 0."""
    world, slot = 3, 256
    name = f"/gm_stale_{os.getpid()}_{seq_value}"
    _stale_segment(name, world, slot, seq_value)
    try:
        procs = [subprocess.Popen([sys.executable, "-c", STALE_WORKER, str(r), str(world), name, "1.5" if r == 0 else "0"],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        outs = [p.communicate(timeout=120) for p in procs]
        for p, (o, e) in zip(procs, outs):
            assert p.returncode == 0, e[-3000:]
        assert not os.path.exists("/dev/shm" + name)
    finally:
        if os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)


CORRUPT_WORKER = textwrap.dedent("""
    import os, sys, time, mmap
    import numpy as np
    sys.path.insert(0, %r)
    os.environ["GM_NO_TORCH_PRELOAD"] = "1"
    os.environ["GM_DIST_TIMEOUT_S"] = "20"
    from gemini_amd import capi, collective
    rank, name = int(sys.argv[1]), sys.argv[2]
    collective.init_shm(rank, 2, name, 256)
    collective.allgather_host(np.array([rank], dtype=np.uint64))
    if rank == 1:
        with open("/dev/shm" + name, "r+b") as f:  # a wild write over this rank's own call counter (seq[1])
            m = mmap.mmap(f.fileno(), 0)
            m[8 * 5:8 * 6] = (10**9).to_bytes(8, "little")
            m.close()
        time.sleep(1.0)
        print("ok 1")
    else:
        time.sleep(0.3)
        try:
            collective.allgather_host(np.array([rank], dtype=np.uint64))
        except capi.GeminiHipError as e:
            assert e.code == -6 and "stale or corrupted" in str(e), str(e)
            print("ok 0")
        else:
            raise SystemExit("a corrupted counter was accepted")
""") % ROOT


def test_a_corrupted_call_counter_is_an_error_not_data():
    name = f"/gm_corrupt_{os.getpid()}"
    try:
        procs = [subprocess.Popen([sys.executable, "-c", CORRUPT_WORKER, str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [p.communicate(timeout=120) for p in procs]
        for p, (o, e) in zip(procs, outs):
            assert p.returncode == 0 and o.startswith("ok"), (o, e[-3000:])
    finally:
        if os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)


def test_reinit_under_the_same_name_does_not_lose_the_new_segment():
    """A slow rank of the previous init must not unlink the segment a re-init created under the same name (shm_detach compares
    the inode the name resolves to with the one it mapped)."""
    os.environ.setdefault("GM_NO_TORCH_PRELOAD", "1")
    from gemini_amd import collective

    name = f"/gm_reinit_{os.getpid()}"
    collective.init_shm(0, 1, name, 64)
    ino = os.stat("/dev/shm" + name).st_ino
    # somebody replaces the name (a new run's rank 0) while this process still holds the old mapping
    os.unlink("/dev/shm" + name)
    with open("/dev/shm" + name, "wb") as f:
        f.write(b"\0" * 4096)
    assert os.stat("/dev/shm" + name).st_ino != ino
    collective.finalize()
    assert os.path.exists("/dev/shm" + name)  # not ours: left alone
    os.unlink("/dev/shm" + name)


ABORT_WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    os.environ["GM_NO_TORCH_PRELOAD"] = "1"
    from gemini_amd import capi, collective
    rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    collective.init_shm(rank, world, name, 256)
    got = collective.allgather_host(np.array([rank], dtype=np.uint64))
    assert (got[:, 0] == np.arange(world)).all()
    t0 = time.time()
    if rank == 1:
        time.sleep(0.3)                      # the others are already waiting for this rank
        capi.check(capi.load().gm_dist_abort())
    else:
        try:
            collective.allgather_host(np.array([rank], dtype=np.uint64))
            raise SystemExit("an all-gather completed without rank 1")
        except RuntimeError as e:
            assert "aborted the run" in str(e), str(e)
        assert time.time() - t0 < 20         # not the 300 s of the segment's timeout
        capi.check(capi.load().gm_dist_abort())   # (what a failing prover does; idempotent)
    assert collective.info()[1:] == (world, "failed")
    try:
        collective.allgather_host(np.array([rank], dtype=np.uint64))
        raise SystemExit("a collective went through a failed transport")
    except RuntimeError as e:
        assert "earlier collective" in str(e), str(e)
    collective.finalize()
    assert collective.info() == (0, 1, "none")
    print("ok", rank)
""") % ROOT


@pytest.mark.parametrize("world", [2, 5])
def test_abort_releases_the_waiting_ranks(world):
    """gm_dist_abort (round 6): the rank that fails outside a collective raises a flag in the segment; every rank waiting for it returns GM_ESTATE within
    moments instead of after the 300 s timeout, and the transport of every rank refuses further collectives until it is initialised again"""
    name = f"/gm_test_abort_{os.getpid()}_{world}"
    procs = [subprocess.Popen([sys.executable, "-c", ABORT_WORKER, str(r), str(world), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and o.strip().startswith("ok"), e[-3000:]


def test_psnark_shard_layout_functions():
    """gm_psnark_shard_block / gm_psnark_shard_level (pure functions of the lengths, no device): the block is a multiple of 4 within 1/64 of
    ceil(longest / world); a family's level is the finest one whose `world` blocks still hold it, the blocks of every level up to it halve exactly"""
    import ctypes as C

    os.environ.setdefault("GM_NO_TORCH_PRELOAD", "1")
    from gemini_amd import capi

    lib = capi.load()
    lib.gm_psnark_shard_block.restype = C.c_size_t
    lib.gm_psnark_shard_level.restype = C.c_size_t
    rng = np.random.default_rng(11)
    for _ in range(400):
        world = int(rng.integers(1, 17))
        longest = int(rng.integers(8, 1 << int(rng.integers(4, 30))))
        tail_log = int(rng.integers(2, 12))
        block = int(lib.gm_psnark_shard_block(C.c_size_t(longest), C.c_int(world)))
        per = -(-longest // world)
        assert block % 4 == 0 and block >= per and block * world >= longest
        assert block - per <= max(per // 32, 8), (longest, world, block)
        for length in (longest, longest // 2 + 1, longest // 7 + 1, 1):
            s = int(lib.gm_psnark_shard_level(C.c_size_t(length), C.c_size_t(block), C.c_size_t(tail_log), C.c_int(world)))
            assert world * (block >> s) >= length                                  # the family fits its blocks
            for j in range(s + 1):
                assert (block >> j) << j == block and (block >> j) % 2 == 0        # every level up to it halves exactly and stays even
            if s > 0:
                assert (block >> s) >= (1 << tail_log)                             # sharded levels are never shorter than the tail
            # maximal: one level finer would not hold the family, or is not a sharded level any more
            nxt = block >> (s + 1)
            finer_ok = world * nxt >= length and (block >> s) % 2 == 0 and nxt % 2 == 0 and nxt >= (1 << tail_log)
            assert not finer_ok, (length, block, s, world, tail_log)

"""The N-GPU provers COMPILED INTO THE LIBRARY (gemini_amd/csrc/sharded.cpp) over its own collective layer
(gemini_amd/csrc/dist.cpp): N processes sharing the one GPU of the test box must produce the single-GPU proof byte for byte.

  * transport shm  : no torch.distributed anywhere -- what a Rust / C++ embedder gets from the C ABI alone
  * transport hook : the same calls over torch.distributed (gloo) behind gm_dist_init_hook
  * transport rccl : ncclAllGather on the library's own communicator.  RCCL refuses two ranks on one device, so here it runs
                     with ONE rank (binding, staging, stream order) -- and gm_dist_selftest() runs it whenever a box has more.
Reference: src/snark/time_prover.rs:19-117, src/subprotocols/sumcheck/proof.rs:36-66, src/kzg/time.rs:81-107,
src/misc.rs:100-110 (general matrices), examples/psnark.rs:54-81."""
import json
import os
import socket
import subprocess
import sys

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


def _run(world, extra, tool="run_snark.py", logn=12, transport="shm", env_extra=None):
    env = dict(os.environ, GM_BENCH_BACKEND="gloo", GM_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    script = [os.path.join(ROOT, "tools", tool), "-i", str(logn), "--repeat", "1"] + list(extra)
    if world == 1 and transport is None:
        cmd = [sys.executable] + script
    else:
        script += ["--transport", transport]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port())] + script
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


_ONE = {}


def _single(extra=(), tool="run_snark.py", logn=12):
    key = (tuple(extra), tool, logn)
    if key not in _ONE:
        _ONE[key] = _run(1, extra, tool, logn, transport=None)
    return _ONE[key]


def test_rccl_binding_with_one_rank():
    """ncclAllGather through the library's own communicator: host payloads (pinned -> device -> all-gather -> pinned, one wait)
    and a device vector, plus the self-test that opens its own one-rank communicator"""
    import ctypes as C

    import gemini_amd as gm
    from gemini_amd import collective
    from gemini_amd.fr import FrVec

    # ONE node, one rank: RCCL needs no network transport here.  Once in ~10 suite runs on the pool this test took 286 s instead of 7-9 s (three
    # communicator creations stalling in RCCL's own bootstrap on that box); keep its bootstrap on the loopback and its InfiniBand probe off
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    gm.capi.init()
    collective.finalize()
    collective.selftest()  # no transport: a temporary one-rank RCCL communicator
    assert collective.info() == (0, 1, "none")
    uid = np.zeros(128, dtype=np.uint8)
    gm.capi.check(gm.capi.load().gm_dist_rccl_unique_id(uid.ctypes.data_as(C.POINTER(C.c_uint8))))
    assert uid.any()
    gm.capi.check(gm.capi.load().gm_dist_init_rccl(C.c_int(0), C.c_int(1), uid.ctypes.data_as(C.POINTER(C.c_uint8))))
    try:
        assert collective.info() == (0, 1, "rccl")
        collective.selftest()
        x = np.arange(18 * 5, dtype=np.uint64).reshape(5, 18) * np.uint64(0x9E3779B97F4A7C15)
        assert (collective.allgather_host(x) == x[None]).all()
        rng = np.random.default_rng(3)
        host = rng.integers(0, 2**62, size=(3000, 4), dtype=np.uint64)
        v = FrVec.from_host(host)
        out = collective.allgather_vec(v)
        assert len(out) == 3000 and (out.to_host() == host).all()
        st = collective.stats()
        assert st["collectives"] >= 8 and st["bytes_received"] > 3000 * 32
        v.free()
        out.free()
    finally:
        collective.finalize()
    # one node, no out-of-band channel: the unique id travels through a shared-memory segment (gm_dist_init_rccl_node)
    collective.init_rccl_node(0, 1, f"/gm_test_rcclnode_{os.getpid()}")
    try:
        assert collective.info() == (0, 1, "rccl")
        collective.selftest()
        # the rendezvous segment stays open as the side channel: field values cross it, partial G1 points take ncclAllGather
        collective.stats(reset=True)
        x = np.arange(8, dtype=np.uint64)
        assert (collective.allgather_host(x) == x[None]).all()
        assert (collective.allgather_host(x, collective.CLASS_FIELD) == x[None]).all()
        pts = np.arange(18 * 3, dtype=np.uint64).reshape(3, 18)
        assert (collective.allgather_host(pts, collective.CLASS_G1) == pts[None]).all()
        routes = collective.stats_routes()
        assert routes["shm"]["collectives"] == 2 and routes["rccl_host_staged"]["collectives"] == 1, routes
        # both routes carry either class when forced (gm_dist_bench: what profiles/r5_collective_latency.txt was measured with)
        for route in ("shm", "rccl_host_staged"):
            assert collective.bench(144, 20, collective.CLASS_G1, route) > 0
        # re-blocking with one rank goes through the grouped ncclSend / ncclRecv branch (no peers: the self part is a device copy)
        host = np.random.default_rng(5).integers(0, 2**62, size=(100, 4), dtype=np.uint64)
        v = FrVec.from_host(host)
        (o64,), (o128,) = collective.reblock_vecs([v], 64), collective.reblock_vecs([v], 128)
        assert len(o64) == 64 and (o64.to_host() == host[:64]).all() and len(o128) == 100 and (o128.to_host() == host).all()
        for x in (v, o64, o128):
            x.free()
    finally:
        collective.finalize()
    assert not os.path.exists(f"/dev/shm/gm_test_rcclnode_{os.getpid()}")


@pytest.mark.parametrize("extra", [[], ["--elastic"]], ids=["time", "elastic"])
def test_cyclic_key_native_provers_same_proof(extra):
    """gm_snark_new_time / gm_snark_new_elastic handed a CYCLIC SHARE of the key: 2 and 3 ranks == 1 GPU"""
    one = _single(extra)
    for world, transport in (((3, "shm"),) if not extra else ((2, "hook"),)):  # (more combinations: tests/soak_dist_native.py)
        many = _run(world, extra, transport=transport)
        assert many["n_gpus"] == world and many["transport"] == transport
        assert many["proof_sha256"] == one["proof_sha256"], (world, transport, extra)
        assert many["collectives"]["collectives"] > 10


@pytest.mark.parametrize("extra", [[], ["--elastic"]], ids=["time", "elastic"])
def test_cyclic_key_native_psnark_same_proof(extra):
    """BASELINE configs[4] (`psnark`, 8 GPUs): gm_psnark_new_time (and the elastic prover) over cyclic shares on 2 and 3 ranks"""
    one = _single(extra, tool="run_psnark.py", logn=10)
    for world in ((3,) if not extra else (2,)):
        many = _run(world, extra, tool="run_psnark.py", logn=10)
        assert many["n_gpus"] == world and many["proof_sha256"] == one["proof_sha256"], (world, extra)


@pytest.mark.parametrize("tail_log", [4, 6, 8])
def test_block_sharded_native_prover_same_proof(tail_log):
    """gm_snark_new_time_sharded, block-diagonal instance (local columns): 1 / 2 / 4 / 8 ranks == gm_snark_new_time.  tail_log = 8 on
    8 ranks (blocks of 512): the first gathered level is TWO blocks long -- every rank takes its range of the replicated levels in
    the n / g opening, not rank 0 all of them (found by tests/soak_dist_native.py, which sweeps transports x worlds x tails)"""
    one = _single()
    for world in {4: (2,), 6: (4,), 8: (8,)}[tail_log]:
        many = _run(world, ["--block-sharded", "--tail-log", str(tail_log)], transport="hook" if (tail_log, world) == (4, 2) else "shm")
        assert many["proof_sha256"] == one["proof_sha256"], (world, tail_log)


def test_block_sharded_native_prover_general_matrices():
    """any Matrix<F> (src/misc.rs:100-110): a random satisfied R1CS with entries in arbitrary columns, row blocks with global column
    indices, on 2 and 4 ranks == the single-GPU prover on the same instance; and dummy_r1cs posed as a general matrix"""
    one = _single(["--random-r1cs", "77"], logn=10)
    assert one["proof_sha256"] != _single(logn=10)["proof_sha256"]
    for world in (4,):  # (2 ranks and more seeds: tests/soak_dist_native.py)
        many = _run(world, ["--random-r1cs", "77", "--block-sharded", "--tail-log", "5"], logn=10)
        assert many["proof_sha256"] == one["proof_sha256"], world
    dummy = _single()
    for world in (4,):
        many = _run(world, ["--block-sharded", "--global-columns", "--tail-log", "6"])
        assert many["proof_sha256"] == dummy["proof_sha256"], world


REBLOCK_WORKER = """
import os, sys
import numpy as np
sys.path.insert(0, %r)
import gemini_amd as gm
from gemini_amd import collective
from gemini_amd.fr import FrVec
rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
gm.capi.init(0)
collective.init_shm(rank, world, name, 4096)
m = 64
glob = lambda j, total: (np.arange(total, dtype=np.uint64)[:, None] * np.uint64(1000003) + np.uint64(7919 * j) + np.arange(4, dtype=np.uint64)[None, :])
# vector j is sharded in blocks of m >> j (the levels of a folding tree); re-block to blocks of m
locs, tots = [], []
for j in range(0, 7):
    b = m >> j
    g = glob(j, world * b)
    locs.append(FrVec.from_host(np.ascontiguousarray(g[rank * b:(rank + 1) * b])))
    tots.append(world * b)
outs = collective.reblock_vecs(locs, m)
for j, (o, total) in enumerate(zip(outs, tots)):
    lo = rank * m
    want = glob(j, total)[lo:lo + m] if lo < total else np.zeros((0, 4), dtype=np.uint64)
    assert len(o) == len(want), (rank, j, len(o), len(want))
    if len(want):
        assert (o.to_host() == want).all(), (rank, j)
st = collective.stats_routes()
assert "shm" in st
collective.finalize()
print("ok", rank)
""" % ROOT


@pytest.mark.parametrize("world", [2, 8])
def test_reblocking_between_ranks_on_the_shared_gpu(world):
    """gm_dist_reblock_vecs over the host transport (N processes share the one GPU): the levels of a folding tree, sharded in
    blocks of m / 2^j, arrive as the blocks [r m, (r + 1) m) of the same global vectors -- what the n / g opening of the
    block-sharded prover needs (src/kzg/time.rs:149-159 opens ONE polynomial; rank r commits ITS coefficient range of its quotient)"""
    name = f"/gm_reblock_{os.getpid()}_{world}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", REBLOCK_WORKER, str(r), str(world), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and o.strip().startswith("ok"), e[-3000:]


def test_sharded_sumcheck_refuses_blocks_that_do_not_tile():
    """gm_sumcheck_prove_sharded is a public entry point; blocks that are not equal, in rank order and covering n_global used to
    deadlock an all-gather or produce messages that differ from the single-GPU prover's.  Now GM_EINVAL before anything is launched."""
    import ctypes as C

    import gemini_amd as gm
    from gemini_amd.fr import FrVec

    gm.capi.init()
    lib = gm.capi.load()
    f, g = FrVec.from_host(np.ones((64, 4), dtype=np.uint64)), FrVec.from_host(np.ones((32, 4), dtype=np.uint64))
    tr = C.c_uint64()
    gm.capi.check(lib.gm_transcript_new(b"x", C.c_size_t(1), C.byref(tr)))
    msgs, chal, fin, rounds = np.zeros(64 * 8, np.uint64), np.zeros(64 * 4, np.uint64), np.zeros(8, np.uint64), C.c_size_t()
    tw = np.ones(4, dtype=np.uint64)
    call = lambda fv, gv, lo, n: lib.gm_sumcheck_prove_sharded(tr, C.c_uint64(fv.handle), C.c_uint64(gv.handle), gm.capi.ptr(tw), C.c_size_t(lo), C.c_size_t(n),
                                                               gm.capi.ptr(msgs), gm.capi.ptr(chal), C.c_size_t(64), gm.capi.ptr(fin), C.byref(rounds))
    assert call(f, g, 0, 64) == -1 and b"blocks of f and g" in lib.gm_last_error()  # unequal blocks
    assert call(f, f, 2, 64) == -1  # one rank, but not the whole vector
    assert call(f, f, 0, 128) == -1
    assert call(f, f, 0, 64) == 0 and rounds.value == 6
    gm.capi.check(lib.gm_transcript_free(tr))
    f.free()
    g.free()


# ---- psnark with the FIELD side block-sharded (gm_psnark_new_time_sharded, gemini_amd/csrc/psnark_sharded.cpp) ---------------------
@pytest.mark.parametrize("world,tail_log,transport", [(1, 4, "shm"), (3, 5, "hook"), (8, 4, "shm")])  # (2, 4, 5, 6, 7, 9 ranks: tests/soak_dist_native.py)
def test_block_sharded_psnark_same_proof(world, tail_log, transport):
    """BASELINE configs[4]: psnark::Proof::new_time (src/psnark/time_prover.rs:69-384) with every vector in blocks over 1 / 2 / 3 / 4 / 8 ranks
    sharing the test GPU == gm_psnark_new_time byte for byte (dummy_r1cs, src/psnark/tests.rs:14-55): lookups from replicated sources, the suffix
    products with a carry between ranks, the 13-prover batch sumcheck with one all-gather per round, the n / g openings"""
    one = _single(tool="run_psnark.py", logn=10)
    many = _run(world, ["--block-sharded", "--tail-log", str(tail_log)], tool="run_psnark.py", logn=10, transport=transport)
    assert many["n_gpus"] == world and many["proof_sha256"] == one["proof_sha256"], (world, tail_log)
    if world == 8:  # the layout really has LEVELS here: the ~n-long families in blocks half the size of the ~2n-long ones
        levels = many["layout"]["levels_by_family_length"]
        assert max(levels.values()) >= 1 and min(levels.values()) == 0, levels


@pytest.mark.parametrize("world,tail_log", [(4, 3)])
def test_block_sharded_psnark_general_matrices(world, tail_log):
    """the same on a random satisfied R1CS (entries of A and B in arbitrary columns, src/psnark/tests.rs:57-125 random circuits): the joint support is
    irregular, the extended frequencies repeat indices, the row blocks read z everywhere"""
    one = _single(["--random-r1cs", "91"], tool="run_psnark.py", logn=9)
    assert one["proof_sha256"] != _single(tool="run_psnark.py", logn=9)["proof_sha256"]
    many = _run(world, ["--random-r1cs", "91", "--block-sharded", "--tail-log", str(tail_log)], tool="run_psnark.py", logn=9)
    assert many["proof_sha256"] == one["proof_sha256"], (world, tail_log)


def test_block_sharded_psnark_elastic_and_verifiable_key():
    """the elastic prover's resident schedule (src/psnark/elastic_prover.rs:60-634: time provers on the same vectors) is the same sharded entry -- against
    the single-GPU ELASTIC prover on its 3 n key; and the time prover on a key one power longer than the example's (nothing truncated)"""
    one = _single(["--elastic"], tool="run_psnark.py", logn=10)
    many = _run(4, ["--elastic", "--block-sharded", "--tail-log", "5"], tool="run_psnark.py", logn=10)
    assert many["proof_sha256"] == one["proof_sha256"]
    # (the key one power longer than the example's, nothing truncated: tests/soak_dist_native.py sweeps it)


# ---- the N-rank RCCL branches of dist.cpp, through a TEST-ONLY stand-in for librccl (tests/fake_rccl) --------------------------------
@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    """RCCL refuses two ranks on one device and no multi-GPU node is available: tests/fake_rccl/fake_rccl.cpp implements the entry points dist.cpp
    binds (all-gather, grouped send / recv, abort) over shared memory and host-staged copies for processes sharing ONE GPU; GM_RCCL_LIB selects it"""
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-ldl", "-lrt", "-lpthread"])
    return {"GM_RCCL_LIB": so, "GM_FAKE_RCCL_TIMEOUT_S": "120"}


@pytest.mark.parametrize("world,transport", [(2, "rccl"), (8, "rccl-node")])
def test_rccl_branches_with_n_ranks_snark(fake_rccl, world, transport):
    """gm_snark_new_time_sharded over the RCCL transport with 2 and 8 ranks: ncclAllGather of host payloads (staged) and of device vectors (the gathered
    level), the grouped ncclSend / ncclRecv of gm_dist_reblock_vecs; with rccl-node the field payloads take the side segment and the G1 points RCCL"""
    one = _single()
    many = _run(world, ["--block-sharded", "--tail-log", "5"], transport=transport, env_extra=fake_rccl)
    assert many["transport"] == "rccl" and many["proof_sha256"] == one["proof_sha256"], (world, transport)


@pytest.mark.parametrize("world,transport", [(2, "rccl-node"), (4, "rccl")])  # (8 ranks over the stand-in: the snark test above and tools/r6_final.sh n2)
def test_rccl_branches_with_n_ranks_psnark(fake_rccl, world, transport):
    """gm_psnark_new_time_sharded the same way (BASELINE configs[4] over the transport the driver's 8-GPU run will use)"""
    logn = 10 if world == 2 else 8  # (the stand-in stages every payload through the host: 8 ranks of it are slow)
    one = _single(tool="run_psnark.py", logn=logn)
    many = _run(world, ["--block-sharded", "--tail-log", "4"], tool="run_psnark.py", logn=logn, transport=transport, env_extra=fake_rccl)
    assert many["transport"] == "rccl" and many["proof_sha256"] == one["proof_sha256"], (world, transport)


RCCL_FAIL_WORKER = """
import os, sys
import numpy as np
sys.path.insert(0, %r)
import gemini_amd as gm
from gemini_amd import collective
from gemini_amd.fr import FrVec
rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
gm.capi.init(0)
collective.init_rccl_node(rank, world, name)
pts = np.arange(18, dtype=np.uint64) + rank
ok = collective.allgather_host(pts, collective.CLASS_G1)        # collective 1 of the communicator: fine
assert (ok[:, 0] == np.arange(world)).all()
codes = []
for attempt in range(3):                                        # collective 2: rank 1 fails inside it (GM_FAKE_RCCL_FAIL_AT=1:2) and aborts the communicator
    try:
        collective.allgather_host(pts, collective.CLASS_G1)
        codes.append("ok")
    except Exception as e:
        codes.append(str(e))
assert codes[0] != "ok", codes                                   # every rank returns with an error instead of hanging ...
assert all("earlier collective" in c for c in codes[1:]), codes  # ... and the transport stays refused (not a one-rank copy) until it is initialised again
assert collective.info()[1:] == (world, "failed")
v = FrVec.from_host(np.ones((8, 4), dtype=np.uint64))
try:
    collective.allgather_vec(v)
    raise SystemExit("a device all-gather went through a failed transport")
except RuntimeError:
    pass
collective.finalize()
assert collective.info() == (0, 1, "none")
print("ok", rank)
""" % ROOT


def test_rccl_failing_rank_aborts_and_poisons_the_transport(fake_rccl):
    """after a failed RCCL collective the transport used to fall back to T_NONE with world = N -- later all-gathers "succeeded" as
    copies of one payload.  Now: the failing rank aborts the communicator, its peer returns with an error, and every later collective is GM_ESTATE"""
    name = f"/gm_rcclfail_{os.getpid()}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GM_FAKE_RCCL_FAIL_AT="1:2", **fake_rccl)
    env["GM_FAKE_RCCL_TIMEOUT_S"] = "20"
    procs = [subprocess.Popen([sys.executable, "-c", RCCL_FAIL_WORKER, str(r), "2", name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and o.strip().startswith("ok"), e[-3000:]


@pytest.mark.parametrize("world", [4])  # (1 / 2 / 8: tests/soak_dist_native.py and tools/r6_final.sh)
def test_block_sharded_elastic_snark_dummy_srs(world):
    """BASELINE configs[3] as written (examples/snark.rs:54-66: the ELASTIC prover on the DummyStreamer key, 8 GPUs): gm_snark_new_elastic_sharded --
    the resident schedule of the elastic prover over blocks, the generator-copies key in slices -- == the single-GPU elastic prover on the same key"""
    one = _single(["--dummy-srs", "--elastic"])
    assert one["proof_sha256"] != _single(["--elastic"])["proof_sha256"]
    many = _run(world, ["--dummy-srs", "--elastic", "--block-sharded", "--tail-log", "5"])
    assert many["proof_sha256"] == one["proof_sha256"] and "elastic_prover_s" in many, world


ABORT_WORKER = """
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
import gemini_amd as gm
from gemini_amd import collective
from gemini_amd.sharded import PsnarkShard, PsnarkShardKey, psnark_new_time_sharded
rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
gm.capi.init(0)
collective.init_shm(rank, world, name)
n = 1 << 9
shard = PsnarkShard.dummy(12345, n, tail_log=4)
key = PsnarkShardKey(2 * n, shard.block, 4, np.array([7, 0, 0, 0], dtype=np.uint64))
index = shard.index(key)
good = psnark_new_time_sharded(shard, key, index)           # a proof goes through
if rank == 1:
    shard.w_len += 1                                         # this rank's input no longer tiles: GM_EINVAL before its first collective
t0 = time.time()
try:
    psnark_new_time_sharded(shard, key, index)
    raise SystemExit("the broken proof went through on rank %%d" %% rank)
except RuntimeError as e:
    msg = str(e)
dt = time.time() - t0
assert dt < 30, dt                                           # the healthy rank does not wait out the 300 s of the segment
assert ("input block" in msg) if rank == 1 else ("aborted the run" in msg), (rank, msg)
assert collective.info()[2] == "failed"
collective.finalize()
print("ok", rank, round(dt, 2))
""" % ROOT


def test_a_prover_failing_on_one_rank_releases_its_peers():
    """gm_dist_abort: a sharded prover that fails on one rank OUTSIDE a collective (here: an input block that does not tile, refused before the first
    all-gather) raises a flag in the node's segment -- its peer, already waiting in the first commitment's all-gather, returns GM_ESTATE within
    seconds instead of after the segment's 300 s timeout, and both transports refuse further collectives until they are initialised again"""
    name = f"/gm_abort_{os.getpid()}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", ABORT_WORKER, str(r), "2", name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and o.strip().startswith("ok"), e[-3000:]


def test_elastic_sharded_entry_refuses_the_literal_schedule():
    """gm_snark_new_elastic_sharded runs the RESIDENT schedule over blocks; min_device_chunk = 1 selects the literal elastic prover (space provers over
    whole streams, src/snark/elastic_prover.rs:174-266 as written), which is single-GPU: refused with GM_EINVAL and a message, before anything is launched"""
    import ctypes as C

    import gemini_amd as gm
    from gemini_amd import collective
    from gemini_amd.sharded import R1csShard, ShardKey, new_time_sharded

    gm.capi.init()
    collective.finalize()
    n = 1 << 8
    shard = R1csShard.dummy(4242, n)
    key = ShardKey(n, 4, np.array([1, 0, 0, 0], dtype=np.uint64))  # the DummyStreamer key: copies of the generator = powers of tau = 1
    resident = new_time_sharded(shard, key, elastic=(1 << 20, 1 << 26))
    assert resident.serialize_compressed() == new_time_sharded(shard, key).serialize_compressed()  # the same proof as the sharded time prover
    with pytest.raises(RuntimeError, match="LITERAL"):
        new_time_sharded(shard, key, elastic=(1 << 20, 1))
    shard.free()
    key.free()

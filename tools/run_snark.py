#!/usr/bin/env python3
"""examples/snark.rs --time-prover -i <logn> on the device path: dummy_r1cs(2^logn), SRS of
2^(logn+1)+1 powers (examples/snark.rs:69-79), then Proof::new_time.  Prints the spans the reference
prints with print-trace (src/snark/time_prover.rs:23,41,51,83,100)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_rows(n: int, seed: int):
    """(a_rows, b_rows, c_rows, z): rows of (value_mont, column) pairs of a satisfied instance, the same on every rank"""
    from gemini_amd.fr import R_MOD, fr_from_int

    rng = np.random.default_rng(seed)
    fr = lambda: int.from_bytes(rng.bytes(40), "little") % R_MOD  # noqa: E731
    z = [fr() or 1 for _ in range(n)]

    def mk():
        rows = []
        for _ in range(n):
            cols = rng.choice(n, size=int(rng.integers(1, 4)), replace=False)
            rows.append([(fr(), int(c)) for c in cols])
        return rows

    a, b = mk(), mk()
    mv = lambda rows: [sum(v * z[c] for v, c in row) % R_MOD for row in rows]  # noqa: E731
    za, zb = mv(a), mv(b)
    c = [[(za[i] * zb[i] % R_MOD * pow(z[i], -1, R_MOD) % R_MOD, i)] for i in range(n)]
    M = lambda rows: [[(fr_from_int(v), col) for v, col in row] for row in rows]  # noqa: E731
    return M(a), M(b), M(c), np.stack([fr_from_int(v) for v in z])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--instance-logsize", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--dummy-srs", action="store_true", help="powers_of_g = copies of the generator, the DummyStreamer key of "
                    "examples/snark.rs:59-63 (elastic main) instead of tau^i * g")
    ap.add_argument("--max-msm-buffer-log", type=int, default=20, help="max_msm_buffer of the elastic prover (examples/snark.rs:57: 2^20)")
    ap.add_argument("--tables", action="store_true", help="gm_g1_bases_precompute on the committer key before proving (13 x the key in HBM)")
    ap.add_argument("--min-device-chunk-log", type=int, default=None, help="CommitterKeyStream.min_device_chunk = 2^k (default: the class default)")
    ap.add_argument("--stepwise", action="store_true", help="the step-wise Python statement of the prover (tests/stepwise: the cross-check, one FFI call per step) instead of the native one")
    ap.add_argument("--native", action="store_true", help="the default since round 6 (kept for old command lines): the provers compiled into the library")
    ap.add_argument("--block-sharded", action="store_true", help="N ranks, field arithmetic sharded as well: every vector and the key in blocks "
                    "(tests/stepwise/dist_prover.py); the world size must be a power of two")
    ap.add_argument("--tail-log", type=int, default=10, help="--block-sharded: blocks shorter than 2^k elements are gathered")
    ap.add_argument("--transport", choices=["shm", "hook", "rccl", "rccl-node"], default=None, help="N ranks through the collective layer INSIDE the library "
                    "(gemini_amd/csrc/dist.cpp) and the provers compiled into it: shm = shared-memory segment (no torch.distributed at all), "
                    "hook = torch.distributed (gloo / nccl) behind gm_dist_init_hook, rccl = the library's own RCCL communicator.  Default sharding: "
                    "the element-cyclic key with the native prover (MSMs sharded); with --block-sharded: gm_snark_new_time_sharded")
    ap.add_argument("--global-columns", action="store_true", help="--block-sharded --transport: the instance as a GENERAL matrix (row blocks with "
                    "global column indices, z whole on every rank) instead of block-diagonal")
    ap.add_argument("--random-r1cs", type=int, default=None, metavar="SEED", help="a satisfied random GENERAL R1CS (1-3 entries per row of A and B in "
                    "random columns, C diagonal) instead of dummy_r1cs: single GPU, or --block-sharded --transport (row blocks, global columns)")
    ap.add_argument("--elastic", action="store_true", help="Proof::new_elastic over device-resident streams, max_msm_buffer = 2^20 "
                    "(examples/snark.rs elastic_snark_main) instead of --time-prover")
    args = ap.parse_args()
    import tests.stepwise  # noqa: F401 -- registers the step-wise cross-check (what --stepwise and the Python-level sharded keys use)
    import gemini_amd as gm
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof

    # N > 1 (launched by torch.distributed.run): the KZG key is sharded by powers across the ranks
    # (tests.stepwise.dist.ShardedCommitterKey), everything else is replicated.  GM_BENCH_BACKEND=gloo +
    # GM_BENCH_SINGLE_DEVICE=1 are the same test hooks as bench.py (N ranks on one GPU).
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    lib_dist = args.transport is not None
    if world > 1 and not (lib_dist and args.transport in ("shm", "rccl-node")):
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    gm.capi.init(local_rank)
    if lib_dist:
        from gemini_amd import collective

        if args.transport == "shm":
            collective.init_shm(rank, world, "/gm_run_snark_%s" % os.environ.get("MASTER_PORT", "0"))
        elif args.transport == "rccl-node":  # the library's own communicator, the id through a shm segment that stays open as the side channel (bench.py's path)
            collective.init_rccl_node(rank, world, "/gm_run_snark_node_%s" % os.environ.get("MASTER_PORT", "0"))
        elif args.transport == "hook":
            collective.init_hook_torch() if world > 1 else None
        else:
            collective.init_rccl_from_torch() if world > 1 else None
        collective.selftest() if world > 1 else None
    n = 1 << args.instance_logsize
    rng = np.random.default_rng(2022420)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD
    t0 = time.perf_counter()
    e_inst = rnd()
    if args.random_r1cs is not None:
        from gemini_amd.circuit import R1cs, SparseMatrix
        from gemini_amd.fr import FrVec

        ra, rb, rc, zh = random_rows(n, args.random_r1cs)
    if args.block_sharded and lib_dist:
        from gemini_amd.sharded import R1csShard, ShardKey, new_time_sharded

        r1cs = R1csShard.from_rows(ra, rb, rc, zh, 1) if args.random_r1cs is not None else R1csShard.dummy(e_inst, n, global_columns=args.global_columns)
    elif args.block_sharded:
        from tests.stepwise.dist_prover import BlockLayout, BlockShardedKey, R1csBlock

        layout = BlockLayout(n, rank, world, args.tail_log)
        r1cs = R1csBlock.dummy(e_inst, layout)
    elif args.random_r1cs is not None:
        mats = [SparseMatrix.from_rows(rows, n) for rows in (ra, rb, rc)] + [SparseMatrix.from_rows(rows, n, transpose=True) for rows in (ra, rb, rc)]
        r1cs = R1cs(*mats, FrVec.from_host(zh), FrVec.from_host(zh[1:]), FrVec.from_host(zh[:1]))
    else:
        r1cs = dummy_r1cs(e_inst, n)
    t_inst = time.perf_counter() - t0
    t0 = time.perf_counter()
    tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    if args.block_sharded and lib_dist:
        # --dummy-srs: the DummyStreamer key of examples/snark.rs:59-63 (copies of the generator) in slices = powers of tau = 1
        ck = ShardKey(n, args.tail_log, np.array([1, 0, 0, 0], dtype=np.uint64) if args.dummy_srs else tau)
    elif lib_dist:
        from gemini_amd.sharded import cyclic_committer_key

        ck = cyclic_committer_key(2 * n, 5, tau, with_g2=False)
    elif args.block_sharded:
        ck = BlockShardedKey.new(n, 5, tau, rank, world, args.tail_log)
    elif world > 1:
        from tests.stepwise.dist import ShardedCommitterKey

        ck = ShardedCommitterKey.new(2 * n, 5, tau, rank, world)
    else:
        if args.dummy_srs:
            from gemini_amd.kzg import g1_generator_mont
            from gemini_amd.msm import G1Bases

            ones = np.zeros((2 * n + 1, 4), dtype=np.uint64)
            ones[:, 0] = 1
            ck = CommitterKey(G1Bases.fixed_base(g1_generator_mont(), ones), 5)
        else:
            ck = CommitterKey.new(2 * n, 5, tau)
    t_srs = time.perf_counter() - t0
    t_tab = None
    if args.tables:
        t0 = time.perf_counter()
        ck.powers_of_g.precompute(0)
        t_tab = time.perf_counter() - t0
    out = {"n_gpus": world, "logn": args.instance_logsize, "instance_s": round(t_inst, 3), "srs_s": round(t_srs, 3), "tables_s": None if t_tab is None else round(t_tab, 3), "runs": []}
    stamps = []  # per proof: clock readings around it, for tools/exposed_time.py --stamps (rocprofv3 timestamps are one of these clocks)
    clocks = lambda: {"boottime_ns": time.clock_gettime_ns(time.CLOCK_BOOTTIME), "monotonic_ns": time.clock_gettime_ns(time.CLOCK_MONOTONIC),
                      "realtime_ns": time.clock_gettime_ns(time.CLOCK_REALTIME)}
    gm.capi.mem_reset_peak()  # the high-water marks below are those of the proofs, not of the key / index setup
    for _ in range(args.repeat):
        if lib_dist:
            collective.allgather_host(np.zeros(1, dtype=np.uint64))  # a barrier through the library's own transport
        elif world > 1:
            dist.barrier()
        stamps.append({"t0": clocks()})
        if args.elastic and args.block_sharded and lib_dist:
            # BASELINE configs[3]: the elastic prover's resident schedule over blocks (gm_snark_new_elastic_sharded)
            proof = new_time_sharded(r1cs, ck, elastic=(1 << args.max_msm_buffer_log, 1 << (26 if args.min_device_chunk_log is None else args.min_device_chunk_log)))
        elif args.elastic:
            from gemini_amd.circuit import R1csStream
            from gemini_amd.kzg import CommitterKeyStream

            stream = R1csStream(r1cs)
            if world > 1 and not lib_dist:
                from tests.stepwise.dist import ShardedCommitterKeyStream

                cks = ShardedCommitterKeyStream.from_sharded_key(ck)
            else:
                cks = CommitterKeyStream.from_committer_key(ck, min_device_chunk=None if args.min_device_chunk_log is None else 1 << args.min_device_chunk_log)
            proof = Proof.new_elastic(stream, cks, 1 << args.max_msm_buffer_log, native=not args.stepwise)
            stream.free()
        elif args.block_sharded and lib_dist:
            proof = new_time_sharded(r1cs, ck)
        elif args.block_sharded:
            from tests.stepwise.dist_prover import new_time_block_sharded

            proof = new_time_block_sharded(r1cs, ck)
        else:
            proof = Proof.new_time(r1cs, ck, native=not args.stepwise)
        stamps[-1]["t1"] = clocks()
        out["runs"].append({k: round(v, 4) for k, v in proof.spans.items()})
        if getattr(proof, "fr_work", None):
            out["fr_work"] = proof.fr_work  # field elements this rank's device passes read + wrote (tests/stepwise/dist_prover.py)
        out["proof_size_B"] = proof.compressed_size()  # examples/snark.rs:96 "proof-size {}B"
    key = "ark_gemini::snark::elastic_prover" if args.elastic else "ark_gemini::snark::time_prover"
    out["elastic_prover_s" if args.elastic else "time_prover_s"] = min(r[key] for r in out["runs"])
    if lib_dist:
        import hashlib

        tk = "elastic_prover_s" if args.elastic else "time_prover_s"
        mine = np.frombuffer(hashlib.sha256(proof.serialize_compressed()).digest() + np.float64(out[tk]).tobytes(), dtype=np.uint64)
        allr = collective.allgather_host(mine)
        assert (allr[:, :4] == allr[0, :4]).all(), "ranks produced different proofs"
        out[tk] = float(allr[:, 4].view(np.float64).max())  # the slowest rank
        out["transport"] = collective.info()[2]
        out["collectives"] = collective.stats()
        collective.finalize()
        if world > 1 and args.transport not in ("shm", "rccl-node"):
            dist.destroy_process_group()
    elif world > 1:
        import hashlib

        digest = hashlib.sha256(proof.serialize_compressed()).hexdigest()
        allt = [None] * world
        dist.all_gather_object(allt, (out.get("time_prover_s", out.get("elastic_prover_s")), digest))
        out["time_prover_s"] = max(t for t, _ in allt)  # the slowest rank
        assert len({d for _, d in allt}) == 1, "ranks produced different proofs"
        dist.destroy_process_group()
    out["proof_sha256"] = __import__("hashlib").sha256(proof.serialize_compressed()).hexdigest()
    out["mem_GB"] = {k: round(v / 1e9, 3) for k, v in gm.capi.mem_stats().items() if k != "spare_table_releases"}
    out["spare_table_releases"] = gm.capi.mem_stats()["spare_table_releases"]
    out["stamps"] = stamps
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

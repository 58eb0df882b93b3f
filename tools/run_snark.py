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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--instance-logsize", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--dummy-srs", action="store_true", help="powers_of_g = copies of the generator, the DummyStreamer key of "
                    "examples/snark.rs:59-63 (elastic main) instead of tau^i * g")
    ap.add_argument("--max-msm-buffer-log", type=int, default=20, help="max_msm_buffer of the elastic prover (examples/snark.rs:57: 2^20)")
    ap.add_argument("--tables", action="store_true", help="gm_g1_bases_precompute on the committer key before proving (13 x the key in HBM)")
    ap.add_argument("--min-device-chunk-log", type=int, default=None, help="CommitterKeyStream.min_device_chunk = 2^k (default: the class default)")
    ap.add_argument("--native", action="store_true", help="gm_snark_new_time / gm_snark_new_elastic: the prover's orchestration compiled into the library (one call per proof)")
    ap.add_argument("--block-sharded", action="store_true", help="N ranks, field arithmetic sharded as well: every vector and the key in blocks "
                    "(gemini_amd/dist_prover.py); the world size must be a power of two")
    ap.add_argument("--tail-log", type=int, default=10, help="--block-sharded: blocks shorter than 2^k elements are gathered")
    ap.add_argument("--elastic", action="store_true", help="Proof::new_elastic over device-resident streams, max_msm_buffer = 2^20 "
                    "(examples/snark.rs elastic_snark_main) instead of --time-prover")
    args = ap.parse_args()
    import gemini_amd as gm
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.snark import Proof

    # N > 1 (launched by torch.distributed.run): the KZG key is sharded by powers across the ranks
    # (gemini_amd.dist.ShardedCommitterKey), everything else is replicated.  GM_BENCH_BACKEND=gloo +
    # GM_BENCH_SINGLE_DEVICE=1 are the same test hooks as bench.py (N ranks on one GPU).
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    gm.capi.init(local_rank)
    n = 1 << args.instance_logsize
    rng = np.random.default_rng(2022420)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD
    t0 = time.perf_counter()
    e_inst = rnd()
    if args.block_sharded:
        from gemini_amd.dist_prover import BlockLayout, BlockShardedKey, R1csBlock

        layout = BlockLayout(n, rank, world, args.tail_log)
        r1cs = R1csBlock.dummy(e_inst, layout)
    else:
        r1cs = dummy_r1cs(e_inst, n)
    t_inst = time.perf_counter() - t0
    t0 = time.perf_counter()
    tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    if args.block_sharded:
        ck = BlockShardedKey.new(n, 5, tau, rank, world, args.tail_log)
    elif world > 1:
        from gemini_amd.dist import ShardedCommitterKey

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
    for _ in range(args.repeat):
        if world > 1:
            dist.barrier()
        stamps.append({"t0": clocks()})
        if args.elastic:
            from gemini_amd.circuit import R1csStream
            from gemini_amd.kzg import CommitterKeyStream

            stream = R1csStream(r1cs)
            if world > 1:
                from gemini_amd.dist import ShardedCommitterKeyStream

                cks = ShardedCommitterKeyStream.from_sharded_key(ck)
            else:
                cks = CommitterKeyStream.from_committer_key(ck, min_device_chunk=None if args.min_device_chunk_log is None else 1 << args.min_device_chunk_log)
            proof = Proof.new_elastic(stream, cks, 1 << args.max_msm_buffer_log, native=args.native)
            stream.free()
        elif args.block_sharded:
            from gemini_amd.dist_prover import new_time_block_sharded

            proof = new_time_block_sharded(r1cs, ck)
        else:
            proof = Proof.new_time(r1cs, ck, native=args.native)
        stamps[-1]["t1"] = clocks()
        out["runs"].append({k: round(v, 4) for k, v in proof.spans.items()})
        if getattr(proof, "fr_work", None):
            out["fr_work"] = proof.fr_work  # field elements this rank's device passes read + wrote (gemini_amd/dist_prover.py)
        out["proof_size_B"] = proof.compressed_size()  # examples/snark.rs:96 "proof-size {}B"
    key = "ark_gemini::snark::elastic_prover" if args.elastic else "ark_gemini::snark::time_prover"
    out["elastic_prover_s" if args.elastic else "time_prover_s"] = min(r[key] for r in out["runs"])
    if world > 1:
        import hashlib

        digest = hashlib.sha256(proof.serialize_compressed()).hexdigest()
        allt = [None] * world
        dist.all_gather_object(allt, (out.get("time_prover_s", out.get("elastic_prover_s")), digest))
        out["time_prover_s"] = max(t for t, _ in allt)  # the slowest rank
        assert len({d for _, d in allt}) == 1, "ranks produced different proofs"
        dist.destroy_process_group()
    out["proof_sha256"] = __import__("hashlib").sha256(proof.serialize_compressed()).hexdigest()
    out["stamps"] = stamps
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

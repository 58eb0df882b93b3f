#!/usr/bin/env python3
"""examples/psnark.rs --time-prover -i <logn> on the device path (examples/psnark.rs:70-81):
dummy_r1cs(2^logn), CommitterKey::new(num_constraints + num_variables, 5), Proof::index, Proof::new_time.
Prints the spans the reference prints with print-trace."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--instance-logsize", type=int, default=16)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--stepwise", action="store_true", help="the step-wise Python statement of the prover (tests/stepwise: the cross-check, one FFI call per step) instead of the native one")
    ap.add_argument("--native", action="store_true", help="the default since round 6 (kept for old command lines): the provers compiled into the library")
    ap.add_argument("--elastic", action="store_true", help="Proof::new_elastic over device-resident streams, max_msm_buffer = 2^20 "
                    "(examples/psnark.rs elastic_snark_main) instead of --time-prover")
    ap.add_argument("--transport", choices=["shm", "hook", "rccl", "rccl-node"], default=None, help="N ranks through the collective layer inside the library "
                    "(gemini_amd/csrc/dist.cpp): the key is an element-cyclic share and the provers compiled into the library commit through gm_ck_*")
    ap.add_argument("--block-sharded", action="store_true", help="--transport: the FIELD side block-sharded as well (gm_psnark_new_time_sharded): every vector "
                    "of the prover in blocks of one size over the ranks, the key in per-level slices; any world size")
    ap.add_argument("--tail-log", type=int, default=10, help="--block-sharded: blocks shorter than 2^k elements are gathered")
    ap.add_argument("--random-r1cs", type=int, default=None, metavar="SEED", help="a satisfied random GENERAL R1CS (1-3 entries per row of A and B in random "
                    "columns, C diagonal: tools/run_snark.py random_rows) instead of dummy_r1cs")
    ap.add_argument("--verifiable-key", action="store_true", help="one more power than examples/psnark.rs:76 asks for: the reference's "
                    "time-prover key (2n + 1 powers) is one short of the longest committed polynomial (2n + 2 coefficients), so the proof "
                    "of the example's configuration does not verify (tests/test_oracle_verifier.py::test_reference_example_key_is_one_power_short)")
    args = ap.parse_args()
    import tests.stepwise  # noqa: F401 -- registers the step-wise cross-check (what --stepwise and the Python-level sharded keys use)
    import warnings

    import gemini_amd as gm
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof

    if not args.verifiable_key:
        warnings.filterwarnings("ignore", message="commit: polynomial of", category=RuntimeWarning)  # the reference's shape, knowingly

    # N > 1 (torch.distributed.run): KZG key sharded by powers (tests.stepwise.dist.ShardedCommitterKey), the
    # field arithmetic replicated; GM_BENCH_BACKEND / GM_BENCH_SINGLE_DEVICE as in bench.py
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("GM_BENCH_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
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
            collective.init_shm(rank, world, "/gm_run_psnark_%s" % os.environ.get("MASTER_PORT", "0"))
        elif args.transport == "rccl-node":  # the library's own communicator, the id through a shm segment that stays open as the side channel (bench.py's path)
            collective.init_rccl_node(rank, world, "/gm_run_psnark_node_%s" % os.environ.get("MASTER_PORT", "0"))
        elif world > 1:
            collective.init_hook_torch() if args.transport == "hook" else collective.init_rccl_from_torch()
        if world > 1:
            collective.selftest()
    n = 1 << args.instance_logsize
    rng = np.random.default_rng(2022420)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD
    e_inst = rnd()
    if args.random_r1cs is not None:
        from gemini_amd.circuit import R1cs, SparseMatrix
        from gemini_amd.fr import FrVec
        from run_snark import random_rows

        ra, rb, rc, zh = random_rows(n, args.random_r1cs)
        mats = [SparseMatrix.from_rows(rows, n) for rows in (ra, rb, rc)] + [SparseMatrix.from_rows(rows, n, transpose=True) for rows in (ra, rb, rc)]
        r1cs = R1cs(*mats, FrVec.from_host(zh), FrVec.from_host(zh[1:]), FrVec.from_host(zh[:1]))
    elif lib_dist and args.block_sharded:
        r1cs = None  # every rank builds ITS blocks of dummy_r1cs in closed form (PsnarkShard.dummy): nothing whole anywhere
    else:
        r1cs = dummy_r1cs(e_inst, n)
    t0 = time.perf_counter()
    tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    # examples/psnark.rs: time main num_constraints + num_variables powers (:76), elastic main 3 * instance_size + 1 (:62)
    # examples/psnark.rs: time main num_constraints + num_variables powers (:76), elastic main 3 * instance_size + 1 (:62); a random
    # general instance has up to 7 n joint non-zero entries and lookup vectors of nnz + n + 2 elements
    max_degree = 10 * n if args.random_r1cs is not None else 3 * n if args.elastic else 2 * n + int(args.verifiable_key)
    shard = None
    if lib_dist and args.block_sharded:
        from gemini_amd.sharded import PsnarkShard, PsnarkShardKey, psnark_new_time_sharded

        # dummy_r1cs in closed form per rank (nothing of size n on the host); a random instance is cut out of the whole one
        shard = PsnarkShard(r1cs, tail_log=args.tail_log) if args.random_r1cs is not None else PsnarkShard.dummy(e_inst, n, tail_log=args.tail_log)
        ck = PsnarkShardKey(max_degree, shard.block, args.tail_log, tau)
    elif lib_dist:
        from gemini_amd.sharded import cyclic_committer_key

        ck = cyclic_committer_key(max_degree, 5, tau)
    elif world > 1:
        from tests.stepwise.dist import ShardedCommitterKey

        ck = ShardedCommitterKey.new(max_degree, 5, tau, rank, world)
    else:
        ck = CommitterKey.new(max_degree, 5, tau)
    t_srs = time.perf_counter() - t0
    t0 = time.perf_counter()
    index = shard.index(ck) if shard else Proof.index(ck, r1cs, native=True) if lib_dist else Proof.index(ck, r1cs)
    t_index = time.perf_counter() - t0
    out = {"logn": args.instance_logsize, "srs_s": round(t_srs, 3), "index_s": round(t_index, 3), "runs": []}
    stamps = []  # clock readings around every proof, for tools/exposed_time.py --stamps
    clocks = lambda: {"boottime_ns": time.clock_gettime_ns(time.CLOCK_BOOTTIME), "monotonic_ns": time.clock_gettime_ns(time.CLOCK_MONOTONIC),
                      "realtime_ns": time.clock_gettime_ns(time.CLOCK_REALTIME)}
    gm.capi.mem_reset_peak()  # the high-water marks below are those of the proofs, not of the key / index setup
    for _ in range(args.repeat):
        stamps.append({"t0": clocks()})
        if shard:  # (the elastic prover's resident schedule is the same entry: time provers on the little-endian vectors)
            proof = psnark_new_time_sharded(shard, ck, index)
            if args.elastic:
                proof.spans["ark_gemini::psnark::elastic_prover"] = proof.spans.pop("ark_gemini::psnark::time_prover")
        elif args.elastic:
            from gemini_amd.circuit import R1csStream
            from gemini_amd.kzg import CommitterKeyStream

            stream = R1csStream(r1cs)
            if world > 1 and not lib_dist:
                from tests.stepwise.dist import ShardedCommitterKeyStream

                cks = ShardedCommitterKeyStream.from_sharded_key(ck)
            else:
                cks = CommitterKeyStream.from_committer_key(ck)
            proof = Proof.new_elastic(cks, stream, index, 1 << 20, native=not args.stepwise)
            stream.free()
        else:
            proof = Proof.new_time(ck, r1cs, index, native=not args.stepwise)
        stamps[-1]["t1"] = clocks()
        out["runs"].append({k: round(v, 4) for k, v in proof.spans.items()})
        out["proof_size_B"] = proof.compressed_size()
    out["mem_GB"] = {k: round(v / 1e9, 3) for k, v in gm.capi.mem_stats().items() if k != "spare_table_releases"}
    out["spare_table_releases"] = gm.capi.mem_stats()["spare_table_releases"]
    out["stamps"] = stamps
    key = "ark_gemini::psnark::elastic_prover" if args.elastic else "ark_gemini::psnark::time_prover"
    out["elastic_prover_s" if args.elastic else "time_prover_s"] = min(r[key] for r in out["runs"])
    import hashlib

    out["n_gpus"] = world
    if shard:
        out["layout"] = {"block": shard.block, "tail_log": shard.tail_log, "levels_by_family_length": {str(k): v for k, v in sorted(shard.levels.items())}}
    out["proof_sha256"] = hashlib.sha256(proof.serialize_compressed()).hexdigest()
    if lib_dist:
        tkey = "elastic_prover_s" if args.elastic else "time_prover_s"
        mine = np.frombuffer(hashlib.sha256(proof.serialize_compressed()).digest() + np.float64(out[tkey]).tobytes(), dtype=np.uint64)
        allr = collective.allgather_host(mine)
        assert (allr[:, :4] == allr[0, :4]).all(), "ranks produced different proofs"
        out[tkey] = float(allr[:, 4].view(np.float64).max())
        out["transport"] = collective.info()[2]
        out["collectives"] = collective.stats()
        collective.finalize()
        if world > 1 and args.transport not in ("shm", "rccl-node"):
            dist.destroy_process_group()
    elif world > 1:
        allt = [None] * world
        tkey = "elastic_prover_s" if args.elastic else "time_prover_s"
        dist.all_gather_object(allt, (out[tkey], out["proof_sha256"]))
        assert len({d for _, d in allt}) == 1, "ranks produced different proofs"
        out[tkey] = max(t for t, _ in allt)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

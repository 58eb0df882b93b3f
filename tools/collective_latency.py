#!/usr/bin/env python3
"""Latency of the small host all-gathers of the sharded provers by route (review r4 item 3): 64 B (a sumcheck round), 144 B (a partial
G1 point), 1 KiB, 64 KiB (the gathered tail of a sharded sumcheck) over ncclAllGather with host staging (pinned -> device ->
collective -> pinned, one stream wait) and over the side segment gm_dist_init_rccl_node keeps open.  Runs with any world size the
launcher gives it (one rank per GPU; on the one-GPU box: world 1 -- the RCCL figure is then the floor of the staging, not of xGMI)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import gemini_amd as gm
    from gemini_amd import collective

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    gm.capi.init(int(os.environ.get("LOCAL_RANK", "0")))
    collective.init_rccl_node(rank, world, "/gm_latency_%s" % os.environ.get("MASTER_PORT", str(os.getppid())))
    out = {"world": world, "usec_per_call": {}}
    for nbytes in (64, 144, 1024, 65536):
        row = {}
        for route in ("rccl_host_staged", "shm"):
            best = min(collective.bench(nbytes, 300, collective.CLASS_FIELD, route) for _ in range(3))
            row[route] = round(best, 2)
        out["usec_per_call"][str(nbytes)] = row
    out["default_routes"] = {"field": "shm", "g1_points": "rccl_host_staged (GM_DIST_G1_ROUTE=shm overrides)"}
    collective.finalize()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

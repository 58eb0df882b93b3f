"""ctypes face of the library's collective layer (gemini_amd/csrc/dist.cpp, include/gemini_hip.h "Multi-GPU"): the all-gather
the sharded provers call INSIDE the library, over RCCL, a shared-memory segment or a hook.  The reference has no multi-device
code; what is sharded is its loop structure (src/snark/time_prover.rs:19-117, src/subprotocols/sumcheck/proof.rs:36-66,
src/kzg/time.rs:81-107).  `tests/stepwise/dist.py` is the older Python composition over torch.distributed; this module only
selects the transport -- the provers themselves are gm_snark_new_time_sharded & co."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

TRANSPORTS = {0: "none", 1: "hook", 2: "rccl", 3: "shm", 4: "failed"}
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_keep = []  # callbacks handed to the library must outlive it


def init_shm(rank: int, world: int, name: str, slot_bytes: int = 0) -> None:
    capi.check(capi.load().gm_dist_init_shm(C.c_int(rank), C.c_int(world), name.encode(), C.c_size_t(slot_bytes)))


def init_rccl_node(rank: int, world: int, name: str) -> None:
    """the library's own RCCL communicator on one node, the unique id carried through a shared-memory segment: no torch.distributed"""
    capi.check(capi.load().gm_dist_init_rccl_node(C.c_int(rank), C.c_int(world), name.encode()))


def init_hook(rank: int, world: int, fn) -> None:
    """fn(send: bytes) -> bytes of world payloads in rank order"""

    def tramp(_ctx, send, nbytes, recv):
        try:
            out = fn(C.string_at(send, nbytes))
            assert len(out) == nbytes * world
            C.memmove(recv, out, len(out))
            return 0
        except Exception:  # noqa: BLE001 -- nothing may unwind through the C frames
            import traceback

            traceback.print_exc()
            return -6

    cb = ALLGATHER_FN(tramp)
    _keep.append(cb)
    capi.check(capi.load().gm_dist_init_hook(C.c_int(rank), C.c_int(world), cb, None))


def init_hook_torch() -> None:
    """the hook over the process group torch.distributed already has (gloo on CPU tensors; with nccl the payload goes through a
    device tensor) -- what the shared-GPU tests use"""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"

    def gather(payload: bytes) -> bytes:
        loc = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        if on_gpu:
            loc = loc.cuda()
        out = torch.empty(world * len(payload), dtype=torch.uint8, device=loc.device)
        dist.all_gather_into_tensor(out, loc)
        return out.cpu().numpy().tobytes()

    init_hook(rank, world, gather)


def init_rccl_from_torch() -> None:
    """the library's own RCCL communicator; rank 0 draws the unique id and torch.distributed carries its 128 bytes to the peers"""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    uid = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        capi.check(capi.load().gm_dist_rccl_unique_id(uid.ctypes.data_as(C.POINTER(C.c_uint8))))
    box = [uid.tobytes()]
    dist.broadcast_object_list(box, src=0)
    uid = np.frombuffer(box[0], dtype=np.uint8).copy()
    capi.check(capi.load().gm_dist_init_rccl(C.c_int(rank), C.c_int(world), uid.ctypes.data_as(C.POINTER(C.c_uint8))))


def finalize() -> None:
    capi.check(capi.load().gm_dist_finalize())


def abort() -> None:
    """gm_dist_abort: this rank failed outside a collective -- release the peers waiting for it (they get GM_ESTATE), poison this rank's transport"""
    capi.check(capi.load().gm_dist_abort())


def info():
    r, w, t = C.c_int(), C.c_int(), C.c_int()
    capi.check(capi.load().gm_dist_info(C.byref(r), C.byref(w), C.byref(t)))
    return r.value, w.value, TRANSPORTS[t.value]


def stats(reset: bool = False) -> dict:
    n, b, s = C.c_uint64(), C.c_uint64(), C.c_double()
    capi.check(capi.load().gm_dist_stats(C.byref(n), C.byref(b), C.byref(s), C.c_int(int(reset))))
    return {"collectives": n.value, "bytes_received": b.value, "seconds": round(s.value, 6)}


ROUTES = ("copy", "hook", "rccl_host_staged", "rccl_device_vectors", "shm")
CLASS_FIELD, CLASS_G1 = 0, 1


def stats_routes() -> dict:
    """gm_dist_stats_routes: collectives, bytes received and seconds by the route they took"""
    n, b, s = (C.c_uint64 * 5)(), (C.c_uint64 * 5)(), (C.c_double * 5)()
    capi.check(capi.load().gm_dist_stats_routes(n, b, s))
    return {r: {"collectives": int(n[i]), "bytes_received": int(b[i]), "seconds": round(float(s[i]), 6)} for i, r in enumerate(ROUTES) if n[i]}


def bench(nbytes: int, iters: int = 200, payload_class: int = CLASS_FIELD, route: str | None = None) -> float:
    """gm_dist_bench: microseconds per all-gather of `nbytes` per rank; route None = where the class goes, else "rccl_host_staged" / "shm" """
    us = C.c_double()
    capi.check(capi.load().gm_dist_bench(C.c_size_t(nbytes), C.c_int(iters), C.c_int(payload_class), C.c_int(-1 if route is None else ROUTES.index(route)),
                                         C.byref(us)))
    return us.value


def selftest() -> None:
    capi.check(capi.load().gm_dist_selftest())


def allgather_host(local: np.ndarray, payload_class: int = CLASS_FIELD) -> np.ndarray:
    """(world, *local.shape) uint64"""
    loc = np.ascontiguousarray(local, dtype=np.uint64)
    _, world, _ = info()
    out = np.empty((world,) + loc.shape, dtype=np.uint64)
    capi.check(capi.load().gm_dist_allgather_host_class(loc.ctypes.data_as(C.c_void_p), C.c_size_t(loc.nbytes), out.ctypes.data_as(C.c_void_p),
                                                        C.c_int(payload_class)))
    return out


def allgather_vec(local, out=None):
    """FrVec of every rank's block back to back"""
    from .fr import FrVec

    _, world, _ = info()
    if out is None:
        out = FrVec.alloc(len(local) * world)
    capi.check(capi.load().gm_dist_allgather_vec(C.c_uint64(local.handle), C.c_uint64(out.handle)))
    return out


def reblock_vecs(local_vecs, new_block: int):
    """gm_dist_reblock_vecs: every vector is block-distributed with equal blocks; returns this rank's block [rank B, (rank + 1) B) of each"""
    from .fr import FrVec

    k = len(local_vecs)
    outs = [FrVec.alloc(max(new_block, 1)) for _ in range(k)]
    a = (C.c_uint64 * k)(*[v.handle for v in local_vecs])
    b = (C.c_uint64 * k)(*[v.handle for v in outs])
    capi.check(capi.load().gm_dist_reblock_vecs(a, C.c_size_t(k), C.c_size_t(new_block), b))
    return outs

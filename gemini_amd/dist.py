"""Multi-GPU composition of the hot path: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on a GPU node, "gloo" in the CPU tests).

The path shards by independent units, so there is no bulk collective:
  * MSM      -- pairs are split across ranks; each rank produces ONE 144-byte partial Jacobian
                point; all-gather (world x 144 B) + local EC add on every rank.  EC addition is not
                an RCCL reduce op, and at 1 KiB the payload is pure latency: one collective per MSM.
  * sumcheck -- contiguous even-aligned blocks; per round each rank contributes (a, b) = 64 bytes;
                all-gather + local addition mod r.  Folding is shard-local until the shards get
                short, then the tails are gathered and every rank finishes the protocol replicated.
The local compute is injected (`partial_msm`, prover objects), so the same code runs over the HIP
library on GPUs and over stand-ins in the gloo tests.
"""
from __future__ import annotations

import numpy as np

from .fr import R_MOD, _to_int, _to_limbs
from .msm import g1_sum


def _dist():
    import torch.distributed as dist

    return dist


def _device():
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if _dist().get_backend() == "nccl" else torch.device("cpu")


def all_gather_u64(local: np.ndarray) -> np.ndarray:
    """all-gather a small uint64 array; returns (world, *local.shape)."""
    import torch

    dist = _dist()
    world = dist.get_world_size()
    loc = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint64).view(np.int64).reshape(-1)).to(_device())
    out = torch.empty(world * loc.numel(), dtype=torch.int64, device=loc.device)
    dist.all_gather_into_tensor(out, loc)
    return out.cpu().numpy().view(np.uint64).reshape((world,) + tuple(np.shape(local)))


def shard_range(n: int, rank: int, world: int, align: int = 1):
    """contiguous block [lo, hi) of n units for `rank`, boundaries multiples of `align`"""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def msm_sharded(partial_msm, n: int) -> np.ndarray:
    """partial_msm(lo, hi) -> (18,) un-normalised Jacobian partial of pairs [lo, hi).
    Returns the normalised sum on every rank."""
    dist = _dist()
    lo, hi = shard_range(n, dist.get_rank(), dist.get_world_size())
    part = partial_msm(lo, hi)
    return g1_sum(all_gather_u64(part))


def fr_sum_allgather(vals_mont: np.ndarray) -> np.ndarray:
    """element-wise sum mod r over ranks of a (k, 4) array of Montgomery Fr (Montgomery form is linear)."""
    g = all_gather_u64(np.asarray(vals_mont, dtype=np.uint64).reshape(-1, 4))
    out = np.empty(g.shape[1:], dtype=np.uint64)
    for k in range(g.shape[1]):
        out[k] = _to_limbs(sum(_to_int(g[r, k]) for r in range(g.shape[0])) % R_MOD)
    return out


class ShardedTimeProver:
    """`trait Prover` (src/subprotocols/sumcheck/prover.rs:30-45) over per-rank shards.

    make_prover(f, g, twist) builds a local prover (gemini_amd.TimeProver on a GPU); it must offer
    next_message / rounds / final_foldings / set_shard / state() -> (f, g, twist).  The global
    vectors have length n (a power of two times world keeps every boundary even); this rank holds
    elements [lo, hi).
    """

    TAIL = 1 << 10

    def __init__(self, make_prover, f_local, g_local, twist_mont, lo: int, n_global: int):
        dist = _dist()
        self.make_prover = make_prover
        self.world = dist.get_world_size()
        self.n = n_global
        assert lo % 2 == 0
        self.local = make_prover(f_local, g_local, twist_mont)
        self.local.set_shard(lo // 2)
        self.replicated = False
        self.tot_rounds = (n_global - 1).bit_length() if n_global > 1 else 0
        self._round = 0
        self._cur_n = n_global

    def _should_gather(self) -> bool:
        # the global length after the pending fold is _cur_n / 2; keep shard-local while the shards
        # stay long and every boundary stays pair-aligned
        per = self._cur_n // self.world
        return not (per > self.TAIL and per % 4 == 0)

    def _gather(self):
        f, g, tw = self.local.state()
        fs = all_gather_u64(f).reshape(-1, 4)
        gs = all_gather_u64(g).reshape(-1, 4)
        self.local.free()
        self.local = self.make_prover(fs, gs, tw)  # its round count = the rounds that remain
        self.replicated = True

    def next_message(self, verifier_message=None):
        if not self.replicated and self._should_gather():
            # apply the pending fold shard-locally first, so the replicated prover starts exactly
            # at a message boundary (its own tot_rounds is then the number of messages left)
            if verifier_message is not None:
                self.local.fold(verifier_message)
                self._cur_n = (self._cur_n + 1) // 2
                verifier_message = None
            self._gather()
        msg = self.local.next_message(verifier_message)
        if verifier_message is not None:
            self._cur_n = (self._cur_n + 1) // 2
        if msg is None:
            return None
        self._round += 1
        if self.replicated:
            return msg
        s = fr_sum_allgather(np.stack(msg))
        return s[0], s[1]

    def rounds(self) -> int:
        return self.tot_rounds

    def final_foldings(self):
        return self.local.final_foldings()

    def free(self):
        self.local.free()

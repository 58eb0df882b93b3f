"""TEST INFRASTRUCTURE (moved out of the product package in round 6: the N-GPU provers of the product are the ones compiled into the library,
gemini_amd/csrc/{sharded,psnark_sharded,dist}.cpp; this is the older Python composition over torch.distributed, kept as a cross-check).

Multi-GPU composition of the hot path: one process per GPU, `torch.distributed` (backend "nccl" =
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

from gemini_amd.fr import R_MOD, _to_int, _to_limbs
from gemini_amd.msm import g1_sum


def _dist():
    import torch.distributed as dist

    return dist


def _device():
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if _dist().get_backend() == "nccl" else torch.device("cpu")


_GATHER_BUFS = {}  # (device, words) -> (pinned in, device in, device out, pinned out): ~100 collectives per proof reuse them


def all_gather_u64(local: np.ndarray) -> np.ndarray:
    """all-gather a small uint64 array; returns (world, *local.shape).  The payloads are host results (an MSM partial
    is finished by the host Horner), so on RCCL every collective is H2D -> all_gather -> D2H: the staging tensors are
    allocated once per size (pinned on the host side), the copies are asynchronous on the current stream and there is
    ONE wait, after the copy back."""
    import torch

    dist = _dist()
    if not dist.is_initialized():  # a single process: the "gather" of one rank
        return np.ascontiguousarray(local, dtype=np.uint64)[None].copy()
    world = dist.get_world_size()
    loc_np = np.ascontiguousarray(local, dtype=np.uint64).view(np.int64).reshape(-1)
    dev = _device()
    if dev.type == "cpu":
        loc = torch.from_numpy(loc_np)
        out = torch.empty(world * loc.numel(), dtype=torch.int64)
        dist.all_gather_into_tensor(out, loc)
        return out.numpy().view(np.uint64).reshape((world,) + tuple(np.shape(local)))
    key = (str(dev), loc_np.size, world)
    bufs = _GATHER_BUFS.get(key)
    if bufs is None:
        bufs = (torch.empty(loc_np.size, dtype=torch.int64).pin_memory(), torch.empty(loc_np.size, dtype=torch.int64, device=dev),
                torch.empty(world * loc_np.size, dtype=torch.int64, device=dev), torch.empty(world * loc_np.size, dtype=torch.int64).pin_memory())
        _GATHER_BUFS[key] = bufs
    h_in, d_in, d_out, h_out = bufs
    h_in.numpy()[:] = loc_np
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return h_out.numpy().view(np.uint64).reshape((world,) + tuple(np.shape(local))).copy()


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
        self.world = dist.get_world_size() if dist.is_initialized() else 1
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


def cyclic_count(length: int, rank: int, world: int) -> int:
    """how many of the indices 0 <= i < length satisfy i = rank (mod world)"""
    return max(0, -(-(length - rank) // world))


class ShardedCommitterKey:
    """`CommitterKey` (src/kzg/time.rs:24-27) sharded over the GPUs ELEMENT-CYCLICALLY: power i of the key lives on
    rank i mod world (local index i // world) -- the "independent MSM chunks shard across the GPUs, final reduce of
    partial G1 points" of the north star.

    Why cyclic and not contiguous blocks: every polynomial the prover commits to is a PREFIX of the key, and the
    tensor check commits to foldings of length n/2, n/4, ..., 1 (src/subprotocols/tensorcheck/mod.rs:124-133,
    190-275) against a key of 2n + 1 powers (examples/snark.rs:75).  With contiguous blocks the upper half of the
    ranks holds powers no commitment ever touches and every folding below n/4 lands on rank 0 alone; with the cyclic
    layout every rank gets length/world pairs (+-1) of EVERY commitment, whatever its length
    (tests/test_multi_gpu_gloo.py::test_cyclic_key_balance).

    Polynomials are replicated (every rank runs the same field arithmetic); a commitment is a strided gather of the
    rank's scalars (gm_fr_stride), one local MSM against the resident powers, one 144-byte all-gather and the EC adds,
    so `Proof.new_time(r1cs, key)` runs unchanged on N GPUs.

    local_msm(polynomial, m) -> (18,) Jacobian of sum over {i < m, i = rank mod world} of polynomial[i] *
    powers_of_g[i]; by default the HIP path over the resident share (gemini_amd.msm.G1Bases)."""

    def __init__(self, local_powers, rank: int, world: int, n_global: int, max_eval_points: int, local_msm=None, powers_of_g2=None):
        self.powers_of_g = local_powers
        self.rank, self.world = rank, world
        self.n_global = n_global
        assert len(local_powers) == cyclic_count(n_global, rank, world)
        self._max_eval_points = max_eval_points
        self._local_msm = local_msm or self._hip_msm
        self.powers_of_g2 = powers_of_g2  # replicated: max_eval_points + 1 G2 points

    @classmethod
    def new(cls, max_degree: int, max_eval_points: int, tau_canonical, rank: int, world: int, g_affine=None) -> "ShardedCommitterKey":
        """src/kzg/time.rs:49-72, each rank generating only its share: base tau^rank * g, ratio tau^world"""
        from gemini_amd.kzg import g1_generator_mont
        from gemini_amd.msm import G1Bases

        n = max_degree + 1
        g = g1_generator_mont() if g_affine is None else g_affine
        tau = _to_int(np.asarray(tau_canonical, dtype=np.uint64))
        first = G1Bases.fixed_base(g, np.array([_to_limbs(pow(tau, rank, R_MOD))], dtype=np.uint64))
        base = first.download()[0]
        first.free()
        from gemini_amd import g2 as G2

        powers_of_g2 = [G2.mul(G2.generator(), pow(tau, i, R_MOD)) for i in range(max_eval_points + 1)]
        ratio = np.array(_to_limbs(pow(tau, world, R_MOD)), dtype=np.uint64)
        return cls(G1Bases.srs(base, ratio, cyclic_count(n, rank, world)), rank, world, n, max_eval_points, powers_of_g2=powers_of_g2)

    def max_eval_points(self) -> int:
        return self._max_eval_points

    def num_powers(self) -> int:
        return self.n_global

    def global_indices(self) -> np.ndarray:
        """the powers this rank holds, in local order"""
        return np.arange(self.rank, self.n_global, self.world)

    def powers_of_g2_bytes(self) -> bytes:
        from gemini_amd.kzg import CommitterKey

        return CommitterKey.powers_of_g2_bytes(self)

    def _hip_msm(self, polynomial, m: int) -> np.ndarray:
        from gemini_amd.fr import _as_vec, stride
        from gemini_amd.msm import g1_zero

        cnt = cyclic_count(m, self.rank, self.world)
        if cnt == 0:
            return g1_zero()
        v, tmp = _as_vec(polynomial)
        try:
            if self.world == 1:
                return self.powers_of_g.msm_vec(v, n=cnt)
            mine = stride(v, self.rank, self.world, cnt)
            try:
                return self.powers_of_g.msm_vec(mine, n=cnt)
            finally:
                mine.free()
        finally:
            if tmp:
                v.free()

    def partial(self, polynomial) -> np.ndarray:
        """this rank's share of commit(polynomial)"""
        return self._local_msm(polynomial, min(len(polynomial), self.n_global))

    def commit(self, polynomial) -> np.ndarray:
        return g1_sum(all_gather_u64(self.partial(polynomial)))

    def batch_partials(self, polys) -> np.ndarray:
        """this rank's UN-NORMALISED shares of commit(p) for every p: the strided gathers of the rank's scalars, then ONE
        pipelined batch call (gm_g1_msm_v_batch_partial: two big lanes + four small ones, host tails under the next
        call's kernels) -- what CommitterKey.batch_commit does on one GPU.  (k, 18)"""
        from gemini_amd.fr import _as_vec, stride
        from gemini_amd.msm import g1_zero

        if self._local_msm != self._hip_msm:  # injected local compute (CPU tests)
            return np.stack([self.partial(p) for p in polys])
        out = np.tile(g1_zero(), (len(polys), 1))
        vecs, cnts, idx, tmps = [], [], [], []
        try:
            for j, p in enumerate(polys):
                cnt = cyclic_count(min(len(p), self.n_global), self.rank, self.world)
                if cnt == 0:
                    continue
                v, tmp = _as_vec(p)
                if tmp:
                    tmps.append(v)
                if self.world != 1:
                    v = stride(v, self.rank, self.world, cnt)
                    tmps.append(v)
                vecs.append(v)
                cnts.append(cnt)
                idx.append(j)
            if vecs:
                out[idx] = self.powers_of_g.msm_vec_batch(vecs, cnts, partial=True)
        finally:
            for v in tmps:
                v.free()
        return out

    def batch_commit(self, polynomials) -> list:
        """one batch of local MSMs, ONE all-gather for the whole batch (k x 144 bytes), one normalisation per commitment"""
        polys = list(polynomials)
        if not polys:
            return []
        parts = all_gather_u64(self.batch_partials(polys))  # (world, k, 18)
        return [g1_sum(parts[:, k]) for k in range(len(polys))]

    # the openings are commitments to quotients computed (replicated) on every rank: same code as the
    # single-GPU key, with `commit` above
    def open_multi_points(self, polynomial, eval_points_mont):
        from gemini_amd.kzg import CommitterKey

        return CommitterKey.open_multi_points(self, polynomial, eval_points_mont)

    def batch_open_multi_points(self, polynomials, eval_points_mont, eval_chal_mont):
        from gemini_amd.kzg import CommitterKey

        return CommitterKey.batch_open_multi_points(self, polynomials, eval_points_mont, eval_chal_mont)


def _stream_key_base():
    from gemini_amd.kzg import CommitterKeyStream

    return CommitterKeyStream


class ShardedCommitterKeyStream(_stream_key_base()):
    """`CommitterKeyStream` (src/kzg/space.rs:59-69) over the same element-cyclic shares -- "MSM chunks sharded across
    the GPUs" for the elastic prover (BASELINE config 4).  Every stream MSM (commit, open, open_multi_points,
    commit_folding, open_folding all funnel into `_msm_stream`) is: the stream positions whose power this rank
    holds (a stride-`world` subsequence), one local MSM walking the resident share backwards, one all-gather of
    144-byte partials, the EC adds.  The streams are replicated."""

    def __init__(self, local_powers, rank: int, world: int, n_global: int, max_eval_points: int, powers_of_g2=None, min_device_chunk=None):
        super().__init__(local_powers, max_eval_points, powers_of_g2, min_device_chunk=min_device_chunk)
        self.rank, self.world, self.n_global = rank, world, n_global

    @classmethod
    def from_sharded_key(cls, key: "ShardedCommitterKey", min_device_chunk=None) -> "ShardedCommitterKeyStream":
        return cls(key.powers_of_g, key.rank, key.world, key.n_global, key.max_eval_points(), getattr(key, "powers_of_g2", None), min_device_chunk)

    def _n(self) -> int:
        return self.n_global

    def _msm_stream(self, scalars_stream, first_stream_pos: int, chunk: int) -> np.ndarray:
        from gemini_amd.fr import stride
        from gemini_amd.msm import g1_zero

        n, total, w = self.n_global, len(scalars_stream), self.world
        # stream position p pairs with power i = n - 1 - (first_stream_pos + p); this rank holds i = rank (mod w):
        # positions p0, p0 + w, ... with local base indices j0, j0 - 1, ...
        top = n - 1 - first_stream_pos
        p0 = (top - self.rank) % w
        cnt = cyclic_count(total, p0, w)
        part = g1_zero()
        if cnt:
            j0 = (top - p0 - self.rank) // w
            mine = scalars_stream if w == 1 else stride(scalars_stream, p0, w, cnt)
            try:
                step = max(1, max(chunk, self.min_device_chunk) // w)  # the caller's flush size, in this rank's pairs
                for off in range(0, cnt, step):
                    m = min(step, cnt - off)
                    p = self.powers_of_g.msm_vec(mine, n=m, voffset=off, offset=j0 - off, reversed_=True)
                    part = g1_sum(np.stack([part, p]))
            finally:
                if w != 1:
                    mine.free()
        return g1_sum(all_gather_u64(part))

"""TEST INFRASTRUCTURE (moved out of the product package in round 6: the N-GPU provers of the product are the ones compiled into the library,
gemini_amd/csrc/{sharded,psnark_sharded,dist}.cpp; this is the older Python composition over torch.distributed, kept as a cross-check).

`snark::Proof::new_time` (src/snark/time_prover.rs:19-117) with the FIELD ARITHMETIC sharded over the GPUs as well.

`gemini_amd/dist.py` shards the MSMs (element-cyclic key) and leaves every O(n) field pass replicated.  Here every vector of
the prover is BLOCK-sharded -- rank r of g holds elements [r m, (r + 1) m), m = n / g -- and so is the key, in per-level slices:

  * sumchecks            `dist.ShardedTimeProver` (64 bytes all-gathered per round, tails gathered when the blocks get short)
  * tensor / powers      block r of tensor(rho) is a scalar times tensor(rho[:log m]); of powers(alpha) it is alpha^(r m) powers(alpha, m)
  * matrix products      block-diagonal instances only (the reference's benchmark instance is diagonal, src/circuit.rs:349-365)
  * foldings             fold(2i, 2i + 1) -> i keeps the top log g index bits, so level j of the tree is block-sharded with block
                         length m / 2^j; a level whose blocks fall below 2^tail_log elements is gathered and finished replicated
  * commitments          level j's block meets its powers on the same rank: the key slice of level j on rank r is
                         [r m / 2^j, (r + 1) m / 2^j) -- 2 m powers per rank in all; one all-gather of k x 144 bytes per batch
  * evaluations          p(x) = sum_r x^(lo_r) P_r(x): a block evaluation, one all-gather of field elements
  * the opening          commit((sum_i eta_i p_i) / Z) = sum_i eta_i commit(p_i div Z) (division by the same Z is linear).  The
                         quotient of a block-sharded p needs the carry from the blocks above, which is the polynomial of
                         degree < 3 that agrees with S_r(x) = sum_{r' > r} x^(lo_r' - hi_r) P_r'(x) at the three roots of Z -- and
                         those values come from the block evaluations that were all-gathered for the transcript anyway.
                         Rank r divides (carry * x^L + P_r) locally: L + 3 coefficients, one device division, no second pass.
                         Price: the opening is one MSM per polynomial in its own layout (2 n / g pairs per rank instead of n / g).

Per rank the field work is (total / g) + O(g 2^tail_log) + the host algebra of the carries; the MSM work is 4 n / g pairs
(witness n / g, foldings n / g, opening 2 n / g).  The proof is byte-identical to the single-GPU one
(tests/test_gpu_world2.py runs 2 and 4 ranks on one GPU over gloo).  Not measured on a multi-GPU node.
"""
from __future__ import annotations

import time

import numpy as np

from tests.stepwise.dist import ShardedTimeProver, all_gather_u64, fr_sum_allgather
from gemini_amd.fr import (FrVec, R_MOD, div_vanishing, evaluate_le_batch, fold_polynomial, fr_from_int, fr_to_int, hadamard, linear_combination,
                 powers, tensor)
from gemini_amd.msm import g1_sum, g1_zero

TAIL_LOG = 10


class BlockLayout:
    def __init__(self, n: int, rank: int, world: int, tail_log: int = TAIL_LOG):
        assert n & (n - 1) == 0 and world & (world - 1) == 0 and n % world == 0, "block sharding needs powers of two"
        self.n, self.rank, self.world = n, rank, world
        self.m = n // world
        self.logn = n.bit_length() - 1
        self.tail = 1 << tail_log
        assert self.m >= self.tail >= 8, "blocks shorter than the tail length: use the replicated prover"
        # level j (0 = the polynomial itself, j >= 1 its foldings) is block-sharded while its blocks hold >= tail elements
        self.jmax = 0
        while (self.m >> (self.jmax + 1)) >= self.tail:
            self.jmax += 1

    def block_len(self, j: int) -> int:
        return self.m >> j

    def lo(self, j: int, rank=None) -> int:
        return (self.rank if rank is None else rank) * (self.m >> j)


class BlockShardedKey:
    """`CommitterKey` (src/kzg/time.rs:24-27) in per-level block slices, all in ONE registered key: segment j holds powers
    [lo_j, lo_j + m / 2^j) of this rank for j = 0 .. jmax, the last segment the first n / 2^(jmax + 1) powers (the gathered levels,
    the same on every rank).  One handle means one pipelined batch call serves MSMs of several levels (`commit`)."""

    PREFIX = -1

    def __init__(self, layout: BlockLayout, bases, offsets, counts, max_eval_points: int):
        self.layout, self.bases, self.offsets, self.counts = layout, bases, offsets, counts
        self._max_eval_points = max_eval_points

    @classmethod
    def new(cls, n_poly: int, max_eval_points: int, tau_canonical, rank: int, world: int, tail_log: int = TAIL_LOG, g_affine=None):
        from gemini_amd.kzg import g1_generator_mont
        from gemini_amd.msm import G1Bases

        L = BlockLayout(n_poly, rank, world, tail_log)
        g = g1_generator_mont() if g_affine is None else g_affine
        tau_l = np.asarray(tau_canonical, dtype=np.uint64).reshape(4)
        starts = [L.lo(j) for j in range(L.jmax + 1)] + [0]
        counts = [L.block_len(j) for j in range(L.jmax + 1)] + [max(L.n >> (L.jmax + 1), 1)]
        offsets = [int(v) for v in np.concatenate([[0], np.cumsum(counts)[:-1]])]
        return cls(L, G1Bases.srs_segments(g, tau_l, starts, counts), offsets, counts, max_eval_points)

    def commit(self, levels, vecs, partial: bool = True) -> np.ndarray:
        """MSMs of vecs[i] against the slice of level levels[i] (PREFIX: the replicated prefix), one pipelined batch; (k, 18).
        A vector longer than its slice is cut (the quotient of a block with its carry appended never is)."""
        offs = [self.offsets[lv] for lv in levels]
        ns = [min(len(v), self.counts[lv]) for lv, v in zip(levels, vecs)]
        return self.bases.msm_vec_batch_at(vecs, ns, offs, partial=partial)

    def free(self):
        self.bases.free()


class R1csBlock:
    """the rows [r m, (r + 1) m) of a BLOCK-DIAGONAL R1CS instance: local CSR blocks with local column indices"""

    def __init__(self, a, b, c, at, bt, ct, z_blk: FrVec, w_blk: FrVec, layout: BlockLayout):
        self.a, self.b, self.c, self.at, self.bt, self.ct = a, b, c, at, bt, ct
        self.z, self.w, self.layout = z_blk, w_blk, layout

    @classmethod
    def dummy(cls, e_canonical: int, layout: BlockLayout) -> "R1csBlock":
        """this rank's block of dummy_r1cs(e, n) (src/circuit.rs:349-365): z = [e; n], w = [e; n - 1], A = B = C = diag(1 / e)"""
        from gemini_amd.circuit import SparseMatrix

        m = layout.m
        e = e_canonical % R_MOD
        d = SparseMatrix.from_csr(np.arange(m + 1, dtype=np.uint64), np.arange(m, dtype=np.uint32), np.tile(fr_from_int(pow(e, -1, R_MOD)), (m, 1)), m, m)
        z = FrVec.alloc(m)
        z.fill(fr_from_int(e))
        w = FrVec.alloc(m - 1 if layout.rank == layout.world - 1 else m)
        w.fill(fr_from_int(e))
        return cls(d, d, d, d, d, d, z, w, layout)

    def free(self):
        seen = set()
        for mtx in (self.a, self.b, self.c, self.at, self.bt, self.ct):
            if id(mtx) not in seen:
                seen.add(id(mtx))
                mtx.free()
        self.z.free()
        self.w.free()


def _interp3(xs, ys):
    """coefficients (c0, c1, c2) of the polynomial of degree < 3 through three points (integers mod r)"""
    c = [0, 0, 0]
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        den = (xs[i] - xs[j]) * (xs[i] - xs[k]) % R_MOD
        assert den, "evaluation points coincide"
        s = ys[i] * pow(den, -1, R_MOD) % R_MOD
        c[0] = (c[0] + s * xs[j] * xs[k]) % R_MOD
        c[1] = (c[1] - s * (xs[j] + xs[k])) % R_MOD
        c[2] = (c[2] + s) % R_MOD
    return c


def _alloc_spare(n: int) -> FrVec:
    """a vector of n elements with room for the three carry coefficients of the opening behind it"""
    v = FrVec.alloc(n + 3)
    v.set_len(n)
    return v


def _fold_spare(cur: FrVec, chal) -> FrVec:
    import ctypes as C

    from gemini_amd import capi

    out = _alloc_spare((len(cur) + 1) // 2)
    capi.check(capi.load().gm_fr_fold(C.c_uint64(cur.handle), capi.ptr(capi.u64(chal).reshape(4)), C.c_uint64(out.handle)))
    return out


def _append_carry(blk: FrVec, coeffs):
    """blk (allocated by _alloc_spare) becomes [blk..., c0, c1, c2] in place"""
    import ctypes as C

    from gemini_amd import capi

    n = len(blk)
    blk.set_len(n + 3)
    tail = np.stack([fr_from_int(v) for v in coeffs])
    capi.check(capi.load().gm_fr_vec_upload(C.c_uint64(blk.handle), C.c_size_t(n), capi.ptr(tail), C.c_size_t(3)))


def fr_work(n: int, g: int, tail_log: int = TAIL_LOG) -> dict:
    """Field elements read + written by the device passes of RANK 0 of new_time_block_sharded, by phase (the sumchecks, which
    `ShardedTimeProver` runs, apart: fr_work_sumcheck) -- the pure statement of the accounting the prover does as it runs
    (`proof.fr_work`; tests/test_gpu_world2.py holds the two equal).  g = 1 is the unsharded total, so
    sum(fr_work(n, g)) <= 1.1 * sum(fr_work(n, 1)) / g is the scaling claim (tests/test_multi_gpu_gloo.py)."""
    L = BlockLayout(n, 0, g, tail_log)
    m = L.m
    w = {"matrix products": 3 * 2 * m, "zc(alpha)": m, "tensor/powers/hadamard": m + m + 3 * m, "abc_tensored": 3 * 2 * m + 4 * m, "body": 3 * m}
    fold, cur = 0, m
    sharded, small = [], []
    for j in range(1, L.logn):
        nxt = (cur + 1) // 2
        fold += cur + nxt
        if j <= L.jmax:
            sharded.append(nxt)
        elif j == L.jmax + 1:
            nxt *= g
            small.append(nxt)
        else:
            small.append(nxt)
        cur = nxt
    w["foldings"] = fold
    blocks = [m if g > 1 else m - 1] + sharded  # w (n - 1 coefficients: the top rank's block is one short) and the sharded levels
    w["evaluations"] = sum(blocks) + sum(small)
    opening = 0
    for Lb in blocks:
        d = Lb + 3 if g > 1 else Lb  # rank 0 carries unless it is also the top rank
        opening += 3 * d + (d - 3)
    if small:
        comb = max(small)
        opening += sum(small) + comb
        if comb > 3:
            opening += 3 * comb + (comb - 3)
    w["opening"] = opening
    return w


def fr_work_sumcheck(n: int, g: int) -> int:
    """one sumcheck on rank 0: shard-local rounds (read f and g, write the folded halves) while the blocks stay longer than
    ShardedTimeProver.TAIL, then the gathered tail replicated"""
    T = ShardedTimeProver.TAIL
    tot, cur, glob = 0, n // g, n
    while glob > 1:
        per = glob // g
        if g > 1 and not (per > T and per % 4 == 0):
            cur = glob  # gathered: every rank holds the whole (short) vectors from here on
            g = 1
        tot += 2 * cur + 2 * ((cur + 1) // 2)
        cur = (cur + 1) // 2
        glob = (glob + 1) // 2
    return tot


class _Acct(dict):
    def add(self, phase: str, *lens):
        self[phase] = self.get(phase, 0) + int(sum(lens))


def new_time_block_sharded(r1cs: R1csBlock, key: BlockShardedKey):
    """Proof::new_time on every rank's block; returns the same `snark.Proof` on all ranks (with `.fr_work`: the field elements
    this rank's device passes read + wrote, by phase; the sumchecks are accounted by `ShardedTimeProver`'s own rounds)"""
    from gemini_amd.msm import VariableBaseMSM
    from gemini_amd.snark import Proof
    from gemini_amd.sumcheck import Sumcheck, TimeProver
    from gemini_amd.tensorcheck import TensorcheckProof
    from gemini_amd.transcript import PROTOCOL_NAME, Transcript

    L = key.layout
    n, g, r, m = L.n, L.world, L.rank, L.m
    spans = {}
    acct = _Acct()
    t_all = time.perf_counter()
    F = fr_from_int
    I = fr_to_int
    make = lambda f, gg, tw: TimeProver(f, gg, tw)  # noqa: E731

    def eval_blocks(blocks, pts, los):
        """values p(x) for block-sharded polynomials: local block evaluations (returned unscaled, all ranks') + the sums"""
        local = evaluate_le_batch(blocks, pts)  # (k, npts, 4)
        allr = all_gather_u64(local)  # (g, k, npts, 4)
        vals = []
        for k in range(len(blocks)):
            row = []
            for q in range(len(pts)):
                x = I(pts[q])
                row.append(sum(pow(x, los[k](rr), R_MOD) * I(allr[rr, k, q]) for rr in range(g)) % R_MOD)
            vals.append(row)
        return allr, vals

    t0 = time.perf_counter()
    z_a, z_b, z_c = r1cs.a.mul(r1cs.z), r1cs.b.mul(r1cs.z), r1cs.c.mul(r1cs.z)  # :32-34, block-diagonal
    acct.add("matrix products", 3 * (len(r1cs.z) + len(z_a)))
    spans["product_matrix_vector x3"] = time.perf_counter() - t0
    transcript = Transcript(PROTOCOL_NAME)
    t0 = time.perf_counter()
    part = key.commit([0], [r1cs.w])[0]  # ck.commit(&r1cs.w) :42
    witness_commitment = g1_sum(all_gather_u64(part))
    spans["Commitment to w"] = time.perf_counter() - t0
    transcript.append_g1(b"witness", witness_commitment)
    alpha = transcript.get_challenge(b"alpha")
    ai = I(alpha)
    _, zc = eval_blocks([z_c], alpha.reshape(1, 4), [lambda rr: rr * m])
    acct.add("zc(alpha)", len(z_c))
    zc_alpha = F(zc[0][0])  # :48
    transcript.append_fr(b"zc(alpha)", zc_alpha)

    t0 = time.perf_counter()
    p1 = ShardedTimeProver(make, z_a, z_b, alpha, r * m, n)
    first_proof = Sumcheck.prove(transcript, p1)  # :52
    p1.free()
    spans["First sumcheck"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    ch1 = [I(c) for c in first_proof.challenges]
    k = m.bit_length() - 1
    s_b = 1
    for j in range(k, len(ch1)):  # the index bits above the block select a factor each
        if (r >> (j - k)) & 1:
            s_b = s_b * ch1[j] % R_MOD
    s_c = pow(ai, r * m, R_MOD)
    # block r of tensor(rho) = s_b * tensor(rho[:log m]); of powers(alpha) = s_c * powers(alpha, m): the scalars go into the
    # coefficients of the linear combination below instead of into passes of their own                                  :56-58
    b_ch = tensor(np.stack(first_proof.challenges[:k]))
    c_ch = powers(alpha, m)
    a_ch = hadamard(b_ch, c_ch)
    acct.add("tensor/powers/hadamard", len(b_ch), len(c_ch), 3 * len(a_ch))
    eta = transcript.get_challenge(b"eta")
    ei = I(eta)
    ta, tb, tc = r1cs.at.mul(a_ch), r1cs.bt.mul(b_ch), r1cs.ct.mul(c_ch)  # :63-81
    abc = linear_combination([ta, tb, tc], np.stack([F(s_b * s_c % R_MOD), F(ei * s_b % R_MOD), F(ei * ei % R_MOD * s_c % R_MOD)]))
    abc.set_len(m)
    acct.add("abc_tensored", 3 * 2 * m, 4 * m)
    for v in (ta, tb, tc, a_ch, b_ch, c_ch):
        v.free()
    spans["tensor/powers/hadamard/abc_tensored"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    p2 = ShardedTimeProver(make, abc, r1cs.z, F(1), r * m, n)
    second_proof = Sumcheck.prove(transcript, p2)  # :84-89
    p2.free()
    spans["Second sumcheck"] = time.perf_counter() - t0

    # ---- TensorcheckProof::new_time(transcript, ck, [w], [([abc_tensored, z], challenges)])   tensorcheck/mod.rs:190-275
    t0 = time.perf_counter()
    batch_challenge = transcript.get_challenge(b"batch_challenge")
    body = linear_combination([abc, r1cs.z], np.stack([F(1), batch_challenge]))
    body.set_len(m)
    acct.add("body", 3 * m)
    tc_ch = list(second_proof.challenges)[:-1]  # strip_last
    sharded = []  # level j = 1 .. jmax: this rank's block (allocated with room for the opening's carry)
    small = []  # the gathered levels, replicated
    cur = body
    for j, chal in enumerate(tc_ch, start=1):
        nxt = _fold_spare(cur, chal)
        acct.add("foldings", len(cur), len(nxt))
        if j <= L.jmax:
            sharded.append(nxt)
        elif j == L.jmax + 1:  # its blocks are shorter than the tail: gather, finish replicated
            full = all_gather_u64(nxt.to_host()).reshape(-1, 4)
            nxt.free()
            nxt = FrVec.from_host(full)
            small.append(nxt)
        else:
            small.append(nxt)
        cur = nxt
    # commitments: one pipelined batch over the sharded levels (against their key slices) and the small ones (against the
    # replicated prefix), un-normalised; one all-gather for the sharded ones
    parts = key.commit(list(range(1, len(sharded) + 1)) + [key.PREFIX] * len(small), sharded + small)
    commitments = []
    if sharded:
        gathered = all_gather_u64(parts[: len(sharded)])
        commitments = [g1_sum(gathered[:, i]) for i in range(len(sharded))]
    commitments += [g1_sum(parts[len(sharded) + i].reshape(1, 18)) for i in range(len(small))]
    for cm in commitments:
        transcript.append_g1(b"commitment", cm)
    eval_chal = transcript.get_challenge(b"evaluation-chal")
    ec = I(eval_chal)
    pts = np.stack([F(ec * ec % R_MOD), eval_chal, F((-ec) % R_MOD)])
    pts_i = [I(p) for p in pts]
    # block evaluations at all three roots of Z (the transcript takes beta^2 for w only; the carries of the opening need it everywhere)
    # w gets its own copy with room for the carry (the instance's vector is not ours to extend)
    w_blk = _alloc_spare(len(r1cs.w))
    import ctypes as C

    from gemini_amd import capi

    capi.check(capi.load().gm_fr_stride(C.c_uint64(r1cs.w.handle), C.c_size_t(0), C.c_size_t(1), C.c_size_t(len(r1cs.w)), C.c_uint64(w_blk.handle)))
    blocks = [w_blk] + sharded
    los = [(lambda rr, j=j: rr * (m >> j)) for j in range(len(blocks))]
    allr, vals = eval_blocks(blocks, pts, los)
    acct.add("evaluations", *[len(b) for b in blocks])
    evaluations_w = np.stack([F(v) for v in vals[0]])
    fold_evals = [np.stack([F(v[1]), F(v[2])]) for v in vals[1:]]
    if small:
        fold_evals += list(evaluate_le_batch(small, pts[1:]))
        acct.add("evaluations", *[len(v) for v in small])
    for e3 in evaluations_w:
        transcript.append_fr(b"eval", e3)
    for e2 in fold_evals:
        for e1 in e2:
            transcript.append_fr(b"eval", e1)
    open_chal = transcript.get_challenge(b"open-chal")
    oi = I(open_chal)
    # the opening: sum_i eta_i commit(p_i div Z), p_0 = w, p_i = level i.  Per polynomial: the carry, one division, one MSM against the
    # level's key slice; the eta_i are applied to the (normalised) partial points in one tiny MSM, not to the vectors
    quots, q_levels, q_idx = [], [], []
    for i, blk in enumerate(blocks):
        Lb = m >> i
        if r < g - 1:  # the carry from the blocks above: the degree < 3 polynomial with S_r's values at the roots
            ys = [sum(pow(x, (rr - r - 1) * Lb, R_MOD) * I(allr[rr, i, q]) for rr in range(r + 1, g)) % R_MOD for q, x in enumerate(pts_i)]
            _append_carry(blk, _interp3(pts_i, ys))
        q, _ = div_vanishing(blk, pts)
        acct.add("opening", 3 * len(blk), len(q))
        if len(q):
            quots.append(q)
            q_levels.append(i)
            q_idx.append(i)
        else:
            q.free()
    comb_at = None
    if small:
        etas = np.stack([F(pow(oi, len(blocks) + i, R_MOD)) for i in range(len(small))])
        comb = linear_combination(small, etas)
        acct.add("opening", *[len(v) for v in small], len(comb))
        if len(comb) > 3:
            q, _ = div_vanishing(comb, pts)
            acct.add("opening", 3 * len(comb), len(q))
            comb_at = len(quots)
            quots.append(q)
            q_levels.append(key.PREFIX)
        comb.free()
    # every quotient through one pipelined batch; normalised one by one (Z = 1 or the identity) for the tiny eta-MSM
    qparts = key.commit(q_levels, quots) if quots else np.empty((0, 18), dtype=np.uint64)
    for q in quots:
        q.free()
    my_points, my_etas = [], []
    for t, i in enumerate(q_idx):
        pt = g1_sum(qparts[t].reshape(1, 18))
        if any(pt[12:]):  # not the identity
            my_points.append(pt[:12])
            my_etas.append(np.array([(pow(oi, i, R_MOD) >> (64 * w_)) & (2**64 - 1) for w_ in range(4)], dtype=np.uint64))
    mine = VariableBaseMSM.msm_bigint(np.stack(my_points), np.stack(my_etas)) if my_points else g1_zero()
    pieces = [all_gather_u64(mine)]
    if comb_at is not None:
        pieces.append(qparts[comb_at].reshape(1, 18))
    evaluation_proof = g1_sum(np.concatenate([p.reshape(-1, 18) for p in pieces]))
    for v in blocks + small + [body, abc, z_a, z_b, z_c]:
        v.free()
    spans["Tensorcheck"] = time.perf_counter() - t0
    transcript.free()
    spans["ark_gemini::snark::time_prover"] = time.perf_counter() - t_all
    tcp = TensorcheckProof(commitments, fold_evals, evaluation_proof, [evaluations_w])
    proof = Proof(witness_commitment, zc_alpha, (first_proof.messages, first_proof.final_foldings),
                  (second_proof.messages, second_proof.final_foldings), tcp)
    proof.spans = spans
    proof.fr_work = dict(acct)
    return proof

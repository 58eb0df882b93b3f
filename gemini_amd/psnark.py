"""Host mirror of the preprocessing SNARK time prover (src/psnark/time_prover.rs:49-384), the
entry-product argument (src/subprotocols/entryproduct/time_prover.rs:53-114) and the plookup vector
builders (src/subprotocols/plookup/time_prover.rs:89-112): orchestration only -- every O(n) step is a
device call (MSM, batched sumcheck, gather / hash / scan / vector passes)."""
from __future__ import annotations

import time

import numpy as np

from .circuit import R1cs
from .fr import (FrVec, IdxVec, R_MOD, accumulated_product_monic, alg_hash, element, evaluate_le, evaluate_le_batch, fr_from_int, fr_to_int, hadamard, ip,
                 linear_combination, lookup, plookup_set, plookup_subset, powers, shift_monic, tensor)
from .kzg import CommitterKey
from .sumcheck import Sumcheck, TimeProver
from .tensorcheck import TensorcheckProof
from .transcript import PROTOCOL_NAME, Transcript

_ONE = fr_from_int(1)


# ---- src/misc.rs:269-366 on the host CSR arrays -------------------------------------------------------
def joint_matrices(a, b, c, num_constraints: int, num_variables: int):
    """sum_matrices + joint_matrices: the union of the supports of A, B, C walked column-major
    (column ascending, row ascending inside a column -- BTreeSet order), with the three value
    vectors (zero where a matrix has no entry; the last duplicate wins like BTreeMap::collect).
    Returns (row_index, col_index, val_a, val_b, val_c) as numpy arrays (values Montgomery (nnz, 4))."""
    keys, vals = {}, {}
    for m in (a, b, c):
        if id(m) in keys:  # the same matrix object used twice (dummy_r1cs: A = B = C)
            continue
        rowptr, cols, v = m.csr
        rows = np.repeat(np.arange(m.nrows, dtype=np.uint64), np.diff(rowptr.astype(np.int64)))
        assert cols.size == 0 or int(cols.max()) < num_variables, "column index outside num_variables"
        keys[id(m)] = cols.astype(np.uint64) * np.uint64(num_constraints) + rows
        vals[id(m)] = np.asarray(v, dtype=np.uint64).reshape(-1, 4)
    union = np.unique(np.concatenate(list(keys.values())))
    row_index = (union % np.uint64(num_constraints)).astype(np.uint32)
    col_index = (union // np.uint64(num_constraints)).astype(np.uint32)
    dense = {}
    for mid, k in keys.items():
        d = np.zeros((len(union), 4), dtype=np.uint64)
        order = np.argsort(k, kind="stable")
        ks = k[order]
        last = np.ones(len(ks), dtype=bool)
        last[:-1] = ks[1:] != ks[:-1]  # the last occurrence of every key
        if not last.all():
            import warnings

            warnings.warn("psnark: a matrix row repeats a column; the joint-matrix encoding keeps the last entry only (as the "
                          "reference's BTreeMap does, src/misc.rs:269-366) and the resulting proof will NOT verify -- coalesce "
                          "repeated (row, column) entries before building the R1CS", RuntimeWarning, stacklevel=3)
        d[np.searchsorted(union, ks[last])] = vals[mid][order][last]
        dense[mid] = d
    return row_index, col_index, dense[id(a)], dense[id(b)], dense[id(c)]


def _joint(r1cs: R1cs):
    """joint_matrices of an instance, computed once per R1cs object: they depend on the matrices only
    (the reference recomputes them in `index` and in `new_time`, src/psnark/time_prover.rs:53-61,102-110)."""
    j = getattr(r1cs, "_joint_matrices", None)
    if j is None:
        j = joint_matrices(r1cs.a, r1cs.b, r1cs.c, r1cs.a.nrows, len(r1cs.z))
        r1cs._joint_matrices = j
    return j


class _JointDevice:
    """the joint-matrix vectors of an instance resident in HBM: a function of the matrices only (what `index` commits to,
    src/psnark/time_prover.rs:49-64), so they are uploaded ONCE per R1cs object and live with it -- like its CSR matrices,
    they are part of the instance that is resident before the prover starts.  (The reference recomputes joint_matrices on the
    CPU inside new_time, :99-110; re-uploading 5 vectors per proof kept the GPU idle for 4 % of psnark -i 22.)"""

    def __init__(self, r1cs: R1cs):
        row_index_h, col_index_h, val_a_h, val_b_h, val_c_h = _joint(r1cs)
        self.row_index_h, self.col_index_h = row_index_h, col_index_h
        self.row_index, self.col_index = IdxVec.from_host(row_index_h), IdxVec.from_host(col_index_h)
        self.row, self.col = _field_of_index(self.row_index), _field_of_index(self.col_index)
        self.val_a, self.val_b, self.val_c = FrVec.from_host(val_a_h), FrVec.from_host(val_b_h), FrVec.from_host(val_c_h)
        self.ext_fre = {}

    def extended_frequencies(self, len_r: int, len_z: int):
        """[extend_frequency(compute_frequency(..)) for the row and the column index] as device index vectors
        (plookup/time_prover.rs:66-79; src/psnark/time_prover.rs:165-178): instance-only as well -- the bincount / repeat over
        the index arrays kept the GPU idle for 44 ms of a 480 ms proof at 2^22 when done per proof"""
        key = (len_r, len_z)
        if key not in self.ext_fre:
            self.ext_fre[key] = [IdxVec.from_host(extend_frequency(compute_frequency(len_r, self.row_index_h))),
                                 IdxVec.from_host(extend_frequency(compute_frequency(len_z, self.col_index_h)))]
        return self.ext_fre[key]

    def free(self):
        for v in (self.row_index, self.col_index, self.row, self.col, self.val_a, self.val_b, self.val_c):
            v.free()
        for pair in self.ext_fre.values():
            for v in pair:
                v.free()
        self.ext_fre = {}


def _joint_device(r1cs: R1cs) -> "_JointDevice":
    cache = r1cs.__dict__.setdefault("_device_cache", {})
    if "joint" not in cache:
        cache["joint"] = _JointDevice(r1cs)
    return cache["joint"]


class _JointNative:
    """the same matrix-only part of the instance built INSIDE the library (gm_psnark_preprocess: joint support, value vectors,
    row / col, extended frequencies; gemini_amd/csrc/psnark.cpp) -- what a Rust / C++ integrator calls instead of restating
    src/misc.rs:269-366 in the shim.  Kept with the R1cs object like _JointDevice and freed with it."""

    def __init__(self, r1cs: R1cs):
        import ctypes as C

        from . import capi

        Instance, _ = _psnark_ctypes()
        self.rec = Instance()
        capi.check(capi.load().gm_psnark_preprocess(C.c_uint64(r1cs.a.handle), C.c_uint64(r1cs.b.handle), C.c_uint64(r1cs.c.handle),
                                                    C.c_size_t(len(r1cs.z)), C.byref(self.rec)))

    def index(self, ck: CommitterKey) -> list:
        import ctypes as C

        from . import capi

        out = np.zeros((5, 18), dtype=np.uint64)
        capi.check(capi.load().gm_psnark_index(C.byref(self.rec), C.c_uint64(ck.powers_of_g.handle), capi.ptr(out)))
        return [out[k].copy() for k in range(5)]

    def free(self):
        import ctypes as C

        from . import capi

        capi.check(capi.load().gm_psnark_preprocess_free(C.byref(self.rec)))


def _joint_native(r1cs: R1cs) -> "_JointNative":
    cache = r1cs.__dict__.setdefault("_device_cache", {})
    if "joint_native" not in cache:
        cache["joint_native"] = _JointNative(r1cs)
    return cache["joint_native"]


def _field_of_index(index: IdxVec) -> FrVec:
    """[F::from(i) for i in index] (the `row` / `col` vectors, src/misc.rs:343-349)"""
    zeros = FrVec.alloc(len(index))
    zeros.fill(fr_from_int(0))
    out = alg_hash(zeros, index, _ONE)
    zeros.free()
    return out


# ---- src/subprotocols/plookup/time_prover.rs ------------------------------------------------------------
def compute_frequency(set_len: int, index: np.ndarray) -> np.ndarray:
    """:66-70"""
    return 1 + np.bincount(index, minlength=set_len).astype(np.int64)


def extend_frequency(frequency: np.ndarray) -> np.ndarray:
    """:72-79"""
    return np.repeat(np.arange(len(frequency), dtype=np.uint32), frequency)


def plookup(subset: FrVec, set_: FrVec, index: IdxVec, ext_fre: IdxVec, y, z, zeta):
    """:89-112 -> [lookup_set, lookup_subset, lookup_sorted]; ext_fre = extend_frequency(compute_frequency(..))
    so that sorted(set, frequency) = [set[i] for i in ext_fre]"""
    tmp = []
    if fr_to_int(zeta) != 0:
        set_h = alg_hash(set_, None, zeta)
        subset_h = alg_hash(subset, index, zeta)
        tmp += [set_h, subset_h]
    else:
        set_h, subset_h = set_, subset
    lookup_set = plookup_set(set_h, y, z)
    lookup_subset = plookup_subset(subset_h, y)
    srt = lookup(set_h, ext_fre)
    lookup_sorted = plookup_set(srt, y, z)
    for v in tmp + [srt]:
        v.free()
    return [lookup_set, lookup_subset, lookup_sorted]


# ---- src/subprotocols/entryproduct -----------------------------------------------------------------------
class EntryProductMsgs:
    """entryproduct/mod.rs:21-25"""

    def __init__(self, acc_v_commitments, claimed_sumchecks):
        self.acc_v_commitments = acc_v_commitments
        self.claimed_sumchecks = claimed_sumchecks


class EntryProduct:
    """entryproduct/mod.rs:27-31"""

    def __init__(self, msgs, chal, provers):
        self.msgs, self.chal, self.provers = msgs, chal, provers

    @staticmethod
    def new_time_batch(transcript, ck, vs, claimed_products, acc_vs=None) -> "EntryProduct":
        """entryproduct/time_prover.rs:53-114.  acc_vs: accumulated_product(monic(v)) when the caller has
        them already (the psnark prover does)."""
        assert len(vs) == len(claimed_products)
        own = acc_vs is None
        if own:
            acc_vs = [accumulated_product_monic(v) for v in vs]
        rrot_vs = [shift_monic(v) for v in vs]
        acc_v_commitments = ck.batch_commit(acc_vs)
        for c in acc_v_commitments:
            transcript.append_g1(b"acc_v", c)
        chal = transcript.get_challenge(b"ep-chal")
        ci = fr_to_int(chal)
        provers = [TimeProver(acc_v, rrot_v, chal) for rrot_v, acc_v in zip(rrot_vs, acc_vs)]
        claimed_sumchecks = []
        acc_v_chals = evaluate_le_batch(list(acc_vs), chal.reshape(1, 4))
        for cp, acc_v, av in zip(claimed_products, acc_vs, acc_v_chals):
            acc_v_chal = fr_to_int(av[0])
            chal_n = pow(ci, len(acc_v), R_MOD)
            claimed_sumchecks.append(fr_from_int((acc_v_chal * ci + fr_to_int(cp) - chal_n) % R_MOD))
        for v in rrot_vs + (acc_vs if own else []):
            v.free()
        return EntryProduct(EntryProductMsgs(acc_v_commitments, claimed_sumchecks), chal, provers)

    @staticmethod
    def new_time(transcript, ck, v, claimed_product) -> "EntryProduct":
        """entryproduct/time_prover.rs:116-147"""
        return EntryProduct.new_time_batch(transcript, ck, [v], [claimed_product])

    @staticmethod
    def new_elastic(transcript, ck_stream, v_stream, claimed_product) -> "EntryProduct":
        """entryproduct/elastic_prover.rs:32-63 over a big-endian device stream: ProductStream /
        RightRotationStreamer are the reversed accumulated-product / shifted vectors"""
        from .fr import reverse
        from .snark import _evaluate_be
        from .sumcheck import ElasticProver

        v = reverse(v_stream)
        acc_s, rrot_s = reverse(accumulated_product_monic(v)), reverse(shift_monic(v))
        cm = ck_stream.commit(acc_s)
        transcript.append_g1(b"acc_v", cm)
        chal = transcript.get_challenge(b"ep-chal")
        ci = fr_to_int(chal)
        claimed = fr_from_int((ci * fr_to_int(_evaluate_be(acc_s, chal.reshape(1, 4))[0]) + fr_to_int(claimed_product) - pow(ci, len(acc_s), R_MOD)) % R_MOD)
        provers = [ElasticProver(acc_s, rrot_s, chal)]
        for x in (v, acc_s, rrot_s):
            x.free()
        return EntryProduct(EntryProductMsgs([cm], [claimed]), chal, provers)


class Proof:
    """src/psnark/mod.rs:29-51"""

    FIELDS = ("witness_commitment", "zc_alpha", "first_sumcheck_msgs", "r_star_commitments", "z_star_commitment", "second_sumcheck_msgs",
              "set_r_ep", "subset_r_ep", "sorted_r_commitment", "set_alpha_ep", "subset_alpha_ep", "sorted_alpha_commitment", "set_z_ep",
              "subset_z_ep", "sorted_z_commitment", "ep_msgs", "ralpha_star_acc_mu_evals", "ralpha_star_acc_mu_proof", "rstars_vals",
              "third_sumcheck_msgs", "tensorcheck_proof")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])
        self.spans = {}

    @staticmethod
    def index(ck: CommitterKey, r1cs: R1cs, native: bool = False) -> list:
        """src/psnark/time_prover.rs:49-64.  native: joint matrices and commitments inside the library (gm_psnark_preprocess +
        gm_psnark_index)"""
        if native:
            return _joint_native(r1cs).index(ck)
        jd = _joint_device(r1cs)
        return ck.batch_commit([jd.row, jd.col, jd.val_a, jd.val_b, jd.val_c])

    @staticmethod
    def new_time(ck: CommitterKey, r1cs: R1cs, index: list, native: bool = False) -> "Proof":
        """src/psnark/time_prover.rs:69-384.  native: the same sequence compiled into the library (gm_psnark_new_time, one call
        per proof) -- for a single-GPU CommitterKey; native = "preprocess": the matrix-only part of the instance record comes from
        gm_psnark_preprocess as well (nothing of src/misc.rs:269-366 is left to the caller)."""
        if native and type(ck) is CommitterKey:
            return _new_time_native(ck, r1cs, index, preprocess_in_library=native == "preprocess")
        spans = {}
        keep = []  # device vectors freed at the end

        def K(v):
            keep.append(v)
            return v

        t_all = time.perf_counter()
        z_a = K(r1cs.a.mul(r1cs.z))  # :74-76
        z_b = K(r1cs.b.mul(r1cs.z))
        z_c = K(r1cs.c.mul(r1cs.z))
        transcript = Transcript(PROTOCOL_NAME)
        t0 = time.perf_counter()
        witness_commitment = ck.commit(r1cs.w)  # :79
        spans["Commitment to w"] = time.perf_counter() - t0

        transcript.append_g1(b"witness", witness_commitment)  # :82-86
        transcript.append_message(b"ck", ck.powers_of_g2_bytes())
        transcript.append_g1(b"instance", np.stack(index), with_len=True)
        alpha = transcript.get_challenge(b"alpha")

        zc_alpha = evaluate_le(z_c, alpha.reshape(1, 4))[0]  # :88-89
        transcript.append_fr(b"zc(alpha)", zc_alpha)

        t0 = time.perf_counter()
        first_proof = Sumcheck.new_time(transcript, z_a, z_b, alpha)  # :92
        spans["First sumcheck"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        b_challenges = K(tensor(np.stack(first_proof.challenges)))  # :95-97
        c_challenges = K(powers(alpha, len(b_challenges)))
        a_challenges = K(hadamard(b_challenges, c_challenges))

        jd = _joint_device(r1cs)  # :99-110, resident with the instance
        row_index, col_index, row, col = jd.row_index, jd.col_index, jd.row, jd.col
        val_a, val_b, val_c = jd.val_a, jd.val_b, jd.val_c
        num_non_zero = len(row_index)
        spans["joint matrices"] = time.perf_counter() - t0

        ralpha_star = K(lookup(a_challenges, row_index))  # :114-117
        r_star = K(lookup(b_challenges, row_index))
        alpha_star = K(lookup(c_challenges, row_index))
        z_star = K(lookup(r1cs.z, col_index))

        # :119-127.  ck.index_by(row_index).commit(a_challenges) = sum_j a_challenges[row_index[j]] * g_j, the
        # commitment to the looked-up vector under ck itself (the reference's commented-out line :126): one
        # MSM over the resident key instead of building an indexed key.  The index_by zip needs as many
        # powers as indices.
        assert ck.num_powers() >= num_non_zero, "committer key shorter than the number of non-zero entries"
        t0 = time.perf_counter()
        z_r_commitments = ck.batch_commit([ralpha_star, r_star, alpha_star]) + [ck.commit(z_star)]
        spans["Commitments to z* and r*"] = time.perf_counter() - t0

        transcript.append_g1(b"ra*", z_r_commitments[0])  # :129-132
        transcript.append_g1(b"rb*", z_r_commitments[1])
        transcript.append_g1(b"rc*", z_r_commitments[2])
        transcript.append_g1(b"z*", z_r_commitments[3])

        eta = transcript.get_challenge(b"chal")  # :134-135
        eta_i = fr_to_int(eta)
        challenges = np.stack([_ONE, eta, fr_from_int(eta_i * eta_i % R_MOD)])

        h_a, h_b, h_c = hadamard(ralpha_star, val_a), hadamard(r_star, val_b), hadamard(alpha_star, val_c)
        r_star_val = K(linear_combination([h_a, h_b, h_c], challenges))  # :137-144
        for v in (h_a, h_b, h_c):
            v.free()

        t0 = time.perf_counter()
        second_proof = Sumcheck.new_time(transcript, z_star, r_star_val, _ONE)  # :147-152
        second_challenges = K(tensor(np.stack(second_proof.challenges)))
        assert len(second_challenges) >= num_non_zero
        second_challenges_head = second_challenges
        second_challenges_head.set_len(num_non_zero)  # &second_challenges[..num_non_zero]
        spans["Second sumcheck"] = time.perf_counter() - t0

        zeta = transcript.get_challenge(b"zeta")  # :157

        t0 = time.perf_counter()
        alg_hash_poly = [K(alg_hash(b_challenges, None, zeta)), K(alg_hash(c_challenges, None, zeta)), K(alg_hash(r1cs.z, None, zeta))]  # :160-164
        ext_fre = jd.extended_frequencies(len(alg_hash_poly[0]), len(alg_hash_poly[2]))  # :165-168, :175-178
        sorted_polynomials = [K(lookup(alg_hash_poly[0], ext_fre[0])), K(lookup(alg_hash_poly[1], ext_fre[0])),
                              K(lookup(alg_hash_poly[2], ext_fre[1]))]  # :169-173
        # :179-183: ck.index_by(ext_fre).commit(alg_hash_poly) = commitment to the sorted vector under ck (:183)
        assert ck.num_powers() >= max(len(ext_fre[0]), len(ext_fre[1])), "committer key shorter than the sorted vectors"
        sorted_commitments = ck.batch_commit(sorted_polynomials)
        spans["Commitments to sorted vectors"] = time.perf_counter() - t0

        transcript.append_g1(b"sorted_alpha_commitment", sorted_commitments[1])  # :186-188
        transcript.append_g1(b"sorted_r_commitment", sorted_commitments[0])
        transcript.append_g1(b"sorted_z_commitment", sorted_commitments[2])

        gamma = transcript.get_challenge(b"gamma")  # :190-191
        chi = transcript.get_challenge(b"chi")

        t0 = time.perf_counter()
        r_lookup_vec = [K(v) for v in plookup(r_star, b_challenges, row_index, ext_fre[0], gamma, chi, zeta)]  # :194-204
        alpha_lookup_vec = [K(v) for v in plookup(alpha_star, c_challenges, row_index, ext_fre[0], gamma, chi, zeta)]
        z_lookup_vec = [K(v) for v in plookup(z_star, r1cs.z, col_index, ext_fre[1], gamma, chi, zeta)]
        lookup_vec = r_lookup_vec + alpha_lookup_vec + z_lookup_vec  # :206-209
        accumulated_vec = [K(accumulated_product_monic(v)) for v in lookup_vec]  # accproduct3, :211-214
        prod = [element(acc, 0) for acc in accumulated_vec]  # product3: the full product is the first accumulated entry
        r_prod_vec, alpha_prod_vec, z_prod_vec = prod[0:3], prod[3:6], prod[6:9]
        spans["plookup vectors + accumulated products"] = time.perf_counter() - t0

        transcript.append_fr(b"set_r_ep", alpha_prod_vec[0])  # :216-221 (labels as in the reference)
        transcript.append_fr(b"subset_r_ep", alpha_prod_vec[1])
        transcript.append_fr(b"set_r_ep", r_prod_vec[0])
        transcript.append_fr(b"subset_r_ep", r_prod_vec[1])
        transcript.append_fr(b"set_z_ep", z_prod_vec[0])
        transcript.append_fr(b"subset_z_ep", z_prod_vec[1])

        t0 = time.perf_counter()
        entry_products = EntryProduct.new_time_batch(transcript, ck, lookup_vec, prod, acc_vs=accumulated_vec)  # :223-239
        spans["Entry products"] = time.perf_counter() - t0

        psi = entry_products.chal  # :241-242
        open_chal = transcript.get_challenge(b"open-chal")

        t0 = time.perf_counter()
        polynomials = [ralpha_star] + accumulated_vec  # :244-251
        ralpha_star_acc_mu_proof = ck.batch_open_multi_points(polynomials, psi.reshape(1, 4), open_chal)
        ralpha_star_acc_mu_evals = [e[0] for e in evaluate_le_batch(polynomials, psi.reshape(1, 4))]
        spans["Opening at psi"] = time.perf_counter() - t0

        h_a, h_b = hadamard(ralpha_star, val_a), hadamard(r_star, val_b)  # :253-254
        s_0_prime, s_1_prime = ip(h_a, second_challenges_head), ip(h_b, second_challenges_head)
        h_a.free()
        h_b.free()
        for e in ralpha_star_acc_mu_evals:  # :258-261
            transcript.append_fr(b"ralpha_star_acc_mu", e)
        transcript.append_g1(b"ralpha_star_mu_proof", ralpha_star_acc_mu_proof)

        provers = list(entry_products.provers)  # :263-290
        for lhs, rhs in ((ralpha_star, val_a), (r_star, val_b), (alpha_star, val_c)):
            h = hadamard(lhs, second_challenges_head)
            provers.append(TimeProver(h, rhs, _ONE))
            h.free()
        provers.append(TimeProver(r_star, alpha_star, psi))

        t0 = time.perf_counter()
        third_proof = Sumcheck.prove_batch(transcript, provers)  # :293
        for p in provers:
            p.free()
        spans["Third sumcheck"] = time.perf_counter() - t0

        tc_base_polynomials = [r1cs.w, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c] + sorted_polynomials + accumulated_vec  # :296-319

        third_ch = [fr_to_int(c) for c in third_proof.challenges]
        second_ch = [fr_to_int(c) for c in second_proof.challenges]
        twist_powers2 = [pow(fr_to_int(psi), 1 << j, R_MOD) for j in range(len(third_ch))]  # :321

        shift_monic_lookup_vec = [K(shift_monic(v)) for v in lookup_vec]  # :323-326
        third_proof_vec = shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star]  # :329-330
        body_polynomials_0 = accumulated_vec + [r_star]  # :334-345
        head = third_ch[: len(second_ch)]  # :346
        F = lambda ints: [fr_from_int(v) for v in ints]
        tc_body_polynomials = [  # :347-359
            (body_polynomials_0, F([a * b % R_MOD for a, b in zip(third_ch, twist_powers2)])),
            (third_proof_vec, F(third_ch)),
            ([z_star], F(second_ch)),
            ([ralpha_star, r_star, alpha_star], F([a * b % R_MOD for a, b in zip(second_ch, head)])),
        ]

        t0 = time.perf_counter()
        tensorcheck_proof = TensorcheckProof.new_time(transcript, ck, tc_base_polynomials, tc_body_polynomials)  # :362-367
        spans["Tensorcheck"] = time.perf_counter() - t0

        for v in keep:
            v.free()
        transcript.free()
        spans["ark_gemini::psnark::time_prover"] = time.perf_counter() - t_all
        proof = Proof(
            witness_commitment=witness_commitment, zc_alpha=zc_alpha,
            first_sumcheck_msgs=(first_proof.messages, first_proof.final_foldings),
            r_star_commitments=z_r_commitments[:3], z_star_commitment=z_r_commitments[3],
            second_sumcheck_msgs=(second_proof.messages, second_proof.final_foldings),
            set_r_ep=r_prod_vec[0], subset_r_ep=r_prod_vec[1], sorted_r_commitment=sorted_commitments[0],
            set_alpha_ep=alpha_prod_vec[0], subset_alpha_ep=alpha_prod_vec[1], sorted_alpha_commitment=sorted_commitments[1],
            set_z_ep=z_prod_vec[0], subset_z_ep=z_prod_vec[1], sorted_z_commitment=sorted_commitments[2],
            ep_msgs=entry_products.msgs, ralpha_star_acc_mu_evals=ralpha_star_acc_mu_evals,
            ralpha_star_acc_mu_proof=ralpha_star_acc_mu_proof, rstars_vals=[s_0_prime, s_1_prime],
            third_sumcheck_msgs=(third_proof.messages, third_proof.final_foldings), tensorcheck_proof=tensorcheck_proof)
        proof.spans = spans
        return proof

    @staticmethod
    def new_elastic(ck, r1cs_stream, index: list, max_msm_buffer: int, native: bool = False) -> "Proof":
        """src/psnark/elastic_prover.rs:60-634 over device-resident streams: `ck` is a CommitterKeyStream,
        every polynomial a big-endian stream (reversed device vector); commitments are chunked stream MSMs,
        sumchecks run on the space / elastic provers, the tensor check on FoldedPolynomialTrees.  The
        reference's test asserts this proof equals new_time's (src/psnark/tests.rs:56-124).  native: the same sequence compiled
        into the library (gm_psnark_new_elastic, gemini_amd/csrc/psnark_elastic.cpp), one call per proof."""
        from .kzg import CommitterKeyStream

        if native and type(ck) is CommitterKeyStream:
            return _new_time_native(ck, r1cs_stream.r1cs, index, elastic=(ck, r1cs_stream, max_msm_buffer))
        from .fr import fold_polynomial, reverse
        from .kzg import FoldedPolynomialTree
        from .msm import g1_sum
        from .snark import _evaluate_be
        from .sumcheck import ElasticProver

        spans = {}
        keep = []

        def K(v):
            keep.append(v)
            return v

        S = lambda v: K(reverse(v))  # little-endian vector -> big-endian stream
        t_all = time.perf_counter()
        r1cs = r1cs_stream.r1cs
        transcript = Transcript(PROTOCOL_NAME)
        witness_commitment = ck.commit(r1cs_stream.witness)  # :82
        transcript.append_g1(b"witness", witness_commitment)  # :86-89
        transcript.append_message(b"ck", ck.powers_of_g2_bytes())
        transcript.append_g1(b"instance", np.stack(index), with_len=True)
        alpha = transcript.get_challenge(b"alpha")
        zc_alpha = _evaluate_be(r1cs_stream.z_c, alpha.reshape(1, 4))[0]  # :92-93
        transcript.append_fr(b"zc(alpha)", zc_alpha)
        t0 = time.perf_counter()
        sumcheck1 = Sumcheck.new_space(transcript, r1cs_stream.z_a, r1cs_stream.z_b, alpha)  # :97
        spans["sumcheck1"] = time.perf_counter() - t0

        # the Joint{Row,Col,Val} streams (:100-146) walk the joint support; here its index / value vectors
        jd = _joint_device(r1cs)
        row_index, col_index, row, col = jd.row_index, jd.col_index, jd.row, jd.col
        val_a, val_b, val_c = jd.val_a, jd.val_b, jd.val_c
        num_non_zero = len(row_index)
        z_le = K(reverse(r1cs_stream.z))
        w_le = K(reverse(r1cs_stream.witness))
        z_star = K(lookup(z_le, col_index))  # :148
        rs = K(tensor(np.stack(sumcheck1.challenges)))  # Tensor(r_short)                         :150-157
        alphas = K(powers(alpha, len(rs)))  # Tensor(powers2(alpha)) = powers of alpha
        ralphas = K(hadamard(rs, alphas))
        ralpha_star, r_star, alpha_star = K(lookup(ralphas, row_index)), K(lookup(rs, row_index)), K(lookup(alphas, row_index))  # :159-161

        t0 = time.perf_counter()
        r_star_commitments = [ck.commit(S(ralpha_star)), ck.commit(S(r_star)), ck.commit(S(alpha_star))]  # :164-172
        z_star_commitment = ck.commit(S(z_star))
        spans["Commitments to z* and r*"] = time.perf_counter() - t0
        transcript.append_g1(b"ra*", r_star_commitments[0])
        transcript.append_g1(b"rb*", r_star_commitments[1])
        transcript.append_g1(b"rc*", r_star_commitments[2])
        transcript.append_g1(b"z*", z_star_commitment)

        challenge = transcript.get_challenge(b"chal")  # :181-192
        ci = fr_to_int(challenge)
        h_a, h_b, h_c = hadamard(ralpha_star, val_a), hadamard(r_star, val_b), hadamard(alpha_star, val_c)
        rhs = K(linear_combination([h_a, h_b, h_c], np.stack([_ONE, challenge, fr_from_int(ci * ci % R_MOD)])))
        for v in (h_a, h_b, h_c):
            v.free()
        t0 = time.perf_counter()
        sumcheck2 = Sumcheck.new_elastic(transcript, S(z_star), S(rhs), _ONE)  # :195
        spans["sumcheck2"] = time.perf_counter() - t0

        zeta = transcript.get_challenge(b"zeta")  # :199
        hashed_r, hashed_alpha, hashed_z = K(alg_hash(rs, None, zeta)), K(alg_hash(alphas, None, zeta)), K(alg_hash(z_le, None, zeta))  # :205-210
        ext_fre = jd.extended_frequencies(len(rs), len(z_le))
        sorted_r, sorted_alpha, sorted_z = K(lookup(hashed_r, ext_fre[0])), K(lookup(hashed_alpha, ext_fre[0])), K(lookup(hashed_z, ext_fre[1]))  # :212-214
        t0 = time.perf_counter()
        sorted_r_commitment, sorted_alpha_commitment, sorted_z_commitment = ck.commit(S(sorted_r)), ck.commit(S(sorted_alpha)), ck.commit(S(sorted_z))
        spans["Commitments to sorted vectors"] = time.perf_counter() - t0
        transcript.append_g1(b"sorted_alpha_commitment", sorted_alpha_commitment)  # :220-222
        transcript.append_g1(b"sorted_r_commitment", sorted_r_commitment)
        transcript.append_g1(b"sorted_z_commitment", sorted_z_commitment)
        gamma = transcript.get_challenge(b"gamma")
        chi = transcript.get_challenge(b"chi")

        pl_r = [K(v) for v in plookup(r_star, rs, row_index, ext_fre[0], gamma, chi, zeta)]  # plookup_streams, :227-232
        pl_alpha = [K(v) for v in plookup(alpha_star, alphas, row_index, ext_fre[0], gamma, chi, zeta)]
        pl_z = [K(v) for v in plookup(z_star, z_le, col_index, ext_fre[1], gamma, chi, zeta)]
        pls = pl_r + pl_alpha + pl_z
        accs = [K(accumulated_product_monic(v)) for v in pls]  # ProductStream
        shifts = [K(shift_monic(v)) for v in pls]  # RightRotationStreamer
        prod = [element(a, 0) for a in accs]  # :235-243
        transcript.append_fr(b"set_r_ep", prod[3])  # :245-250
        transcript.append_fr(b"subset_r_ep", prod[4])
        transcript.append_fr(b"set_r_ep", prod[0])
        transcript.append_fr(b"subset_r_ep", prod[1])
        transcript.append_fr(b"set_z_ep", prod[6])
        transcript.append_fr(b"subset_z_ep", prod[7])

        assert len(K(tensor(np.stack(sumcheck2.challenges)))) >= num_non_zero
        ep_r = keep[-1]  # Tensor(&sumcheck2.challenges), cut to the looked-up length             :254-257
        ep_r.set_len(num_non_zero)

        # EntryProduct::new_elastic_batch (entryproduct/elastic_prover.rs:66-127)
        t0 = time.perf_counter()
        acc_streams = [S(a) for a in accs]
        acc_v_commitments = []
        for a in acc_streams:
            cm = ck.commit(a)
            transcript.append_g1(b"acc_v", cm)
            acc_v_commitments.append(cm)
        psi = transcript.get_challenge(b"ep-chal")
        pi = fr_to_int(psi)
        claimed_sumchecks, provers = [], []
        for cp, a, a_s, sh in zip(prod, accs, acc_streams, shifts):
            acc_v_chal = fr_to_int(_evaluate_be(a_s, psi.reshape(1, 4))[0])
            claimed_sumchecks.append(fr_from_int((acc_v_chal * pi + fr_to_int(cp) - pow(pi, len(a), R_MOD)) % R_MOD))
            provers.append(ElasticProver(a_s, S(sh), psi))
        msgs = EntryProductMsgs(acc_v_commitments, claimed_sumchecks)
        spans["Entry products"] = time.perf_counter() - t0

        open_chal = transcript.get_challenge(b"open-chal")  # :313-330
        oc10 = powers(open_chal, 10)
        polynomial = K(linear_combination([ralpha_star] + accs, oc10.to_host()))
        oc10.free()
        ralpha_star_acc_mu_proof = ck.open(S(polynomial), psi, max_msm_buffer)[1]
        ralpha_star_acc_mu_evals = [e[0] for e in evaluate_le_batch([ralpha_star] + accs, psi.reshape(1, 4))]  # :332-343
        lhs = [K(hadamard(v, ep_r)) for v in (ralpha_star, r_star, alpha_star)]
        r_val_chal_a, r_val_chal_b = ip(lhs[0], val_a), ip(lhs[1], val_b)  # :348-349
        for e in ralpha_star_acc_mu_evals:
            transcript.append_fr(b"ralpha_star_acc_mu", e)
        transcript.append_g1(b"ralpha_star_mu_proof", ralpha_star_acc_mu_proof)
        for l, v in zip(lhs, (val_a, val_b, val_c)):  # :358-377
            provers.append(ElasticProver(S(l), S(v), _ONE))
        provers.append(ElasticProver(S(r_star), S(alpha_star), psi))
        t0 = time.perf_counter()
        sumcheck3 = Sumcheck.prove_batch_generic(transcript, provers)  # :380
        for p in provers:
            p.free()
        spans["sumcheck3"] = time.perf_counter() - t0

        # tensorcheck (:384-600)
        t0 = time.perf_counter()
        tc_chal = transcript.get_challenge(b"batch_challenge")
        tcc_v = powers(tc_chal, 13)
        tcc = tcc_v.to_host()
        tcc_v.free()
        bodies = [K(linear_combination(accs + [r_star], tcc)), K(linear_combination(shifts + [val_a, val_b, val_c, alpha_star], tcc)), z_star,
                  K(linear_combination([ralpha_star, r_star, alpha_star], tcc))]
        ch2 = [fr_to_int(c) for c in sumcheck2.challenges]
        ch3 = [fr_to_int(c) for c in sumcheck3.challenges]
        psi_squares = [pow(pi, 1 << j, R_MOD) for j in range(len(ch3))]
        F = lambda ints: [fr_from_int(v) for v in ints]
        tc_challenges = [F([a * b % R_MOD for a, b in zip(ch3, psi_squares)][:-1]), F(ch3[:-1]), F(ch2[:-1]),
                         F([a * b % R_MOD for a, b in zip(ch2, ch3[: len(ch2)])][:-1])]
        trees = [FoldedPolynomialTree(S(b), c) for b, c in zip(bodies, tc_challenges)]
        folded_polynomials_commitments = []
        for t in trees:
            folded_polynomials_commitments.extend(ck.commit_folding(t, max_msm_buffer))
        for c in folded_polynomials_commitments:
            transcript.append_g1(b"commitment", c)
        eval_chal = transcript.get_challenge(b"evaluation-chal")
        ec = fr_to_int(eval_chal)
        pts = np.stack([fr_from_int(ec * ec % R_MOD), eval_chal, fr_from_int((-ec) % R_MOD)])
        folded_polynomials_evaluations = []  # evaluate_folding at +-eval_chal, tree by tree
        for b, chs in zip(bodies, tc_challenges):
            cur = b
            for ch in chs:
                nxt = fold_polynomial(cur, ch)
                if cur is not b:
                    cur.free()
                cur = nxt
                folded_polynomials_evaluations.append(evaluate_le(cur, pts[1:]))
            if cur is not b:
                cur.free()
        base = [w_le, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c, sorted_r, sorted_alpha, sorted_z] + accs
        base_polynomials_evaluations = []
        for p in base:  # evaluate_base_polynomial appends as it goes (:36-57)
            e3 = evaluate_le(p, pts)
            for e in e3:
                transcript.append_fr(b"eval", e)
            base_polynomials_evaluations.append(e3)
        for e2 in folded_polynomials_evaluations:
            for e in e2:
                transcript.append_fr(b"eval", e)
        open_chal = transcript.get_challenge(b"open-chal")
        open_chal_len = len(folded_polynomials_evaluations) * trees[2].depth() + 3 * len(base)
        ocv = powers(open_chal, max(open_chal_len, len(base) + len(folded_polynomials_evaluations)))
        oc = ocv.to_host()
        ocv.free()
        partial_eval = K(linear_combination(base, oc[: len(base)]))
        parts = [ck.open_multi_points(S(partial_eval), pts, max_msm_buffer)[1]]
        off = len(base)
        for t in trees:
            parts.append(ck.open_folding(t, pts, oc[off: off + t.depth()], max_msm_buffer)[1])
            off += t.depth()
        evaluation_proof = g1_sum(np.stack(parts))
        tensorcheck_proof = TensorcheckProof(folded_polynomials_commitments, folded_polynomials_evaluations, evaluation_proof, base_polynomials_evaluations)
        spans["tensorcheck"] = time.perf_counter() - t0

        for v in keep:
            v.free()
        transcript.free()
        spans["ark_gemini::psnark::elastic_prover"] = time.perf_counter() - t_all
        proof = Proof(
            witness_commitment=witness_commitment, zc_alpha=zc_alpha,
            first_sumcheck_msgs=(sumcheck1.messages, sumcheck1.final_foldings),
            r_star_commitments=r_star_commitments, z_star_commitment=z_star_commitment,
            second_sumcheck_msgs=(sumcheck2.messages, sumcheck2.final_foldings),
            set_r_ep=prod[0], subset_r_ep=prod[1], sorted_r_commitment=sorted_r_commitment,
            set_alpha_ep=prod[3], subset_alpha_ep=prod[4], sorted_alpha_commitment=sorted_alpha_commitment,
            set_z_ep=prod[6], subset_z_ep=prod[7], sorted_z_commitment=sorted_z_commitment,
            ep_msgs=msgs, ralpha_star_acc_mu_evals=ralpha_star_acc_mu_evals, ralpha_star_acc_mu_proof=ralpha_star_acc_mu_proof,
            rstars_vals=[r_val_chal_a, r_val_chal_b], third_sumcheck_msgs=(sumcheck3.messages, sumcheck3.final_foldings),
            tensorcheck_proof=tensorcheck_proof)
        proof.spans = spans
        return proof

    def serialize(self, compress: bool = True, enc=0) -> bytes:
        """derive(CanonicalSerialize) of src/psnark/mod.rs:29-51 (formats: gemini_amd/wire.py)"""
        from . import wire

        return wire.serialize(wire.PSNARK_PROOF, self, compress, enc)

    def serialize_compressed(self, enc=0) -> bytes:
        return self.serialize(True, enc)

    def serialize_uncompressed(self, enc=0) -> bytes:
        return self.serialize(False, enc)

    @staticmethod
    def deserialize(data: bytes, compress: bool = True, enc=0, validate: bool = True) -> "Proof":
        from . import wire

        return wire.deserialize(wire.PSNARK_PROOF, data, compress, enc, validate)

    def __eq__(self, other) -> bool:
        """derive(PartialEq, Eq)"""
        from . import wire

        return isinstance(other, Proof) and wire.equal(wire.PSNARK_PROOF, self, other)

    __hash__ = None

    def compressed_size(self) -> int:
        return len(self.serialize_compressed())


# ---- gm_psnark_new_time: the orchestration above compiled into the library (gemini_amd/csrc/psnark.cpp) -------------------
_PSNARK_SPANS = ["Commitment to w", "First sumcheck", "joint matrices", "Commitments to z* and r*", "Second sumcheck",
                 "Commitments to sorted vectors", "plookup vectors + accumulated products", "Entry products", "Opening at psi", "Third sumcheck",
                 "Tensorcheck", "ark_gemini::psnark::time_prover"]


def _psnark_ctypes():
    import ctypes as C

    U64P = C.POINTER(C.c_uint64)

    class Instance(C.Structure):
        _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("z", C.c_uint64), ("w", C.c_uint64), ("row_index", C.c_uint64),
                    ("col_index", C.c_uint64), ("nnz", C.c_size_t), ("row", C.c_uint64), ("col", C.c_uint64), ("val_a", C.c_uint64),
                    ("val_b", C.c_uint64), ("val_c", C.c_uint64), ("ext_fre_row", C.c_uint64), ("ext_fre_col", C.c_uint64),
                    ("ext_fre_row_len", C.c_size_t), ("ext_fre_col_len", C.c_size_t), ("index_commitments", U64P),
                    ("ck_g2_bytes", C.POINTER(C.c_uint8)), ("ck_g2_len", C.c_size_t)]

    class ProofRec(C.Structure):
        _fields_ = [("witness_commitment", C.c_uint64 * 18), ("zc_alpha", C.c_uint64 * 4), ("rounds", C.c_size_t * 3), ("messages", U64P * 3),
                    ("final_foldings", (C.c_uint64 * 8) * 2), ("third_final_foldings", (C.c_uint64 * 8) * 13),
                    ("r_star_commitments", (C.c_uint64 * 18) * 3), ("z_star_commitment", C.c_uint64 * 18),
                    ("sorted_commitments", (C.c_uint64 * 18) * 3), ("products", (C.c_uint64 * 4) * 9),
                    ("acc_v_commitments", (C.c_uint64 * 18) * 9), ("claimed_sumchecks", (C.c_uint64 * 4) * 9),
                    ("ralpha_star_acc_mu_evals", (C.c_uint64 * 4) * 10), ("ralpha_star_acc_mu_proof", C.c_uint64 * 18),
                    ("rstars_vals", (C.c_uint64 * 4) * 2), ("nfold", C.c_size_t), ("cap_folds", C.c_size_t), ("fold_commitments", U64P),
                    ("fold_evaluations", U64P), ("evaluation_proof", C.c_uint64 * 18), ("base_evaluations", (C.c_uint64 * 12) * 22),
                    ("spans", C.c_double * 12)]

    return Instance, ProofRec


def _proof_buffers(num_variables: int, nnz: int):
    """a gm_psnark_proof record with its caller-owned arrays: (record, cap_rounds, (messages, fold commitments, fold evaluations))"""
    import ctypes as C

    _, ProofRec = _psnark_ctypes()
    U = C.POINTER(C.c_uint64)
    cap = max(2 * max(num_variables, nnz) + 4, 4).bit_length() + 3
    m = [np.zeros((cap, 8), dtype=np.uint64) for _ in range(3)]
    cap_folds = 4 * cap
    fc = np.zeros((cap_folds, 18), dtype=np.uint64)
    fe = np.zeros((cap_folds, 8), dtype=np.uint64)
    P = ProofRec()
    for k in range(3):
        P.messages[k] = m[k].ctypes.data_as(U)
    P.cap_folds = cap_folds
    P.fold_commitments = fc.ctypes.data_as(U)
    P.fold_evaluations = fe.ctypes.data_as(U)
    return P, cap, (m, fc, fe)


def _unpack_proof(P, bufs) -> "Proof":
    m, fc, fe = bufs
    A = lambda a: np.array(a, dtype=np.uint64)  # noqa: E731
    msgs = lambda k: [(m[k][i, :4].copy(), m[k][i, 4:].copy()) for i in range(P.rounds[k])]  # noqa: E731
    ff = lambda rec: [(A(rec)[:4].copy(), A(rec)[4:].copy())]  # noqa: E731
    prods = [A(P.products[k]) for k in range(9)]
    nf = P.nfold
    tc = TensorcheckProof([fc[i].copy() for i in range(nf)], [fe[i].reshape(2, 4).copy() for i in range(nf)], A(P.evaluation_proof),
                          [A(P.base_evaluations[k]).reshape(3, 4) for k in range(22)])
    ep = EntryProductMsgs([A(P.acc_v_commitments[k]) for k in range(9)], [A(P.claimed_sumchecks[k]) for k in range(9)])
    third_ff = []
    for j in range(13):
        r = A(P.third_final_foldings[j])
        third_ff.append((r[:4].copy(), r[4:].copy()))
    sc = [A(P.sorted_commitments[k]) for k in range(3)]
    proof = Proof(
        witness_commitment=A(P.witness_commitment), zc_alpha=A(P.zc_alpha), first_sumcheck_msgs=(msgs(0), ff(P.final_foldings[0])),
        r_star_commitments=[A(P.r_star_commitments[k]) for k in range(3)], z_star_commitment=A(P.z_star_commitment),
        second_sumcheck_msgs=(msgs(1), ff(P.final_foldings[1])),
        set_r_ep=prods[0], subset_r_ep=prods[1], sorted_r_commitment=sc[0], set_alpha_ep=prods[3], subset_alpha_ep=prods[4],
        sorted_alpha_commitment=sc[1], set_z_ep=prods[6], subset_z_ep=prods[7], sorted_z_commitment=sc[2], ep_msgs=ep,
        ralpha_star_acc_mu_evals=[A(P.ralpha_star_acc_mu_evals[k]) for k in range(10)], ralpha_star_acc_mu_proof=A(P.ralpha_star_acc_mu_proof),
        rstars_vals=[A(P.rstars_vals[0]), A(P.rstars_vals[1])], third_sumcheck_msgs=(msgs(2), third_ff), tensorcheck_proof=tc)
    return proof


def _new_time_native(ck: CommitterKey, r1cs: R1cs, index: list, preprocess_in_library: bool = False, elastic=None) -> "Proof":
    """gm_psnark_new_time; elastic = (ck_stream, r1cs_stream, max_msm_buffer): gm_psnark_new_elastic over the same instance record"""
    import ctypes as C

    from . import capi
    from .transcript import default_group_encoding

    Instance, ProofRec = _psnark_ctypes()
    U = C.POINTER(C.c_uint64)
    idx = np.ascontiguousarray(np.stack(index), dtype=np.uint64)
    g2 = ck.powers_of_g2_bytes()
    g2buf = (C.c_uint8 * len(g2)).from_buffer_copy(g2)
    if preprocess_in_library:  # the matrix-only part of the record comes from gm_psnark_preprocess; z, w, index, G2 bytes are ours
        I = Instance.from_buffer_copy(_joint_native(r1cs).rec)
        I.z, I.w = r1cs.z.handle, r1cs.w.handle
        I.index_commitments = idx.ctypes.data_as(U)
        I.ck_g2_bytes, I.ck_g2_len = C.cast(g2buf, C.POINTER(C.c_uint8)), len(g2)
        nnz = I.nnz
    else:
        jd = _joint_device(r1cs)
        nrows = max(r1cs.a.nrows, r1cs.b.nrows)
        len_r = 1 << max(nrows - 1, 0).bit_length()  # the first sumcheck's tensor: 2^rounds, rounds = ceil(log2 max(|A z|, |B z|))
        ext = jd.extended_frequencies(len_r, len(r1cs.z))
        I = Instance(r1cs.a.handle, r1cs.b.handle, r1cs.c.handle, r1cs.z.handle, r1cs.w.handle, jd.row_index.handle, jd.col_index.handle,
                     len(jd.row_index), jd.row.handle, jd.col.handle, jd.val_a.handle, jd.val_b.handle, jd.val_c.handle, ext[0].handle, ext[1].handle,
                     len(ext[0]), len(ext[1]), idx.ctypes.data_as(U), C.cast(g2buf, C.POINTER(C.c_uint8)), len(g2))
        nnz = len(jd.row_index)
    P, cap, bufs = _proof_buffers(len(r1cs.z), nnz)
    if elastic is None:
        capi.check(capi.load().gm_psnark_new_time(C.byref(I), C.c_uint64(ck.powers_of_g.handle), C.c_int(int(default_group_encoding())), C.c_size_t(cap),
                                                  C.byref(P)))
    else:
        cks, st, max_msm_buffer = elastic
        h = lambda v: C.c_uint64(v.handle)  # noqa: E731
        capi.check(capi.load().gm_psnark_new_elastic(C.byref(I), h(st.z), h(st.witness), h(st.z_a), h(st.z_b), h(st.z_c), C.c_uint64(cks.powers_of_g.handle),
                                                     C.c_size_t(max_msm_buffer), C.c_size_t(cks.min_device_chunk), C.c_int(int(default_group_encoding())),
                                                     C.c_size_t(cap), C.byref(P)))
    proof = _unpack_proof(P, bufs)
    proof.spans = {name: P.spans[i] for i, name in enumerate(_PSNARK_SPANS)}
    if elastic is not None:
        proof.spans["ark_gemini::psnark::elastic_prover"] = proof.spans.pop("ark_gemini::psnark::time_prover")
    return proof

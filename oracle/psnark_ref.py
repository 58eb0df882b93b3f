"""CPU restatement of the preprocessing SNARK time prover (TEST INFRASTRUCTURE ONLY).

Follows src/psnark/time_prover.rs:49-384, src/subprotocols/entryproduct/time_prover.rs:14-114,
src/subprotocols/plookup/time_prover.rs:5-112, src/misc.rs:269-366 and src/kzg/time.rs:86-95 on Python
integers (oracle/pyref.py), literally: the indexed committer keys of `index_by` are built point by point
and committed against (the HIP path commits the looked-up vectors under the plain key instead, so this
also checks that equivalence).  Parity with a Rust run of the reference is unpinned at the byte level
(see pyref.py header); what this pins is HIP path == independent restatement, transcript included.
"""
from __future__ import annotations

from . import oracle as orc
from . import pyref as P
from . import snark_ref as sr

R = P.R_MOD
Q = P.Q_MOD

# ---- G2 for `transcript.append_serializable(b"ck", &ck.powers_of_g2)` --------------------------------
G2_GEN = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
           0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
          (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
           0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _f2inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
    return (a[0] * d % Q, (Q - a[1]) * d % Q)


def g2_add(p, q):
    """affine chord-and-tangent on y^2 = x^3 + 4(1+u)"""
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if ((y1[0] + y2[0]) % Q, (y1[1] + y2[1]) % Q) == (0, 0):
            return None
        xx = _f2mul(x1, x1)
        lam = _f2mul(((3 * xx[0]) % Q, (3 * xx[1]) % Q), _f2inv(((2 * y1[0]) % Q, (2 * y1[1]) % Q)))
    else:
        lam = _f2mul(((y2[0] - y1[0]) % Q, (y2[1] - y1[1]) % Q), _f2inv(((x2[0] - x1[0]) % Q, (x2[1] - x1[1]) % Q)))
    l2 = _f2mul(lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % Q, (l2[1] - x1[1] - x2[1]) % Q)
    t = _f2mul(lam, ((x1[0] - x3[0]) % Q, (x1[1] - x3[1]) % Q))
    return (x3, ((t[0] - y1[0]) % Q, (t[1] - y1[1]) % Q))


def g2_mul(p, k: int):
    """LSB-first double-and-add"""
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, p)
        p = g2_add(p, p)
        k >>= 1
    return acc


def g2_serialize_uncompressed(p) -> bytes:
    """x.c0 | x.c1 | y.c0 | y.c1, flags in the top bits of the last byte; `y > -y` compares c1 first
    (QuadExtField's Ord)  [recalled like pyref.g1_serialize_uncompressed -- unverifiable here]"""
    if p is None:
        out = bytearray(192)
        out[191] |= 1 << 6
        return bytes(out)
    (x0, x1), (y0, y1) = p
    out = bytearray(b"".join(v.to_bytes(48, "little") for v in (x0, x1, y0, y1)))
    ny = ((Q - y0) % Q, (Q - y1) % Q)
    if (y1, y0) > (ny[1], ny[0]):
        out[191] |= 1 << 7
    return bytes(out)


def powers_of_g2(tau: int, max_eval_points: int):
    """src/kzg/time.rs:60-67"""
    return [g2_mul(G2_GEN, pow(tau, i, R)) for i in range(max_eval_points + 1)]


# ---- src/misc.rs:269-366 -----------------------------------------------------------------------------
def sum_matrices(a, b, c, num_variables):
    new_matrix = [set() for _ in range(num_variables)]
    for row, (ra, rb, rc) in enumerate(zip(a, b, c)):
        for _, col in list(ra) + list(rb) + list(rc):
            new_matrix[col].add(row)
    return [sorted(s) for s in new_matrix]


def joint_matrices(joint_matrix, a, b, c):
    da = {(r, i): f for r, row in enumerate(a) for f, i in row}
    db = {(r, i): f for r, row in enumerate(b) for f, i in row}
    dc = {(r, i): f for r, row in enumerate(c) for f, i in row}
    row_vec, col_vec, row_index, col_index, va, vb, vc = [], [], [], [], [], [], []
    for cc, col in enumerate(joint_matrix):
        for i in col:
            row_index.append(i)
            col_index.append(cc)
            row_vec.append(i % R)
            col_vec.append(cc % R)
            va.append(da.get((i, cc), 0))
            vb.append(db.get((i, cc), 0))
            vc.append(dc.get((i, cc), 0))
    return row_vec, col_vec, row_index, col_index, va, vb, vc


# ---- src/subprotocols/plookup/time_prover.rs ---------------------------------------------------------
def lookup(v, index):
    return [v[i] for i in index]


def alg_hash(v, index, chal):
    return [(vi + i * chal) % R for vi, i in zip(v, index)]


def plookup_set(v, y, z):
    y1z = (1 + z) * y % R
    n = len(v)
    if n == 0:
        return []
    return [(y1z + z * v[0]) % R] + [(y1z + v[i] + z * v[i + 1]) % R for i in range(n - 1)] + [(y1z + v[n - 1]) % R]


def plookup_subset(v, y):
    return [(e + y) % R for e in v]


def compute_frequency(set_len, index):
    f = [1] * set_len
    for i in index:
        f[i] += 1
    return f


def extend_frequency(frequency):
    res = []
    for i, f in enumerate(frequency):
        res += [i] * f
    return res


def sorted_(set_, frequency):
    out = []
    for f, e in zip(frequency, set_):
        out += [e] * f
    return out


def plookup(subset, set_, index, y, z, zeta):
    if zeta != 0:
        set_, subset = alg_hash(set_, range(len(set_)), zeta), alg_hash(subset, index, zeta)
    lookup_set = plookup_set(set_, y, z)
    lookup_subset = plookup_subset(subset, y)
    frequency = compute_frequency(len(set_), index)
    lookup_sorted = plookup_set(sorted_(set_, frequency), y, z)
    return [lookup_set, lookup_subset, lookup_sorted]


# ---- src/subprotocols/entryproduct/time_prover.rs ----------------------------------------------------
def right_rotation(v):
    return [v[-1]] + list(v[:-1]) if v else []


def accumulated_product(v):
    out, state = [], 1
    for e in reversed(v):
        state = state * e % R
        out.append(state)
    out.reverse()
    return out


def monic(v):
    return list(v) + [1]


def product(v):
    acc = 1
    for e in v:
        acc = acc * e % R
    return acc


def entry_product_new_time_batch(tr, powers_of_g, vs, claimed_products):
    assert len(vs) == len(claimed_products)
    monic_vs = [monic(v) for v in vs]
    rrot_vs = [right_rotation(v) for v in monic_vs]
    acc_vs = [accumulated_product(v) for v in monic_vs]
    acc_v_commitments = [sr.commit(powers_of_g, a) for a in acc_vs]
    for c in acc_v_commitments:
        tr.append_message(b"acc_v", P.g1_serialize_uncompressed(c))
    chal = tr.get_challenge(b"ep-chal")
    provers = [P.TimeProver(acc_v, rrot_v, chal) for rrot_v, acc_v in zip(rrot_vs, acc_vs)]
    claimed_sumchecks = [(P.evaluate_le(acc_v, chal) * chal + cp - pow(chal, len(acc_v), R)) % R for cp, acc_v in zip(claimed_products, acc_vs)]
    return {"acc_v_commitments": acc_v_commitments, "claimed_sumchecks": claimed_sumchecks}, chal, provers


# ---- src/kzg/time.rs:86-95 -----------------------------------------------------------------------------
def index_by(powers_of_g_aff, indices):
    """powers_of_g_aff: list of affine integer points (or None)"""
    out = [None] * len(powers_of_g_aff)
    for i, g in zip(indices, powers_of_g_aff):
        out[i] = P.g1_add(out[i], g)
    return out


def _commit_aff(powers_aff, poly):
    """msm over affine integer points (src/kzg/time.rs:81-83 truncation), naive"""
    n = min(len(powers_aff), len(poly))
    return P.msm_naive(powers_aff[:n], [c % R for c in poly[:n]])


def index(powers_of_g, r1cs):
    """src/psnark/time_prover.rs:49-64"""
    jm = sum_matrices(r1cs["a"], r1cs["b"], r1cs["c"], len(r1cs["z"]))
    row, col, _, _, va, vb, vc = joint_matrices(jm, r1cs["a"], r1cs["b"], r1cs["c"])
    return [sr.commit(powers_of_g, p) for p in (row, col, va, vb, vc)]


def psnark_new_time(powers_of_g, g2_powers, r1cs, index_commitments):
    """src/psnark/time_prover.rs:69-384.  powers_of_g: (n, 12) limb array as snark_ref.srs returns."""
    a, b, c, z, w = r1cs["a"], r1cs["b"], r1cs["c"], r1cs["z"], r1cs["w"]
    G1 = P.g1_serialize_uncompressed
    z_a, z_b, z_c = sr.matvec(a, z), sr.matvec(b, z), sr.matvec(c, z)
    tr = P.GeminiTranscript(P.PROTOCOL_NAME)
    witness_commitment = sr.commit(powers_of_g, w)
    tr.append_message(b"witness", G1(witness_commitment))
    tr.append_message(b"ck", len(g2_powers).to_bytes(8, "little") + b"".join(g2_serialize_uncompressed(p) for p in g2_powers))
    tr.append_message(b"instance", len(index_commitments).to_bytes(8, "little") + b"".join(G1(p) for p in index_commitments))
    alpha = tr.get_challenge(b"alpha")
    zc_alpha = P.evaluate_le(z_c, alpha)
    tr.append_fr(b"zc(alpha)", zc_alpha)
    m1, ch1, ff1 = P.sumcheck_prove(tr, P.TimeProver(z_a, z_b, alpha))
    b_ch = P.tensor(ch1)
    c_ch = P.powers(alpha, len(b_ch))
    a_ch = P.hadamard(b_ch, c_ch)
    num_variables = len(z)
    jm = sum_matrices(a, b, c, num_variables)
    row, col, row_index, col_index, val_a, val_b, val_c = joint_matrices(jm, a, b, c)
    nnz = len(row)
    ralpha_star, r_star, alpha_star, z_star = lookup(a_ch, row_index), lookup(b_ch, row_index), lookup(c_ch, row_index), lookup(z, col_index)

    pg_aff = [orc.affine_to_ints(powers_of_g[i]) for i in range(len(powers_of_g))]
    ck_row, ck_col = index_by(pg_aff, row_index), index_by(pg_aff, col_index)
    z_r_commitments = [_commit_aff(ck_row, a_ch), _commit_aff(ck_row, b_ch), _commit_aff(ck_row, c_ch), _commit_aff(ck_col, z)]
    for label, cm in zip((b"ra*", b"rb*", b"rc*", b"z*"), z_r_commitments):
        tr.append_message(label, G1(cm))
    eta = tr.get_challenge(b"chal")
    challenges = P.powers(eta, 3)
    r_star_val = P.linear_combination([P.hadamard(ralpha_star, val_a), P.hadamard(r_star, val_b), P.hadamard(alpha_star, val_c)], challenges)
    m2, ch2, ff2 = P.sumcheck_prove(tr, P.TimeProver(z_star, r_star_val, 1))
    second_challenges_head = P.tensor(ch2)[:nnz]
    zeta = tr.get_challenge(b"zeta")
    alg_hash_poly = [alg_hash(b_ch, range(len(b_ch)), zeta), alg_hash(c_ch, range(len(c_ch)), zeta), alg_hash(z, range(len(z)), zeta)]
    frequency = [compute_frequency(len(alg_hash_poly[0]), row_index), compute_frequency(len(alg_hash_poly[2]), col_index)]
    sorted_polynomials = [sorted_(alg_hash_poly[0], frequency[0]), sorted_(alg_hash_poly[1], frequency[0]), sorted_(alg_hash_poly[2], frequency[1])]
    ext_fre = [extend_frequency(frequency[0]), extend_frequency(frequency[1])]
    ck_fre = [index_by(pg_aff, ext_fre[0]), index_by(pg_aff, ext_fre[1])]
    sorted_commitments = [_commit_aff(ck_fre[0], alg_hash_poly[0]), _commit_aff(ck_fre[0], alg_hash_poly[1]), _commit_aff(ck_fre[1], alg_hash_poly[2])]
    tr.append_message(b"sorted_alpha_commitment", G1(sorted_commitments[1]))
    tr.append_message(b"sorted_r_commitment", G1(sorted_commitments[0]))
    tr.append_message(b"sorted_z_commitment", G1(sorted_commitments[2]))
    gamma = tr.get_challenge(b"gamma")
    chi = tr.get_challenge(b"chi")
    r_lookup_vec = plookup(r_star, b_ch, row_index, gamma, chi, zeta)
    alpha_lookup_vec = plookup(alpha_star, c_ch, row_index, gamma, chi, zeta)
    z_lookup_vec = plookup(z_star, z, col_index, gamma, chi, zeta)
    r_prod_vec, alpha_prod_vec, z_prod_vec = [product(v) for v in r_lookup_vec], [product(v) for v in alpha_lookup_vec], [product(v) for v in z_lookup_vec]
    lookup_vec = r_lookup_vec + alpha_lookup_vec + z_lookup_vec
    accumulated_vec = [accumulated_product(monic(v)) for v in lookup_vec]
    tr.append_fr(b"set_r_ep", alpha_prod_vec[0])
    tr.append_fr(b"subset_r_ep", alpha_prod_vec[1])
    tr.append_fr(b"set_r_ep", r_prod_vec[0])
    tr.append_fr(b"subset_r_ep", r_prod_vec[1])
    tr.append_fr(b"set_z_ep", z_prod_vec[0])
    tr.append_fr(b"subset_z_ep", z_prod_vec[1])
    ep_msgs, psi, ep_provers = entry_product_new_time_batch(tr, powers_of_g, lookup_vec, r_prod_vec + alpha_prod_vec + z_prod_vec)
    open_chal = tr.get_challenge(b"open-chal")
    polynomials = [ralpha_star] + accumulated_vec
    mu_proof = sr.batch_open_multi_points(powers_of_g, polynomials, [psi], open_chal)
    mu_evals = [P.evaluate_le(p, psi) for p in polynomials]
    s_0_prime = P.ip(P.hadamard(ralpha_star, val_a), second_challenges_head)
    s_1_prime = P.ip(P.hadamard(r_star, val_b), second_challenges_head)
    for e in mu_evals:
        tr.append_fr(b"ralpha_star_acc_mu", e)
    tr.append_message(b"ralpha_star_mu_proof", G1(mu_proof))
    provers = list(ep_provers)
    provers.append(P.TimeProver(P.hadamard(ralpha_star, second_challenges_head), val_a, 1))
    provers.append(P.TimeProver(P.hadamard(r_star, second_challenges_head), val_b, 1))
    provers.append(P.TimeProver(P.hadamard(alpha_star, second_challenges_head), val_c, 1))
    provers.append(P.TimeProver(r_star, alpha_star, psi))
    m3, ch3, ff3 = P.sumcheck_prove_batch(tr, provers)
    tc_base = [w, ralpha_star, r_star, alpha_star, z_star, row, col, val_a, val_b, val_c] + sorted_polynomials + accumulated_vec
    twist_powers2 = P.powers2(psi, len(ch3))
    shift_monic_lookup_vec = [right_rotation(monic(v)) for v in lookup_vec]
    third_proof_vec = shift_monic_lookup_vec + [val_a, val_b, val_c, alpha_star]
    body0 = accumulated_vec + [r_star]
    head = ch3[: len(ch2)]
    tc_body = [(body0, P.hadamard(ch3, twist_powers2)), (third_proof_vec, ch3), ([z_star], ch2),
               ([ralpha_star, r_star, alpha_star], P.hadamard(ch2, head))]
    tc = sr.tensorcheck_new_time(tr, powers_of_g, tc_base, tc_body)
    return {
        "witness_commitment": witness_commitment, "zc_alpha": zc_alpha, "first_sumcheck_msgs": (m1, ff1),
        "r_star_commitments": z_r_commitments[:3], "z_star_commitment": z_r_commitments[3], "second_sumcheck_msgs": (m2, ff2),
        "set_r_ep": r_prod_vec[0], "subset_r_ep": r_prod_vec[1], "sorted_r_commitment": sorted_commitments[0],
        "set_alpha_ep": alpha_prod_vec[0], "subset_alpha_ep": alpha_prod_vec[1], "sorted_alpha_commitment": sorted_commitments[1],
        "set_z_ep": z_prod_vec[0], "subset_z_ep": z_prod_vec[1], "sorted_z_commitment": sorted_commitments[2],
        "ep_msgs": ep_msgs, "ralpha_star_acc_mu_evals": mu_evals, "ralpha_star_acc_mu_proof": mu_proof,
        "rstars_vals": [s_0_prime, s_1_prime], "third_sumcheck_msgs": (m3, ff3), "tensorcheck_proof": tc,
        "challenges": {"alpha": alpha, "eta": eta, "zeta": zeta, "gamma": gamma, "chi": chi, "psi": psi, "third": ch3},
    }

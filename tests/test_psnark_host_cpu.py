"""CPU checks of the host-side logic of the preprocessing SNARK mirror (no GPU needed): joint matrices,
plookup frequency helpers, the G2 half of the committer key and its serialization, against the
restatement in oracle/psnark_ref.py; plus the reference's RNG-free plookup tests on the oracle itself."""
import numpy as np

from oracle import psnark_ref as pr
from oracle import pyref as P


class _Csr:
    """stand-in for gemini_amd.circuit.SparseMatrix: only what joint_matrices reads"""

    def __init__(self, rows, nrows):
        trip = [(i, c, v) for i, row in enumerate(rows) for (v, c) in row]
        rowptr = np.zeros(nrows + 1, dtype=np.uint64)
        for i, _, _ in trip:
            rowptr[i + 1] += 1
        self.nrows = nrows
        self.csr = (np.cumsum(rowptr).astype(np.uint64), np.array([c for _, c, _ in trip], dtype=np.uint32),
                    np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for _, _, v in trip], dtype=np.uint64).reshape(-1, 4))


def _ints(a):
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) for row in np.asarray(a).reshape(-1, 4)]


def test_joint_matrices_match_the_restatement():
    """src/misc.rs:269-366: column-major walk of the union support, zeros where a matrix has no entry,
    the last duplicate wins (BTreeMap::collect)"""
    from gemini_amd.psnark import joint_matrices

    rng = P.SplitMix64(5)
    for n, nv in ((8, 8), (33, 20), (64, 64)):
        mk = lambda: [[(rng.fr() % (1 << 200), int(rng.next() % nv)) for _ in range(int(rng.next() % 4))] for _ in range(n)]
        a, b, c = mk(), mk(), mk()
        a[0] = [(5, 1), (7, 1)]  # duplicate (row, col): the later value wins
        jm = pr.sum_matrices(a, b, c, nv)
        row, col, ri, ci, va, vb, vc = pr.joint_matrices(jm, a, b, c)
        A, B, Cm = _Csr(a, n), _Csr(b, n), _Csr(c, n)
        gri, gci, gva, gvb, gvc = joint_matrices(A, B, Cm, n, nv)
        assert gri.tolist() == ri and gci.tolist() == ci
        assert _ints(gva) == va and _ints(gvb) == vb and _ints(gvc) == vc
        # the same object used three times (dummy_r1cs) takes the shared-key path
        gri2, gci2, x, y, z = joint_matrices(A, A, A, n, nv)
        jm2 = pr.sum_matrices(a, a, a, nv)
        _, _, ri2, ci2, va2, _, _ = pr.joint_matrices(jm2, a, a, a)
        assert gri2.tolist() == ri2 and gci2.tolist() == ci2 and _ints(x) == va2 and x is y and y is z


def test_frequency_helpers():
    """plookup/time_prover.rs:66-87"""
    from gemini_amd.psnark import compute_frequency, extend_frequency

    rng = np.random.default_rng(3)
    for set_len, m in ((6, 4), (1, 0), (50, 500)):
        idx = rng.integers(0, set_len, size=m).astype(np.uint32)
        f = compute_frequency(set_len, idx)
        assert f.tolist() == pr.compute_frequency(set_len, idx.tolist())
        assert extend_frequency(f).tolist() == pr.extend_frequency(f.tolist())


def test_g2_powers_and_serialization():
    """two independent G2 implementations (Jacobian MSB-first vs affine LSB-first) and byte framings agree;
    the generator has order r"""
    from gemini_amd import g2

    assert g2.on_curve(g2.generator()) and g2.generator() == pr.G2_GEN
    assert g2.mul(g2.generator(), g2.R_ORDER - 1) == (g2.G2_X, g2.f2_neg(g2.G2_Y))
    assert pr.g2_add(g2.mul(g2.generator(), g2.R_ORDER - 1), pr.G2_GEN) is None
    tau = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF
    want = pr.powers_of_g2(tau, 3)
    got = [g2.mul(g2.generator(), pow(tau, i, P.R_MOD)) for i in range(4)]
    assert got == want and all(g2.on_curve(p) for p in got)
    blob = g2.serialize_vec_uncompressed(got + [None])
    assert blob == (5).to_bytes(8, "little") + b"".join(pr.g2_serialize_uncompressed(p) for p in want + [None])
    assert len(blob) == 8 + 5 * 192 and blob[-1] == 0x40
    # the sign flag follows y > -y with c1 compared first
    neg = (got[1][0], g2.f2_neg(got[1][1]))
    assert (g2.serialize_uncompressed(got[1])[-1] ^ g2.serialize_uncompressed(neg)[-1]) & 0x80


def test_plookup_relation_reference_tests():
    """plookup/time_prover.rs:114-148 test_plookup_relation and :37-60 test_plookup_set_correct on the oracle"""
    R = P.R_MOD
    set_ = [10, 12, 13, 14, 15, 42]
    subset = [10, 13, 15, 42]
    indices = [0, 2, 4, 5]
    y, z = 47, 52
    lv = pr.plookup(subset, set_, indices, y, z, 0)
    prod = [pr.product(v) for v in lv]
    assert prod[2] == prod[0] * prod[1] % R * pow(1 + z, len(subset), R) % R
    rng = P.SplitMix64(9)
    s3 = [rng.fr() for _ in range(3)]
    yy, zz, chal = rng.fr(), rng.fr(), rng.fr()
    pl = pr.plookup_set(s3, yy, zz)
    y1z = (1 + zz) * yy % R
    first = y1z * (pow(chal, len(s3) + 1, R) - 1) % R * pow(chal - 1, -1, R) % R
    assert P.evaluate_le(pl, chal) == (first + P.evaluate_le(s3, chal) * (chal + zz)) % R
    # accumulated_product is the reverse prefix product; right_rotation moves the last element first
    v = [rng.fr() for _ in range(7)]
    acc = pr.accumulated_product(v)
    assert acc[0] == pr.product(v) and acc[-1] == v[-1] and pr.right_rotation(pr.monic(v)) == [1] + v

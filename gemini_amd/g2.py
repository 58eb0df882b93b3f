"""BLS12-381 G2 on the host, for the O(max_eval_points) verifier half of the KZG key
(`powers_of_g2`, src/kzg/time.rs:60-67) that the preprocessing SNARK absorbs into its transcript
(src/psnark/time_prover.rs:85).  Setup-time work on a handful of points: plain Python integers,
Jacobian coordinates over Fq2 = Fq[u]/(u^2 + 1)."""
from __future__ import annotations

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001

# generator of the r-torsion on the twist y^2 = x^3 + 4(1 + u)
G2_X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2_Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)


def f2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


ZERO2, ONE2 = (0, 0), (1, 0)


def on_curve(p) -> bool:
    if p is None:
        return True
    x, y = p
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), (4, 4))


def _jdbl(P):
    X, Y, Z = P
    if Z == ZERO2:
        return P
    A = f2_mul(X, X)
    B = f2_mul(Y, Y)
    Cc = f2_mul(B, B)
    t = f2_add(X, B)
    D = f2_sub(f2_sub(f2_mul(t, t), A), Cc)
    D = f2_add(D, D)
    E = f2_add(f2_add(A, A), A)
    F = f2_mul(E, E)
    X3 = f2_sub(F, f2_add(D, D))
    C8 = f2_add(Cc, Cc)
    C8 = f2_add(C8, C8)
    C8 = f2_add(C8, C8)
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), C8)
    Z3 = f2_mul(Y, Z)
    return (X3, Y3, f2_add(Z3, Z3))


def _jadd_affine(P, q):
    """Jacobian + affine (q != identity)"""
    X1, Y1, Z1 = P
    if Z1 == ZERO2:
        return (q[0], q[1], ONE2)
    Z1Z1 = f2_mul(Z1, Z1)
    U2 = f2_mul(q[0], Z1Z1)
    S2 = f2_mul(f2_mul(q[1], Z1), Z1Z1)
    if U2 == X1:
        return _jdbl(P) if S2 == Y1 else (ONE2, ONE2, ZERO2)
    H = f2_sub(U2, X1)
    r = f2_sub(S2, Y1)
    HH = f2_mul(H, H)
    HHH = f2_mul(H, HH)
    V = f2_mul(X1, HH)
    X3 = f2_sub(f2_sub(f2_mul(r, r), HHH), f2_add(V, V))
    Y3 = f2_sub(f2_mul(r, f2_sub(V, X3)), f2_mul(Y1, HHH))
    return (X3, Y3, f2_mul(Z1, H))


def mul(p, k: int):
    """k * p for an affine point p ((x0, x1), (y0, y1)) or None; returns affine / None.  MSB first."""
    k %= R_ORDER
    if p is None or k == 0:
        return None
    acc = (ONE2, ONE2, ZERO2)
    for bit in bin(k)[2:]:
        acc = _jdbl(acc)
        if bit == "1":
            acc = _jadd_affine(acc, p)
    X, Y, Z = acc
    if Z == ZERO2:
        return None
    zi = f2_inv(Z)
    zi2 = f2_mul(zi, zi)
    return (f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)))


def generator():
    return (G2_X, G2_Y)


def _f2_gt(a, b) -> bool:
    """ark-ff QuadExtField Ord: c1 first, then c0"""
    return (a[1], a[0]) > (b[1], b[0])


def serialize_uncompressed(p, enc: int | None = None) -> bytes:
    """ark-serialize short-Weierstrass affine over Fq2, Compress::No.
    enc 0 (ark-ec default, ark-test-curves): x.c0 | x.c1 | y.c0 | y.c1, 48-byte little-endian each, flags in the top
    bits of the last byte (bit 7: y > -y, bit 6: infinity).
    enc 1 (ark-bls12-381, zcash): x.c1 | x.c0 | y.c1 | y.c0 big-endian, flags in the top bits of the first byte
    (bit 7 clear = uncompressed, bit 6: infinity).  Default: the process-wide curve crate (gemini_amd.transcript)."""
    if enc is None:
        from .transcript import default_group_encoding

        enc = default_group_encoding()
    if enc == 1:
        if p is None:
            return bytes([0x40]) + bytes(191)
        x, y = p
        return x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big") + y[1].to_bytes(48, "big") + y[0].to_bytes(48, "big")
    if p is None:
        out = bytearray(192)
        out[-1] |= 1 << 6
        return bytes(out)
    x, y = p
    out = bytearray(x[0].to_bytes(48, "little") + x[1].to_bytes(48, "little") + y[0].to_bytes(48, "little") + y[1].to_bytes(48, "little"))
    if _f2_gt(y, f2_neg(y)):
        out[-1] |= 1 << 7
    return bytes(out)


def deserialize_uncompressed(b: bytes, enc: int = 0):
    """inverse of serialize_uncompressed (no subgroup check: used on the reference's own outputs)"""
    assert len(b) == 192
    if enc == 1:
        if b[0] & 0x40:
            return None
        v = [int.from_bytes(bytes([b[48 * i] & (0x1F if i == 0 else 0xFF)]) + b[48 * i + 1: 48 * i + 48], "big") for i in range(4)]
        return ((v[1], v[0]), (v[3], v[2]))
    if b[-1] & 0x40:
        return None
    v = [int.from_bytes(b[48 * i: 48 * i + 48], "little") for i in range(3)] + [int.from_bytes(b[144:191] + bytes([b[191] & 0x3F]), "little")]
    return ((v[0], v[1]), (v[2], v[3]))


def serialize_vec_uncompressed(points, enc: int | None = None) -> bytes:
    """Vec<G2Affine>: u64 length + items"""
    return len(points).to_bytes(8, "little") + b"".join(serialize_uncompressed(p, enc) for p in points)

"""BLS12-381 ate pairing on Python integers (TEST INFRASTRUCTURE ONLY -- see pyref.py header).

The reference's verifiers decide with `E::pairing(a, g2) == E::pairing(proof, z)` (src/kzg/mod.rs:155-244; ark-ec 0.4.2
`Pairing`, not vendored under /root/reference).  Deciding such an equality needs a bilinear, non-degenerate map on
G1 x G2, not ark-ec's particular normalisation of it: `pairing_product_is_one` below multiplies Miller loops and applies
one final exponentiation, which is the textbook ate pairing up to a fixed power (the loop runs over |x| without the
final conjugation for the negative BLS parameter).  Written for obviousness, not speed: F_q^12 is F_q[w] / (w^12 - 2 w^6 + 2)
(w^6 = 1 + u, u^2 = -1), points of the twist are carried to E(F_q^12) and the line functions are the affine
chord-and-tangent lines evaluated there.  Pinned by bilinearity and non-degeneracy in tests/test_oracle_verifier.py.
"""
from __future__ import annotations

from . import pyref as P
from .psnark_ref import G2_GEN, g2_add, g2_mul  # noqa: F401  (re-exported for the verifier)

Q = P.Q_MOD
R = P.R_MOD
ATE_LOOP = 0xD201000000010000  # |x| of BLS12-381
DEG = 12


# ---- F_q^12 = F_q[w] / (w^12 - 2 w^6 + 2): coefficient lists of length 12 ----------------------------------
def f12(c0: int = 0):
    return [c0 % Q] + [0] * 11


ONE = f12(1)


def f12_add(a, b):
    return [(x + y) % Q for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % Q for x, y in zip(a, b)]


def f12_scalar(a, k: int):
    return [x * k % Q for x in a]


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    # w^12 = 2 w^6 - 2
    for k in range(22, 11, -1):
        v = t[k]
        if v:
            t[k - 6] += 2 * v
            t[k - 12] -= 2 * v
    return [v % Q for v in t[:12]]


def _poly_deg(p):
    d = len(p) - 1
    while d >= 0 and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """extended Euclid in F_q[w] against the modulus polynomial"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [2, 0, 0, 0, 0, 0, Q - 2, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low) > 0:
        # r = high div low
        r = [0] * 13
        tmp = list(high)
        dl = _poly_deg(low)
        inv_lead = pow(low[dl], -1, Q)
        for i in range(_poly_deg(tmp) - dl, -1, -1):
            c = tmp[dl + i] * inv_lead % Q
            r[i] = c
            if c:
                for j in range(dl + 1):
                    tmp[i + j] = (tmp[i + j] - c * low[j]) % Q
        nm, new = list(hm), list(high)
        for i in range(13):
            if lm[i] or low[i]:
                for j in range(13 - i):
                    if r[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                        new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    assert low[0] % Q != 0, "inverse of zero in F_q^12"
    inv0 = pow(low[0], -1, Q)
    return [x * inv0 % Q for x in lm[:12]]


def f12_pow(a, e: int):
    acc = ONE
    while e:
        if e & 1:
            acc = f12_mul(acc, a)
        a = f12_mul(a, a)
        e >>= 1
    return acc


# ---- points of E(F_q^12) ---------------------------------------------------------------------------------------
W = [0, 1] + [0] * 10
_W2_INV = f12_inv(f12_mul(W, W))
_W3_INV = f12_inv(f12_mul(f12_mul(W, W), W))


def _f2_to_f12(c):
    """a + b u with u = w^6 - 1"""
    a, b = c
    out = [0] * 12
    out[0] = (a - b) % Q
    out[6] = b % Q
    return out


def untwist(q2):
    """E'(F_q^2): y^2 = x^3 + 4 (1 + u)  ->  E(F_q^12): y^2 = x^3 + 4,  (x, y) -> (x / w^2, y / w^3)"""
    x, y = q2
    return (f12_mul(_f2_to_f12(x), _W2_INV), f12_mul(_f2_to_f12(y), _W3_INV))


def embed_g1(p1):
    return (f12(p1[0]), f12(p1[1]))


def _e12_double(p):
    x, y = p
    lam = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_mul(lam, lam), f12_scalar(x, 2))
    ny = f12_sub(f12_mul(lam, f12_sub(x, nx)), y)
    return (nx, ny)


def _e12_add(p, q):
    (x1, y1), (x2, y2) = p, q
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    ny = f12_sub(f12_mul(lam, f12_sub(x1, nx)), y1)
    return (nx, ny)


def _line(p1, p2, t):
    """the line through p1 and p2 (tangent if equal) evaluated at t"""
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    elif y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
    else:
        return f12_sub(xt, x1)
    return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))


def miller_loop(q2, p1):
    """f_{|x|, Q}(P) for Q on the twist (affine over F_q^2) and P in G1 (affine); 1 if either is the identity"""
    if q2 is None or p1 is None:
        return ONE
    qq, pp = untwist(q2), embed_g1(p1)
    r = qq
    f = ONE
    for bit in bin(ATE_LOOP)[3:]:
        f = f12_mul(f12_mul(f, f), _line(r, r, pp))
        r = _e12_double(r)
        if bit == "1":
            f = f12_mul(f, _line(r, qq, pp))
            r = _e12_add(r, qq)
    return f


FINAL_EXP = (Q**12 - 1) // R


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(q2, p1):
    return final_exponentiation(miller_loop(q2, p1))


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 for (G1 affine, G2 affine) pairs: one final exponentiation for the whole product"""
    f = ONE
    for p1, q2 in pairs:
        f = f12_mul(f, miller_loop(q2, p1))
    return final_exponentiation(f) == ONE

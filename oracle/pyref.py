"""Independent big-integer restatement of the Gemini prover hot path (TEST INFRASTRUCTURE ONLY).

This file is *not* product code.  It is the slow, obviously-correct Python `int`
statement of the arithmetic the hot path computes, used for exactly two things:

  * pinning `oracle/gemini_oracle.c` (the C restatement that the parity tests and
    bench.py's cpu_baseline use) on small inputs, and
  * generating the committed fixtures under tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.

Reference citations are relative to /root/reference (arkworks-rs/gemini).  The
field/curve arithmetic itself lives in third-party crates that are NOT vendored
there (ark-ff / ark-ec 0.4.2 @ arkworks-rs/algebra df51425, merlin 3.0.0,
keccak 0.1.4 -- Cargo.lock:44-46,62-64,606-608,564-566), so it is restated from
the published definitions (BLS12-381 parameters, Montgomery form, STROBE-128,
Keccak-f[1600]).

Parity status: MSM / sumcheck results are mathematically unique (a group element,
field elements), so any correct implementation is bit-exact after normalisation;
byte-level commitment/transcript parity with a Rust run is UNPINNED (no Rust
toolchain, reference tests hold no golden bytes -- SURVEY.md section 8c).  What
is pinned: the reference's RNG-free known-answer tests (tests/test_oracle_kat.py),
SHA3/Keccak against hashlib, and Merlin against merlin's published test vector.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------
# BLS12-381 parameters
# ----------------------------------------------------------------------------
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # Fr
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB  # Fq
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
B_COEFF = 4

FR_LIMBS64 = 4
FQ_LIMBS64 = 6
FR_MONT_R = (1 << 256) % R_MOD
FQ_MONT_R = (1 << 384) % Q_MOD
FR_MONT_R2 = (FR_MONT_R * FR_MONT_R) % R_MOD
FQ_MONT_R2 = (FQ_MONT_R * FQ_MONT_R) % Q_MOD
FR_INV64 = (-pow(R_MOD, -1, 1 << 64)) % (1 << 64)
FQ_INV64 = (-pow(Q_MOD, -1, 1 << 64)) % (1 << 64)
FR_INV32 = FR_INV64 & 0xFFFFFFFF
FQ_INV32 = FQ_INV64 & 0xFFFFFFFF


def fr_to_mont(a: int) -> int:
    return (a * FR_MONT_R) % R_MOD


def fr_from_mont(a: int) -> int:
    return (a * pow(FR_MONT_R, -1, R_MOD)) % R_MOD


def fq_to_mont(a: int) -> int:
    return (a * FQ_MONT_R) % Q_MOD


def fq_from_mont(a: int) -> int:
    return (a * pow(FQ_MONT_R, -1, Q_MOD)) % Q_MOD


def limbs64(a: int, n: int) -> list[int]:
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs64(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ----------------------------------------------------------------------------
# G1: y^2 = x^3 + 4 over Fq.  Affine points are (x, y) tuples, identity is None.
# ----------------------------------------------------------------------------
def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - B_COEFF) % Q_MOD == 0


def g1_neg(P):
    if P is None:
        return None
    return (P[0], (-P[1]) % Q_MOD)


def g1_add(P, Q):
    """Affine chord-and-tangent addition, complete (handles P==Q, P==-Q, identity)."""
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    y3 = (lam * (x1 - x3) - y1) % Q_MOD
    return (x3, y3)


# Jacobian helpers (faster scalar multiplication in pure Python): (X, Y, Z), Z == 0 is identity
def _jac_double(P):
    X, Y, Z = P
    if Z == 0:
        return P
    A = X * X % Q_MOD
    B = Y * Y % Q_MOD
    C = B * B % Q_MOD
    D = 2 * ((X + B) * (X + B) - A - C) % Q_MOD
    E = 3 * A % Q_MOD
    F = E * E % Q_MOD
    X3 = (F - 2 * D) % Q_MOD
    Y3 = (E * (D - X3) - 8 * C) % Q_MOD
    Z3 = 2 * Y * Z % Q_MOD
    return (X3, Y3, Z3)


def _jac_add(P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % Q_MOD
    Z2Z2 = Z2 * Z2 % Q_MOD
    U1 = X1 * Z2Z2 % Q_MOD
    U2 = X2 * Z1Z1 % Q_MOD
    S1 = Y1 * Z2 * Z2Z2 % Q_MOD
    S2 = Y2 * Z1 * Z1Z1 % Q_MOD
    if U1 == U2:
        if S1 == S2:
            return _jac_double(P)
        return (1, 1, 0)
    H = (U2 - U1) % Q_MOD
    I = (2 * H) * (2 * H) % Q_MOD
    J = H * I % Q_MOD
    r = 2 * (S2 - S1) % Q_MOD
    V = U1 * I % Q_MOD
    X3 = (r * r - J - 2 * V) % Q_MOD
    Y3 = (r * (V - X3) - 2 * S1 * J) % Q_MOD
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % Q_MOD
    return (X3, Y3, Z3)


def jac_to_affine(P):
    X, Y, Z = P
    if Z % Q_MOD == 0:
        return None
    zi = pow(Z, -1, Q_MOD)
    zi2 = zi * zi % Q_MOD
    return (X * zi2 % Q_MOD, Y * zi2 * zi % Q_MOD)


def affine_to_jac(P):
    if P is None:
        return (1, 1, 0)
    return (P[0], P[1], 1)


def g1_mul(P, k: int):
    """k*P by double-and-add (k taken as a plain non-negative integer, like mul_bigint)."""
    acc = (1, 1, 0)
    if P is None or k == 0:
        return None
    base = affine_to_jac(P)
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == "1":
            acc = _jac_add(acc, base)
    return jac_to_affine(acc)


G1_GEN = (G1_X, G1_Y)


def msm_naive(bases, scalars):
    """sum_i scalars[i] * bases[i]; the definition that
    src/kzg/msm/variable_base.rs:182-194 (naive_var_base_msm) tests Pippenger against.
    Truncates to the shorter input like `msm_unchecked` (zip)."""
    acc = (1, 1, 0)
    for P, s in zip(bases, scalars):
        T = g1_mul(P, s)
        acc = _jac_add(acc, affine_to_jac(T))
    return jac_to_affine(acc)


# ----------------------------------------------------------------------------
# Field-vector helpers (src/misc.rs) over Fr, canonical ints
# ----------------------------------------------------------------------------
def fold_polynomial(f, r):
    """src/misc.rs:52-56: f' = [f[2i] + r*f[2i+1]], odd tail padded with 0."""
    out = []
    for i in range(0, len(f), 2):
        odd = f[i + 1] if i + 1 < len(f) else 0
        out.append((f[i] + r * odd) % R_MOD)
    return out


def powers(x, n):
    """src/misc.rs:59-65"""
    out = [1] * n
    for i in range(1, n):
        out[i] = out[i - 1] * x % R_MOD
    return out


def powers2(x, n):
    """src/misc.rs:68-77: [x, x^2, x^4, ...]"""
    out = [1] * n
    if n > 0:
        out[0] = x % R_MOD
    for i in range(1, n):
        out[i] = out[i - 1] * out[i - 1] % R_MOD
    return out


def tensor(elements):
    """src/misc.rs:133-149: tensor[sum b_j 2^j] = prod rho_j^{b_j}"""
    assert len(elements) > 0
    t = [1] * (1 << len(elements))
    t[1] = elements[0] % R_MOD
    for i in range(1, len(elements)):
        for j in range(1 << i):
            t[(1 << i) + j] = t[j] * elements[i] % R_MOD
    return t


def evaluate_be(poly, x):
    """src/misc.rs:180-190"""
    acc = 0
    for c in poly:
        acc = (acc * x + c) % R_MOD
    return acc


def evaluate_le(poly, x):
    """src/misc.rs:194-199"""
    return evaluate_be(reversed(poly), x)


def hadamard(a, b):
    """src/misc.rs:205-208 (asserts equal length)"""
    assert len(a) == len(b)
    return [x * y % R_MOD for x, y in zip(a, b)]


def ip(a, b):
    """src/misc.rs:215-218"""
    assert len(a) == len(b)
    return sum(x * y for x, y in zip(a, b)) % R_MOD


def linear_combination(polys, challenges):
    """src/misc.rs:37-48: sum_j c_j p_j padded to the longest, trailing zeros stripped
    (DensePolynomial::from_coefficients_vec truncates leading-zero high coefficients)."""
    if not polys or not challenges:
        return []
    n = max(len(p) for p, _ in zip(polys, challenges))
    out = [0] * n
    for p, c in zip(polys, challenges):
        for i, v in enumerate(p):
            out[i] = (out[i] + v * c) % R_MOD
    while out and out[-1] == 0:
        out.pop()
    return out


def vanishing_polynomial(points):
    """src/kzg/mod.rs:262-268: prod (x - p), little-endian coefficients."""
    poly = [1]
    for p in points:
        nxt = [0] * (len(poly) + 1)
        for i, c in enumerate(poly):
            nxt[i] = (nxt[i] - p * c) % R_MOD
            nxt[i + 1] = (nxt[i + 1] + c) % R_MOD
        poly = nxt
    return poly


def poly_divmod(f, z):
    """Schoolbook division of little-endian f by monic little-endian z -> (q, rem)."""
    f = list(f)
    dz = len(z) - 1
    assert z[-1] == 1
    if len(f) < len(z):
        return [], f
    q = [0] * (len(f) - dz)
    for i in range(len(f) - 1, dz - 1, -1):
        c = f[i]
        q[i - dz] = c
        if c:
            for j in range(dz + 1):
                f[i - dz + j] = (f[i - dz + j] - c * z[j]) % R_MOD
    return q, f[:dz]


# ----------------------------------------------------------------------------
# Sumcheck time prover (src/subprotocols/sumcheck/time_prover.rs)
# ----------------------------------------------------------------------------
def ceil_log2(n: int) -> int:
    """ark_std::log2 = ceil(log2(n)), with log2(0) = log2(1) = 0."""
    if n <= 1:
        return 0
    return (n - 1).bit_length()


class TimeProver:
    """Follows src/subprotocols/sumcheck/time_prover.rs:42-137 on canonical ints."""

    def __init__(self, f, g, twist):
        self.f = [x % R_MOD for x in f]
        self.g = [x % R_MOD for x in g]
        self.twist = twist % R_MOD
        self.round = 0
        self.tot_rounds = ceil_log2(max(len(self.f), len(self.g)))  # :35-38

    def fold(self, r):  # :75-80
        self.f = fold_polynomial(self.f, r * self.twist % R_MOD)
        self.g = fold_polynomial(self.g, r)
        self.twist = self.twist * self.twist % R_MOD

    def next_message(self, verifier_message=None):  # :83-123
        assert self.round <= self.tot_rounds
        if verifier_message is not None:
            self.fold(verifier_message)
        if self.round == self.tot_rounds:
            return None
        a = b = 0
        twist2 = self.twist * self.twist % R_MOD
        runner = 1
        npairs = min((len(self.f) + 1) // 2, (len(self.g) + 1) // 2)  # chunks(2).zip
        for i in range(npairs):
            fe = self.f[2 * i]
            ge = self.g[2 * i]
            fo = self.f[2 * i + 1] if 2 * i + 1 < len(self.f) else 0
            go = self.g[2 * i + 1] if 2 * i + 1 < len(self.g) else 0
            a = (a + fe * ge * runner) % R_MOD
            b = (b + (fe * go + ge * fo * self.twist) * runner) % R_MOD
            runner = runner * twist2 % R_MOD
        self.round += 1
        return (a, b)

    def final_foldings(self):  # :135-137
        if self.round == self.tot_rounds:
            return (self.f[0], self.g[0])
        return None


# ----------------------------------------------------------------------------
# MSM scalar recoding (src/kzg/msm/variable_base.rs:16-61)
# ----------------------------------------------------------------------------
def ln_without_floats(a: int) -> int:
    """variable_base.rs:16-19"""
    return ceil_log2(a) * 69 // 100


def arkworks_window(size: int) -> int:
    """variable_base.rs:105-109"""
    return 3 if size < 32 else ln_without_floats(size) + 2


def signed_digits(a: int, w: int, num_bits: int):
    """variable_base.rs:21-61 (make_digits)."""
    radix = 1 << w
    carry = 0
    if num_bits == 0:
        num_bits = a.bit_length()
    count = (num_bits + w - 1) // w
    digits = [0] * count
    for i in range(count):
        coef = carry + ((a >> (i * w)) & (radix - 1))
        carry = (coef + radix // 2) >> w
        digits[i] = coef - (carry << w)
    digits[count - 1] += carry << w
    return digits


def pippenger(bases, scalars):
    """variable_base.rs:99-176 restated literally (buckets, running sum, window Horner)."""
    pairs = list(zip(bases, scalars))
    size = len(pairs)
    if size == 0:
        return None
    c = arkworks_window(size)
    num_bits = 255
    digits_count = (num_bits + c - 1) // c
    sd = [signed_digits(s, c, num_bits) for _, s in pairs]
    window_sums = []
    for i in range(digits_count):
        buckets = [(1, 1, 0)] * (1 << c)
        for (P, _), d in zip(pairs, sd):
            s = d[i]
            if s > 0:
                buckets[s - 1] = _jac_add(buckets[s - 1], affine_to_jac(P))
            elif s < 0:
                buckets[-s - 1] = _jac_add(buckets[-s - 1], affine_to_jac(g1_neg(P)))
        running = (1, 1, 0)
        res = (1, 1, 0)
        for b in reversed(buckets):
            running = _jac_add(running, b)
            res = _jac_add(res, running)
        window_sums.append(res)
    total = window_sums[-1]
    for ws in reversed(window_sums[:-1]):
        for _ in range(c):
            total = _jac_double(total)
        total = _jac_add(total, ws)
    return jac_to_affine(total)


# ----------------------------------------------------------------------------
# Keccak-f[1600], STROBE-128 (as used by merlin 3.0.0), Merlin transcript
# ----------------------------------------------------------------------------
_KECCAK_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_KECCAK_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray) -> None:
    """FIPS-202 Keccak-p[1600,24] on a 200-byte little-endian lane state, in place."""
    A = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _KECCAK_ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _KECCAK_RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = (A[x][y] & _M64).to_bytes(8, "little")


def sha3_256(data: bytes) -> bytes:
    """SHA3-256 built on keccak_f1600 above -- exists only to check the permutation against hashlib."""
    rate = 136
    st = bytearray(200)
    padded = bytearray(data) + b"\x06"
    while len(padded) % rate:
        padded += b"\x00"
    padded[-1] |= 0x80
    for off in range(0, len(padded), rate):
        for i in range(rate):
            st[i] ^= padded[off + i]
        keccak_f1600(st)
    return bytes(st[:32])


class Strobe128:
    """STROBE-128/1600 subset used by merlin (meta-AD, AD, PRF, KEY), per the STROBE v1.0.2 spec."""

    R = 166
    FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32

    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[self.R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert not (flags & self.FLAG_T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        force_f = (flags & (self.FLAG_C | self.FLAG_K)) != 0
        if force_f and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_M | self.FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(self.FLAG_I | self.FLAG_A | self.FLAG_C, more)
        return self._squeeze(n)


class MerlinTranscript:
    """merlin 3.0.0 `Transcript`: new / append_message / challenge_bytes."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


# ----------------------------------------------------------------------------
# Gemini transcript conventions (src/transcript.rs) + ark-serialize framing (recalled)
# ----------------------------------------------------------------------------
PROTOCOL_NAME = b"GEMINI-v0"  # src/lib.rs:74


def fr_serialize(a: int) -> bytes:
    """ark-serialize of a prime-field element: canonical value, 32 bytes little-endian."""
    return (a % R_MOD).to_bytes(32, "little")


def fr_from_random_bytes(b: bytes):
    """ark-ff `Fp::from_random_bytes` (0.4): take the first 32 bytes little-endian, mask the
    top bit(s) above the 255-bit modulus size, accept iff < r.  [recalled, see module header]"""
    v = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    return v if v < R_MOD else None


def g1_serialize_uncompressed(P) -> bytes:
    """ark-serialize default short-Weierstrass uncompressed framing as used by
    ark-test-curves' bls12_381 (x LE || y LE, flags in the top bits of the last byte;
    infinity = all-zero with bit 6 set).  [recalled -- unverifiable here, isolated on purpose]"""
    if P is None:
        out = bytearray(96)
        out[95] |= 1 << 6
        return bytes(out)
    x, y = P
    out = bytearray(x.to_bytes(48, "little") + y.to_bytes(48, "little"))
    if y > (Q_MOD - y) % Q_MOD:  # "negative" y flag (bit 7)
        out[95] |= 1 << 7
    return bytes(out)


class GeminiTranscript(MerlinTranscript):
    """src/transcript.rs:16-34 on top of Merlin."""

    def append_serializable_bytes(self, label: bytes, message: bytes):
        self.append_message(label, message)

    def append_fr(self, label: bytes, a: int):
        self.append_message(label, fr_serialize(a))

    def append_round_msg(self, label: bytes, a: int, b: int):
        # RoundMsg(a, b) derives CanonicalSerialize: a || b  (sumcheck/prover.rs:9-10)
        self.append_message(label, fr_serialize(a) + fr_serialize(b))

    def get_challenge(self, label: bytes) -> int:
        while True:  # src/transcript.rs:26-34
            v = fr_from_random_bytes(self.challenge_bytes(label, 64))
            if v is not None:
                return v


def sumcheck_prove(transcript: GeminiTranscript, prover):
    """src/subprotocols/sumcheck/proof.rs:36-66.  `prover` needs next_message / final_foldings."""
    messages, challenges = [], []
    vm = None
    while True:
        msg = prover.next_message(vm)
        if msg is None:
            break
        transcript.append_round_msg(b"evaluations", msg[0], msg[1])
        vm = transcript.get_challenge(b"challenge")
        messages.append(msg)
        challenges.append(vm)
    ff = prover.final_foldings()
    transcript.append_fr(b"final-folding", ff[0])
    transcript.append_fr(b"final-folding", ff[1])
    return messages, challenges, ff


class HerringTimeProver:
    """src/herring/time_prover.rs:42-137 over a bilinear module given by (p, add, zero, scale_lhs):
    module "F":  Lhs = Rhs = Target = Fr (ints);   module "G1": Lhs = Target = G1 affine points, Rhs = Fr."""

    def __init__(self, module, f, g, twist):
        self.module = module
        self.f = list(f)
        self.g = [x % R_MOD for x in g]
        self.twist = twist % R_MOD
        self.round = 0
        self.tot_rounds = ceil_log2(min(len(self.f), len(self.g)))  # :36-39

    def _ip(self, lhs, rhs):
        if self.module == "F":
            return sum(a * b for a, b in zip(lhs, rhs)) % R_MOD
        return msm_naive(list(lhs)[: len(list(rhs))] if False else list(lhs), list(rhs))

    def _split_fold_lhs(self, r):
        out = []
        for i in range(0, len(self.f), 2):
            lo = self.f[i]
            hi = self.f[i + 1] if i + 1 < len(self.f) else None
            if self.module == "F":
                out.append((lo + (hi or 0) * r) % R_MOD)
            else:
                out.append(g1_add(lo, g1_mul(hi, r)) if hi is not None else lo)
        return out

    def fold(self, r):  # :82-88
        self.f = self._split_fold_lhs(r * self.twist % R_MOD)
        self.g = fold_polynomial(self.g, r)
        self.twist = self.twist * self.twist % R_MOD

    def next_message(self, vm=None):  # :91-123
        if vm is not None:
            self.fold(vm)
        if self.round == self.tot_rounds:
            return None
        fe, fo = self.f[0::2], self.f[1::2]
        ge, go = self.g[0::2], self.g[1::2]
        a = self._ip(fe, ge)
        b1, b2 = self._ip(fe, go), self._ip(fo, ge)
        b = (b1 + b2) % R_MOD if self.module == "F" else g1_add(b1, b2)
        self.round += 1
        return (a, b)

    def final_foldings(self):
        return (self.f[0], self.g[0]) if self.round == self.tot_rounds else None


def sumcheck_prove_batch(transcript: GeminiTranscript, provers):
    """src/subprotocols/sumcheck/proof.rs:69-122"""
    rounds = max(p.tot_rounds for p in provers) + 1
    coefficients = [transcript.get_challenge(b"batch-sumcheck") for _ in provers]
    messages, challenges = [], []
    vm = None
    for _ in range(rounds):
        a = b = 0
        for p, c in zip(provers, coefficients):
            m = p.next_message(vm)
            if m is None:
                ff = p.final_foldings()
                m = (ff[0] * ff[1] % R_MOD, 0)
            a = (a + m[0] * c) % R_MOD
            b = (b + m[1] * c) % R_MOD
        transcript.append_round_msg(b"evaluations", a, b)
        vm = transcript.get_challenge(b"challenge")
        messages.append((a, b))
        challenges.append(vm)
    finals = []
    for p in provers:
        ff = p.final_foldings()
        transcript.append_fr(b"final-folding-lhs", ff[0])
        transcript.append_fr(b"final-folding-rhs", ff[1])
        finals.append(ff)
    return messages, challenges, finals


# ----------------------------------------------------------------------------
# Deterministic test-input generator shared by oracle, tests and bench
# (the reference draws from ark_std::test_rng(), which is not reproducible without Rust)
# ----------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def fr(self) -> int:
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < R_MOD:
                return v

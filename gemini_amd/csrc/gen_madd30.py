#!/usr/bin/env python3
"""Emits g1_madd30_gen.inc: the WHOLE XYZZ mixed addition of k_acc0 as one asm statement for gfx950.

Why: k_acc0 is bound by integer issue, and ~20 % of every Fq product of the canonical (12 x 32-bit) element
layer is conversion -- unpack both operands to 13 x 30-bit limbs, repack, conditional subtraction
(gen_field_mul30.py) -- plus ~870 argument / result moves per addition around the out-of-line calls.  Here the
bucket accumulator LIVES in the product's own representation across iterations:

  * acc = (X, Y, ZZ, ZZZ), 4 x 13 limbs of 30 bits in 52 pinned VGPRs, "loose": a value v < 2^386 stands for
    v mod q, Montgomery factor 2^390 (field30.cuh).  A Montgomery product of loose values is < q + ab / 2^390
    with NO conditional subtraction; a - b is a + K q - b with a carry pass (K q's limbs raised so no limb goes
    negative).  Bounds are tracked by the generator (class V below) and asserted.
  * the base point arrives as the 24 canonical words k_acc0 loads, INSIDE two of the temporary elements, and is
    unpacked in place (2 x 24 instructions); the statement therefore consumes them (in/out operands).
  * 6 products + 2 squares + ONE fused double product: Y3 = R (Q - X3) + (2q - Y1) PPP accumulates both
    13 x 13 products into the same columns in front of a single Montgomery reduction (-182 multiply-adds).
  * every register is physical and statically renamed: no moves between products, no calls.

EFD madd-2008-s; the exceptional cases:
  * acc == identity (ZZ limbs all zero): the lane is set to (x, y, 1, 1) up front and sits out the arithmetic
    (EXEC mask) -- no divergent branch around 4 000 instructions in the kernel.
  * a negative digit adds -P: y is negated on those lanes after the unpack (13-limb q - y), inside the statement.
  * P = x2 ZZ1 - X1 == 0 (mod q) (doubling / cancellation; the reference's elastic benchmark makes EVERY base the
    generator, examples/snark.rs:59-63): PP = P^2 is then exactly q (or 0), so one compare of PP's low limb
    against q_0 and 0 filters (false positives 2^-29); a wave with a match branches to a cold block behind the end
    of the statement: exact test, then for the lanes concerned acc = 2 acc (dbl-2008-s-1 on the accumulator, still
    untouched at that point) when R == 0 as well, acc = identity otherwise; the other lanes resume the normal law.
    The statement is therefore complete and always returns flag = 0 (the output is kept for the wrapper's shape).

The generator INTERPRETS the instruction list it emits against big-integer arithmetic (`--selftest`), bounds
included (64-bit column accumulators, 32-bit limb sums), so mistakes show up without a GPU.

Run:  python3 gen_madd30.py > g1_madd30_gen.inc
"""
import random
import sys

from gen_field_mul30 import INV30, M32, MASK30, P30, Q, Q32, Prog

R390 = 1 << 390
LIMIT = 1 << 386          # every loose value stays below this (top limb < 2^26)

# ---- register plan (64-bit and wider VGPR operands must start at an even register on gfx90a+) ---------
# v[VB .. VB+51]   acc: X, Y, ZZ, ZZZ (13 limbs each)                         in/out
# E0, E1           two temporary elements; the base's x / y words ARRIVE in E0[1..12] / E1[1..12] and are unpacked in place
# E2, E3           two more temporary elements; accumulators, scratch, flag, sign
VB = 48
ACC_X = [f"v{VB + i}" for i in range(13)]
ACC_Y = [f"v{VB + 13 + i}" for i in range(13)]
ACC_ZZ = [f"v{VB + 26 + i}" for i in range(13)]
ACC_ZZZ = [f"v{VB + 39 + i}" for i in range(13)]
CAR = f"v{VB + 52}"
E0B, E1B, E2B, E3B = VB + 53, VB + 67, VB + 80, VB + 93
TMP = f"v{VB + 66}"
E = [[f"v{b + i}" for i in range(13)] for b in (E0B, E1B, E2B, E3B)]
BXW = E[0][1:]            # 12 words, first register even
BYW = E[1][1:]
assert (E0B + 1) % 2 == 0 and (E1B + 1) % 2 == 0
ACCUM = VB + 106          # v[ACCUM:ACCUM+1]  column accumulator
SPL = VB + 108            # v[SPL:SPL+1]      split-off high part of a column
SPT = VB + 110            # v[SPT:SPT+1]      second split of the same column
FLAG = f"v{VB + 112}"
SGN = f"v{VB + 113}"      # in: != 0 -> add the NEGATED base
V_END = VB + 114          # first register NOT used
assert ACCUM % 2 == 0
# scalar registers
SB = 40
SP = [f"s{SB + i}" for i in range(13)]     # q in radix 2^30
SINV = f"s{SB + 13}"
S_IDENT = SB + 14         # s[54:55]: lanes whose accumulator is the identity
S_SAVE = SB + 16          # s[56:57]: EXEC at entry
S_TMP = SB + 18           # s[58:59]
S_TMP2 = SB + 20          # s[60:61]
S_EXC = SB + 22           # s[62:63]: lanes with p == 0 (mod q)
S_ARITH = SB + 24         # s[64:65]: the lanes doing arithmetic (not identity) while the exceptional ones are handled
S_END = SB + 26
ONE30 = [(R390 % Q >> (30 * i)) & MASK30 for i in range(13)]


class Asm(Prog):
    """Prog + the few extra opcodes of the group law (sign-free carries, masks, EXEC handling)"""

    def text(self):
        out = []
        o = self._o
        plain = Prog()
        for ins in self.ins:
            op = ins[0]
            if op == "mad64":
                d, x, y = ins[1:]
                out.append(f"v_mad_u64_u32 v[{d}:{d + 1}], vcc, {o(x)}, {o(y)}, v[{d}:{d + 1}]")
            elif op == "or3":
                out.append(f"v_or3_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}, {o(ins[4])}")
            elif op == "or":
                out.append(f"v_or_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}")
            elif op == "xor":
                out.append(f"v_xor_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}")
            elif op == "s_and":
                out.append(f"s_and_b64 s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}], s[{ins[3]}:{ins[3] + 1}]")
            elif op == "s_mov64":
                out.append(f"s_mov_b64 s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}]")
            elif op == "exec_and":      # exec = s[a] & s[b]
                out.append(f"s_and_b64 exec, s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}]")
            elif op == "cmp_eq_s":      # s[d:d+1] = (a == b) per active lane
                out.append(f"v_cmp_eq_u32 s[{ins[1]}:{ins[1] + 1}], {o(ins[2])}, {o(ins[3])}")
            elif op == "s_or":
                out.append(f"s_or_b64 s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}], s[{ins[3]}:{ins[3] + 1}]")
            elif op == "s_andn2":
                out.append(f"s_andn2_b64 s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}], s[{ins[3]}:{ins[3] + 1}]")
            elif op == "s_nop":
                out.append(f"s_nop {ins[1]}")
            elif op == "cbranch_s_z":   # branch if s[a:a+1] == 0
                out.append(f"s_cmp_eq_u64 s[{ins[1]}:{ins[1] + 1}], 0")
                out.append(f"s_cbranch_scc1 {ins[2]}f")
            elif op == "branch":
                out.append(f"s_branch {ins[1]}f")
            elif op == "branch_b":      # backward
                out.append(f"s_branch {ins[1]}b")
            elif op == "cbranch_s_nz":  # branch (forward) if s[a:a+1] != 0
                out.append(f"s_cmp_lg_u64 s[{ins[1]}:{ins[1] + 1}], 0")
                out.append(f"s_cbranch_scc1 {ins[2]}f")
            elif op == "cbranch_s_z_b":
                out.append(f"s_cmp_eq_u64 s[{ins[1]}:{ins[1] + 1}], 0")
                out.append(f"s_cbranch_scc1 {ins[2]}b")
            elif op == "cbranch_execz_b":
                out.append(f"s_cbranch_execz {ins[1]}b")
            elif op == "saveexec_and":  # save = exec; exec &= s[m]
                out.append(f"s_and_saveexec_b64 s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}]")
            elif op == "cbranch_execz":
                out.append(f"s_cbranch_execz {ins[1]}f")
            elif op == "restore_exec":
                out.append(f"s_mov_b64 exec, s[{ins[1]}:{ins[1] + 1}]")
            elif op == "exec_andn2":    # exec = s[a] & ~s[b]
                out.append(f"s_andn2_b64 exec, s[{ins[1]}:{ins[1] + 1}], s[{ins[2]}:{ins[2] + 1}]")
            elif op == "cmp_ne_s":
                out.append(f"v_cmp_ne_u32 s[{ins[1]}:{ins[1] + 1}], {o(ins[2])}, {o(ins[3])}")
            elif op == "label":
                out.append(f"{ins[1]}:")
            else:
                plain.ins = [ins]
                out.extend(plain.text())
        return out

    def run(self, regs):
        """one lane; regs also holds 's<N>' scalars; 64-bit scalar pairs are stored as 0 / 1 in s<lo>"""
        def g(x):
            return x & M32 if isinstance(x, int) else regs[x]

        labels = {ins[1]: i for i, ins in enumerate(self.ins) if ins[0] == "label"}
        ex = 1
        vcc = 0
        pc = 0
        n = len(self.ins)
        while pc < n:
            ins = self.ins[pc]
            pc += 1
            op = ins[0]
            # ---- scalar / control -----------------------------------------------------------------
            if op == "smov":
                regs[ins[1]] = g(ins[2])
            elif op == "s_or":
                regs[f"s{ins[1]}"] = regs[f"s{ins[2]}"] | regs[f"s{ins[3]}"]
            elif op == "s_andn2":
                regs[f"s{ins[1]}"] = regs[f"s{ins[2]}"] & ~regs[f"s{ins[3]}"] & 1
            elif op == "s_and":
                regs[f"s{ins[1]}"] = regs[f"s{ins[2]}"] & regs[f"s{ins[3]}"]
            elif op == "s_mov64":
                regs[f"s{ins[1]}"] = regs[f"s{ins[2]}"]
            elif op == "exec_and":
                ex = regs[f"s{ins[1]}"] & regs[f"s{ins[2]}"] & 1
            elif op == "s_nop" or op == "label":
                pass
            elif op == "cbranch_s_z":
                if regs[f"s{ins[1]}"] == 0:
                    pc = labels[ins[2]]
            elif op in ("branch", "branch_b"):
                pc = labels[ins[1]]
            elif op == "cbranch_s_nz":
                if regs[f"s{ins[1]}"] != 0:
                    pc = labels[ins[2]]
            elif op == "cbranch_s_z_b":
                if regs[f"s{ins[1]}"] == 0:
                    pc = labels[ins[2]]
            elif op == "cbranch_execz_b":
                if ex == 0:
                    pc = labels[ins[1]]
            elif op == "saveexec_and":
                regs[f"s{ins[1]}"] = ex
                ex = ex & regs[f"s{ins[2]}"]
            elif op == "cbranch_execz":
                if ex == 0:
                    pc = labels[ins[1]]
            elif op == "restore_exec":
                ex = regs[f"s{ins[1]}"]
            elif op == "exec_andn2":
                ex = regs[f"s{ins[1]}"] & ~regs[f"s{ins[2]}"] & 1
            elif op == "cmp_ne_s":
                regs[f"s{ins[1]}"] = int(g(ins[2]) != g(ins[3])) if ex else 0
            elif op == "cmp_eq_s":
                regs[f"s{ins[1]}"] = int(g(ins[2]) == g(ins[3])) if ex else 0
            elif not ex:
                continue
            # ---- vector ---------------------------------------------------------------------------
            elif op == "mad64":
                d, x, y = ins[1:]
                acc = regs[f"v{d}"] | (regs[f"v{d + 1}"] << 32)
                acc += g(x) * g(y)
                assert acc < (1 << 64), "64-bit column accumulator overflow"
                regs[f"v{d}"] = acc & M32
                regs[f"v{d + 1}"] = acc >> 32
                vcc = 0
            elif op == "mul_lo":
                regs[ins[1]] = (g(ins[2]) * g(ins[3])) & M32
            elif op == "and":
                regs[ins[1]] = g(ins[2]) & g(ins[3])
            elif op == "or":
                regs[ins[1]] = g(ins[2]) | g(ins[3])
            elif op == "or3":
                regs[ins[1]] = g(ins[2]) | g(ins[3]) | g(ins[4])
            elif op == "xor":
                regs[ins[1]] = g(ins[2]) ^ g(ins[3])
            elif op == "lshr":
                regs[ins[1]] = g(ins[3]) >> (g(ins[2]) & 31)
            elif op == "lshl":
                v = g(ins[3]) << (g(ins[2]) & 31)
                assert v <= M32, "left shift drops bits"
                regs[ins[1]] = v
            elif op == "add":
                s = g(ins[2]) + g(ins[3])
                assert s <= M32, "32-bit limb sum wraps"
                regs[ins[1]] = s
            elif op == "sub":
                s = g(ins[2]) - g(ins[3])
                assert s >= 0, "32-bit limb difference negative"
                regs[ins[1]] = s
            elif op == "alignbit":
                regs[ins[1]] = (((g(ins[2]) << 32) | g(ins[3])) >> (g(ins[4]) & 31)) & M32
            elif op == "lshr64":
                d, sh, a = ins[1:]
                v = (regs[f"v{a}"] | (regs[f"v{a + 1}"] << 32)) >> (g(sh) & 63)
                regs[f"v{d}"], regs[f"v{d + 1}"] = v & M32, v >> 32
            elif op == "mov":
                regs[ins[1]] = g(ins[2])
            elif op == "add_co":
                s = g(ins[2]) + g(ins[3])
                regs[ins[1]], vcc = s & M32, s >> 32
            elif op == "addc_co":
                s = g(ins[2]) + g(ins[3]) + vcc
                assert s <= M32, "64-bit split sum wraps"
                regs[ins[1]], vcc = s & M32, s >> 32
            else:
                raise ValueError(op)
        return regs


class V:
    """a 13-limb element in registers with an upper bound on its integer value"""

    def __init__(self, regs, bound, norm=True):
        self.r = regs
        self.bound = bound
        assert bound <= LIMIT, "loose value bound exceeds 2^386"

    def lb(self, i):
        """upper bound of limb i (normalised limbs)"""
        return MASK30 if i < 12 else min((1 << 26) - 1, self.bound >> 360)


def raised(kq):
    """K q with every limb below the top raised by 2^30, the borrow taken from the next limb (field30.cuh)"""
    l = [(kq >> (30 * i)) & MASK30 for i in range(12)] + [kq >> 360]
    m = [l[0] + (1 << 30)] + [l[i] + (1 << 30) - 1 for i in range(1, 12)] + [l[12] - 1]
    assert sum(m[i] << (30 * i) for i in range(13)) == kq and m[12] >= 0
    return m


class Plan:
    """scratch registers of the element operations: column accumulator pair, two split pairs, two single temporaries"""

    def __init__(self, accum, spl, spt, tmp, car):
        self.ACCUM, self.SPL, self.SPT, self.TMP, self.CAR = accum, spl, spt, tmp, car


class Gen:
    def __init__(self, plan=None):
        self.p = Asm()
        self.pl = plan or Plan(ACCUM, SPL, SPT, TMP, CAR)

    # ---- element ops -----------------------------------------------------------------------------
    def unpack(self, words, dst):
        """12 canonical 32-bit words -> 13 limbs; words may be dst[1..12] (in place: limb i is written after its last use as a word)"""
        assert words == dst[1:] or not set(words) & set(dst)
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        p.emit("and", dst[0], MASK30, words[0])
        for i in range(1, 12):
            p.emit("alignbit", dst[i], words[i], words[i - 1], 32 - 2 * i)
            p.emit("and", dst[i], MASK30, dst[i])
        p.emit("lshr", dst[12], 8, words[11])
        return V(dst, Q - 1)

    def sub(self, a, b, k, dst=None):
        """dst = a - b + k q  (b < k q), normalised; dst may alias a"""
        assert b.bound <= k * Q, "subtrahend may exceed k q"
        m = raised(k * Q)
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        dst = dst or a.r
        for i in range(13):
            p.emit("add", TMP, m[i], a.r[i])
            p.emit("sub", TMP, TMP, b.r[i])
            if i > 0:
                p.emit("add", TMP, TMP, CAR)
            if i < 12:
                p.emit("and", dst[i], MASK30, TMP)
                p.emit("lshr", CAR, 30, TMP)
            else:
                p.emit("mov", dst[i], TMP)
        return V(dst, a.bound + k * Q)

    def rsub(self, b, k, dst=None):
        """dst = k q - b  (b < k q), normalised; dst may alias b"""
        assert b.bound <= k * Q
        m = raised(k * Q)
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        dst = dst or b.r
        for i in range(13):
            p.emit("sub", TMP, m[i], b.r[i])
            if i > 0:
                p.emit("add", TMP, TMP, CAR)
            if i < 12:
                p.emit("and", dst[i], MASK30, TMP)
                p.emit("lshr", CAR, 30, TMP)
            else:
                p.emit("mov", dst[i], TMP)
        return V(dst, k * Q)

    def add(self, a, b, dst):
        """dst = a + b, normalised; dst may alias a or b"""
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        for i in range(13):
            p.emit("add", TMP, a.r[i], b.r[i])
            if i > 0:
                p.emit("add", TMP, TMP, CAR)
            if i < 12:
                p.emit("and", dst[i], MASK30, TMP)
                p.emit("lshr", CAR, 30, TMP)
            else:
                p.emit("mov", dst[i], TMP)
        return V(dst, a.bound + b.bound)

    def is_zero_mod_q(self, v, sdst, stmp):
        """s[sdst] = lanes where the product output v (normalised, < 2 q) is 0 mod q, i.e. exactly 0 or exactly q"""
        assert v.bound < 2 * Q
        p = self.p
        TMP, CAR = self.pl.TMP, self.pl.CAR
        p.emit("or3", TMP, v.r[0], v.r[1], v.r[2])
        for i in range(3, 13, 2):
            p.emit("or3", TMP, TMP, v.r[i], v.r[i + 1])
        p.emit("cmp_eq_s", sdst, 0, TMP)                       # == 0
        p.emit("xor", TMP, SP[0], v.r[0])
        for i in range(1, 13):
            p.emit("xor", CAR, SP[i], v.r[i])
            p.emit("or", TMP, TMP, CAR)
        p.emit("cmp_eq_s", stmp, 0, TMP)                       # == q
        p.emit("s_nop", 4)
        p.emit("s_or", sdst, sdst, stmp)

    def dbl(self, a, dst):
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        for i in range(13):
            p.emit("lshl", TMP, 1, a.r[i])
            if i > 0:
                p.emit("add", TMP, TMP, CAR)
            if i < 12:
                p.emit("and", dst[i], MASK30, TMP)
                p.emit("lshr", CAR, 30, TMP)
            else:
                p.emit("mov", dst[i], TMP)
        return V(dst, 2 * a.bound)

    def mont(self, prods, m_regs, t_regs, sq_tmp=None):
        """t = (sum of a_i * b_i over `prods`) * 2^-390 mod q, loose.

        prods: list of (a, b) with V operands; (a, None) = square of a (needs sq_tmp: 13 registers for 2 a_j).
        m_regs: 13 registers for the quotient digits m_k; t_regs: the result (may equal m_regs, or the `a` registers of
        the LAST product: t_j is written in column j + 13, a_j / m_j are last read in column j + 12)."""
        p = self.p
        TMP, CAR, ACCUM, SPL, SPT = self.pl.TMP, self.pl.CAR, self.pl.ACCUM, self.pl.SPL, self.pl.SPT
        total = 0
        for a, b in prods:
            total += a.bound * (a.bound if b is None else b.bound)
        out_bound = Q + total // R390 + 1
        dbl_regs = None
        if any(b is None for _, b in prods):
            assert sum(1 for _, b in prods if b is None) == 1 and sq_tmp is not None
            a = next(a for a, b in prods if b is None)
            dbl_regs = sq_tmp
            for j in range(1, 13):
                p.emit("lshl", dbl_regs[j], 1, a.r[j])
        p.emit("mov", f"v{ACCUM}", 0)
        p.emit("mov", f"v{ACCUM + 1}", 0)
        st = {"bound": 0, "nsplit": 0, "spl_bound": 0}

        def split():
            # move the accumulator's bits >= 30 aside so that the running sum restarts below 2^30
            if st["nsplit"] == 0:
                p.emit("lshr64", SPL, 30, ACCUM)
            else:
                p.emit("lshr64", SPT, 30, ACCUM)
                p.emit("add_co", f"v{SPL}", f"v{SPL}", f"v{SPT}")
                p.emit("addc_co", f"v{SPL + 1}", f"v{SPL + 1}", f"v{SPT + 1}")
            p.emit("and", f"v{ACCUM}", MASK30, f"v{ACCUM}")
            p.emit("mov", f"v{ACCUM + 1}", 0)
            st["spl_bound"] += st["bound"] >> 30
            st["bound"] = MASK30
            st["nsplit"] += 1

        def mad(x, y, bx, by):
            if st["bound"] + bx * by >= (1 << 64):
                split()
            st["bound"] += bx * by
            assert st["bound"] < (1 << 64)
            p.emit("mad64", ACCUM, x, y)

        for k in range(25):
            lo_i, hi_i = max(0, k - 12), min(k, 12)
            for a, b in prods:
                if b is None:
                    for i in range(lo_i, hi_i + 1):
                        j = k - i
                        if i < j:
                            mad(a.r[i], dbl_regs[j], a.lb(i), 2 * a.lb(j))
                        elif i == j:
                            mad(a.r[i], a.r[i], a.lb(i), a.lb(i))
                else:
                    for i in range(lo_i, hi_i + 1):
                        mad(a.r[i], b.r[k - i], a.lb(i), b.lb(k - i))
            red = [(i, k - i) for i in (range(0, k) if k < 13 else range(k - 12, 13))]
            for i, j in red:
                mad(m_regs[i], SP[j], MASK30, P30[j])
            if k < 13:
                p.emit("mul_lo", m_regs[k], f"v{ACCUM}", SINV)
                p.emit("and", m_regs[k], MASK30, m_regs[k])
                mad(m_regs[k], SP[0], MASK30, P30[0])
            else:
                p.emit("and", t_regs[k - 13], MASK30, f"v{ACCUM}")
            p.emit("lshr64", ACCUM, 30, ACCUM)
            st["bound"] >>= 30
            if st["nsplit"]:
                p.emit("add_co", f"v{ACCUM}", f"v{ACCUM}", f"v{SPL}")
                p.emit("addc_co", f"v{ACCUM + 1}", f"v{ACCUM + 1}", f"v{SPL + 1}")
                st["bound"] += st["spl_bound"]
                st["nsplit"] = 0
                st["spl_bound"] = 0
        p.emit("mov", t_regs[12], f"v{ACCUM}")
        return V(t_regs, out_bound)

    # ---- the group law -----------------------------------------------------------------------------
    def double_acc(self, X, Y, ZZ, ZZZ, E=None):
        """acc = 2 acc, EFD dbl-2008-s-1 (a = 0), in place; Y != 0 mod q for a point of odd order.  Uses four temporaries."""
        E = E or globals()["E"]
        U = self.dbl(Y, E[0])                                   # u = 2 Y
        Vv = self.mont([(U, None)], E[1], E[1], sq_tmp=E[2])    # v = u^2
        W = self.mont([(U, Vv)], E[2], E[2])                    # w = u v               (u dead)
        S = self.mont([(X, Vv)], E[3], E[3])                    # s = X v
        ZZn = self.mont([(ZZ, Vv)], E[0], ACC_ZZ)          # ZZ3 = v ZZ            (in place; v dead)
        ZZZn = self.mont([(ZZZ, W)], E[0], ACC_ZZZ)             # ZZZ3 = w ZZZ          (in place)
        XX = self.mont([(X, None)], E[0], E[0], sq_tmp=E[1])    # X^2                   (X dead)
        D2 = self.dbl(XX, E[1])
        M = self.add(D2, XX, E[1])                              # m = 3 X^2
        M2 = self.mont([(M, None)], ACC_X, ACC_X, sq_tmp=E[0])  # m^2 into X's registers
        DS = self.dbl(S, E[0])
        X3 = self.sub(M2, DS, 3)                                # X3 = m^2 - 2 s
        T = self.sub(S, X3, 5)                                  # s - X3                (E3)
        NY = self.rsub(Y, 2)                                    # 2 q - Y               (in place)
        Y3 = self.mont([(M, T), (W, NY)], E[0], ACC_Y)          # Y3 = m (s - X3) - w Y, one reduction
        assert X3.bound <= X.bound and Y3.bound <= Y.bound and ZZn.bound <= ZZ.bound and ZZZn.bound <= ZZZ.bound
        return X3, Y3, ZZn, ZZZn

    def madd(self):
        p = self.p
        for j in range(13):
            p.emit("smov", SP[j], P30[j])
        p.emit("smov", SINV, INV30)
        p.emit("mov", FLAG, 0)
        # invariant bounds of the accumulator at entry (checked against what leaves, below)
        BX, BY, BZ = 7 * Q, 2 * Q, 2 * Q
        X, Y, ZZ, ZZZ = V(ACC_X, BX), V(ACC_Y, BY), V(ACC_ZZ, BZ), V(ACC_ZZZ, BZ)
        qx = self.unpack(BXW, E[0])
        qy = self.unpack(BYW, E[1])
        # negative digit: y2 = q - y2 on the lanes that ask for it (y2 != 0: the curve has no point of order 2)
        p.emit("cmp_ne_s", S_TMP, 0, SGN)
        p.emit("s_nop", 4)
        p.emit("saveexec_and", S_SAVE, S_TMP)
        p.emit("cbranch_execz", "1")
        self.rsub(qy, 1)
        p.emit("label", "1")
        p.emit("restore_exec", S_SAVE)
        qy = V(E[1], Q)
        # identity lanes (ZZ == 0 exactly): acc = (x2, y2, 1, 1); they sit out the arithmetic below
        p.emit("or3", TMP, ACC_ZZ[0], ACC_ZZ[1], ACC_ZZ[2])
        for i in range(3, 13, 2):
            p.emit("or3", TMP, TMP, ACC_ZZ[i], ACC_ZZ[i + 1])
        p.emit("cmp_eq_s", S_IDENT, 0, TMP)
        p.emit("s_nop", 4)
        p.emit("saveexec_and", S_SAVE, S_IDENT)
        p.emit("cbranch_execz", "2")
        for i in range(13):
            p.emit("mov", ACC_X[i], qx.r[i])
            p.emit("mov", ACC_Y[i], qy.r[i])
            p.emit("mov", ACC_ZZ[i], ONE30[i])
            p.emit("mov", ACC_ZZZ[i], ONE30[i])
        p.emit("label", "2")
        p.emit("exec_andn2", S_SAVE, S_IDENT)
        p.emit("cbranch_execz", "9")
        U = self.mont([(qx, ZZ)], E[2], E[2])                  # u2 = x2 ZZ1
        P = self.sub(U, X, 7)                                  # p = u2 - X1            (E2)
        S = self.mont([(qy, ZZZ)], E[3], E[3])                 # s2 = y2 ZZZ1
        R = self.sub(S, Y, 2)                                  # r = s2 - Y1            (E3)
        PP = self.mont([(P, None)], E[0], E[0], sq_tmp=E[1])   # pp = p^2               (E0)
        assert PP.bound < 2 * Q
        # p == 0 (mod q)  <=>  pp in {0, q}.  One compare of the low limb filters (2^-29 false positives); a wave with a
        # match takes the exact test and, for the lanes that really have x2 ZZ1 == X1, the exceptional law right here:
        # r == 0 as well -> acc = 2 acc (dbl-2008-s-1 on the accumulator, which still holds its input), else acc = identity.
        # The reference's elastic benchmark makes EVERY base the generator (examples/snark.rs:59-63), so the second
        # entry of every run is a doubling; lanes reach it at different iterations and a wave would otherwise leave the
        # statement on almost every iteration.
        p.emit("cmp_eq_s", S_TMP, SP[0], PP.r[0])
        p.emit("cmp_eq_s", S_TMP2, 0, PP.r[0])
        p.emit("s_nop", 4)
        p.emit("s_or", S_TMP, S_TMP, S_TMP2)
        p.emit("cbranch_s_nz", S_TMP, "6")                     # the cold block sits behind the end of the statement
        p.emit("label", "3")
        ZZn = self.mont([(ZZ, PP)], E[1], ACC_ZZ)              # ZZ3 = ZZ1 pp           (in place)
        PPP = self.mont([(P, PP)], E[1], E[1])                 # ppp = p pp             (E1)
        ZZZn = self.mont([(ZZZ, PPP)], E[2], ACC_ZZZ)          # ZZZ3 = ZZZ1 ppp        (in place; P dead)
        QQ = self.mont([(X, PP)], E[2], E[2])                  # q = X1 pp              (E2; X, pp dead)
        R2 = self.mont([(R, None)], ACC_X, ACC_X, sq_tmp=E[0])  # r^2 into X's registers
        X3a = self.sub(R2, PPP, 2)
        D = self.dbl(QQ, E[0])
        X3 = self.sub(X3a, D, 3)                               # X3 = r^2 - ppp - 2 q
        T = self.sub(QQ, X3, 7)                                # q - X3                 (E2)
        NY = self.rsub(Y, 2)                                   # 2 q - Y1               (in place)
        Y3 = self.mont([(R, T), (PPP, NY)], E[0], ACC_Y)       # Y3 = r (q - X3) - Y1 ppp, ONE reduction (in place on NY)
        assert X3.bound <= BX and Y3.bound <= BY and ZZn.bound <= BZ and ZZZn.bound <= BZ, (X3.bound / Q, Y3.bound / Q, ZZn.bound / Q)
        self.bounds = {"X": X3.bound / Q, "Y": Y3.bound / Q, "ZZ": ZZn.bound / Q, "P": P.bound / Q, "T": T.bound / Q}
        p.emit("label", "9")
        p.emit("restore_exec", S_SAVE)
        p.emit("branch", "7")
        # ---- cold: some lane's pp has the low limb of 0 or q
        p.emit("label", "6")
        self.is_zero_mod_q(PP, S_EXC, S_TMP2)                  # exact
        p.emit("cbranch_s_z_b", S_EXC, "3")
        p.emit("saveexec_and", S_ARITH, S_EXC)                 # exec = exceptional lanes; S_ARITH = the arithmetic lanes
        RR = self.mont([(R, None)], E[0], E[0], sq_tmp=E[1])
        self.is_zero_mod_q(RR, S_TMP, S_TMP2)                  # lanes with r == 0: doubling
        p.emit("exec_andn2", S_EXC, S_TMP)                     # cancellation lanes: identity
        p.emit("cbranch_execz", "4")
        for i in range(13):
            p.emit("mov", ACC_ZZ[i], 0)
        p.emit("label", "4")
        p.emit("exec_and", S_EXC, S_TMP)
        p.emit("cbranch_execz", "5")
        self.double_acc(X, Y, ZZ, ZZZ)
        p.emit("label", "5")
        p.emit("exec_andn2", S_ARITH, S_EXC)                   # back to the lanes that are not exceptional
        p.emit("cbranch_execz_b", "9")
        p.emit("branch_b", "3")
        p.emit("label", "7")
        return p


# ---- big-integer model -------------------------------------------------------------------------------
def model_madd(acc, base):
    """acc = (X, Y, ZZ, ZZZ) residues (Montgomery form), base = (x, y) canonical Montgomery; returns the new residues
    or 'flag' when p == 0 (mod q); identity = ZZ residue... identity is decided by the caller (exact zero limbs)"""
    Ri = pow(R390, -1, Q)
    mm = lambda a, b: a * b * Ri % Q
    X, Y, ZZ, ZZZ = acc
    x, y = base
    u2 = mm(x, ZZ)
    s2 = mm(y, ZZZ)
    p_ = (u2 - X) % Q
    r = (s2 - Y) % Q
    if p_ == 0:
        return "flag"
    pp = mm(p_, p_)
    ppp = mm(p_, pp)
    qq = mm(X, pp)
    x3 = (mm(r, r) - ppp - 2 * qq) % Q
    y3 = (mm(r, (qq - x3) % Q) - mm(Y, ppp)) % Q
    return (x3, y3, mm(ZZ, pp), mm(ZZZ, ppp))


def model_dbl(acc):
    """dbl-2008-s-1 (a = 0) on Montgomery residues"""
    Ri = pow(R390, -1, Q)
    mm = lambda a, b: a * b * Ri % Q
    X, Y, ZZ, ZZZ = acc
    u = 2 * Y % Q
    v = mm(u, u)
    w = mm(u, v)
    s_ = mm(X, v)
    m = 3 * mm(X, X) % Q
    x3 = (mm(m, m) - 2 * s_) % Q
    y3 = (mm(m, (s_ - x3) % Q) - mm(w, Y)) % Q
    return (x3, y3, mm(v, ZZ), mm(w, ZZZ))


def selftest(ncases=400):
    g = Gen()
    prog = g.madd()
    rnd = random.Random(7)
    n_mad = sum(1 for i in prog.ins if i[0] == "mad64")
    print(f"madd30: {len(prog.ins)} instructions, {n_mad} v_mad_u64_u32; bounds leaving (units of q): {g.bounds}", file=sys.stderr)

    def limbs(v):
        return [(v >> (30 * i)) & MASK30 for i in range(12)] + [v >> 360]

    def run(acc_vals, bx, by, sgn=0):
        regs = {f"v{i}": rnd.getrandbits(32) for i in range(V_END + 4)}
        regs[SGN] = sgn
        regs.update({f"s{i}": rnd.getrandbits(1) for i in range(S_END + 2)})
        for regs_, v in zip((ACC_X, ACC_Y, ACC_ZZ, ACC_ZZZ), acc_vals):
            for r, l in zip(regs_, limbs(v)):
                regs[r] = l
        for i in range(12):
            regs[BXW[i]] = (bx >> (32 * i)) & M32
            regs[BYW[i]] = (by >> (32 * i)) & M32
        keep = {k: regs[k] for k in ACC_X + ACC_Y + ACC_ZZ + ACC_ZZZ}
        prog.run(regs)
        out = [sum(regs[r] << (30 * i) for i, r in enumerate(regs_)) for regs_ in (ACC_X, ACC_Y, ACC_ZZ, ACC_ZZZ)]
        for regs_ in (ACC_X, ACC_Y, ACC_ZZ, ACC_ZZZ):
            assert all(regs[r] <= MASK30 for r in regs_[:12]), "result limb not normalised"
        return regs[FLAG], out, all(regs[k] == keep[k] for k in keep)

    worst = [0, 0, 0, 0]
    nflag = 0
    for case in range(ncases):
        bx, by = rnd.randrange(Q), rnd.randrange(1, Q)
        kind = case % 8
        sgn = rnd.choice((0, 0, 1, 0x80000000))
        by_eff = Q - by if sgn else by
        if kind == 0:      # identity accumulator (X, Y, ZZZ arbitrary: only ZZ == 0 marks it)
            flag, out, _ = run((rnd.randrange(7 * Q), rnd.randrange(2 * Q), 0, rnd.randrange(2 * Q)), bx, by, sgn)
            assert flag == 0 and out == [bx, by_eff, R390 % Q, R390 % Q], "identity lane"
            continue
        # loose representatives at the invariant bounds: residue + j q
        res = [rnd.randrange(Q) for _ in range(4)]
        if res[2] == 0:
            res[2] = 1
        mult = (7, 2, 2, 2)
        if kind == 1:      # extreme representatives
            vals = [r + (m - 1) * Q for r, m in zip(res, mult)]
        else:
            vals = [r + rnd.randrange(m) * Q for r, m in zip(res, mult)]
        if kind in (2, 3):  # force p == 0: X = x2 * ZZ
            Ri = pow(R390, -1, Q)
            res[0] = bx * res[2] * Ri % Q
            vals[0] = res[0] + rnd.randrange(7) * Q
        if kind == 3:       # ... and r == 0: Y = (+-y2) * ZZZ  -> the accumulator IS the base: doubling
            res[1] = by_eff * res[3] * Ri % Q
            vals[1] = res[1] + rnd.randrange(2) * Q
        exp = model_madd(res, (bx, by_eff))
        flag, out, untouched = run(vals, bx, by, sgn)
        assert flag == 0
        if exp == "flag":
            nflag += 1
            if kind == 3:
                exp = model_dbl(res)
            else:
                assert out[2] == 0, "p == 0, r != 0: the sum is the identity (ZZ == 0 exactly)"
                continue
        for i in range(4):
            assert out[i] % Q == exp[i], ("coordinate", i, kind)
            worst[i] = max(worst[i], out[i] / Q)
    assert nflag > 0
    print(f"madd30: {ncases} cases ok ({nflag} exceptional); largest values leaving (units of q): {[round(w, 3) for w in worst]}", file=sys.stderr)


def emit(out):
    g = Gen()
    prog = g.madd()
    lines = prog.text()
    out.append("// GENERATED by gen_madd30.py -- do not edit; edit the generator.")
    out.append("// clang-format off")
    out.append(f"// XYZZ mixed addition on 13 x 30-bit loose limbs, one asm statement on physical registers: {len(prog.ins)} instructions,")
    out.append(f"// {sum(1 for i in prog.ins if i[0] == 'mad64')} v_mad_u64_u32.  acc = v[{VB}:{VB + 51}] (X, Y, ZZ, ZZZ); the base words arrive in v[{E0B + 1}:{E0B + 12}] (x) and")
    out.append(f"// v[{E1B + 1}:{E1B + 12}] (y) and are CONSUMED; every register of the statement is below v{V_END}.")
    out.append(f"constexpr int GM_MADD30_VGPRS = {V_END};")
    out.append("typedef uint32_t gm_u8v __attribute__((ext_vector_type(8)));")
    out.append("typedef uint32_t gm_u4v __attribute__((ext_vector_type(4)));")
    out.append("// acc: six 8-register groups + one 4-register group (52 limbs)")
    out.append("struct Acc30 { gm_u8v a0, a1, a2, a3, a4, a5; gm_u4v a6; };")
    out.append("// x0..x2 / y0..y2: the 12 + 12 canonical words of the affine base (device form, NOT the identity); neg != 0 adds -P.")
    out.append("// Complete (identity accumulator, doubling, cancellation); the return value is always 0.")
    out.append("__device__ __forceinline__ uint32_t g1_madd30_asm(Acc30& A, gm_u4v x0, gm_u4v x1, gm_u4v x2, gm_u4v y0, gm_u4v y1, gm_u4v y2, uint32_t neg) {")
    out.append("  uint32_t flag;")
    out.append("  asm volatile(")
    for line in lines:
        out.append(f'      "{line}\\n\\t"')
    outs = []
    ins = []
    for k in range(6):
        outs.append(f'"={{v[{VB + 8 * k}:{VB + 8 * k + 7}]}}"(A.a{k})')
        ins.append(f'"{{v[{VB + 8 * k}:{VB + 8 * k + 7}]}}"(A.a{k})')
    outs.append(f'"={{v[{VB + 48}:{VB + 51}]}}"(A.a6)')
    ins.append(f'"{{v[{VB + 48}:{VB + 51}]}}"(A.a6)')
    outs.append(f'"={{{FLAG}}}"(flag)')
    for k, name in enumerate(("x0", "x1", "x2")):
        outs.append(f'"={{v[{E0B + 1 + 4 * k}:{E0B + 4 + 4 * k}]}}"({name})')
        ins.append(f'"{{v[{E0B + 1 + 4 * k}:{E0B + 4 + 4 * k}]}}"({name})')
    for k, name in enumerate(("y0", "y1", "y2")):
        outs.append(f'"={{v[{E1B + 1 + 4 * k}:{E1B + 4 + 4 * k}]}}"({name})')
        ins.append(f'"{{v[{E1B + 1 + 4 * k}:{E1B + 4 + 4 * k}]}}"({name})')
    ins.append(f'"{{{SGN}}}"(neg)')
    out.append("      : " + ", ".join(outs))
    out.append("      : " + ", ".join(ins))
    pinned = set(range(VB, VB + 52)) | set(range(E0B + 1, E0B + 13)) | set(range(E1B + 1, E1B + 13)) | {int(FLAG[1:]), int(SGN[1:])}
    clob = ['"vcc"', '"scc"'] + [f'"s{i}"' for i in range(SB, S_END)] + [f'"v{i}"' for i in range(VB, V_END) if i not in pinned]
    out.append("      : " + ", ".join(clob) + ");")
    out.append("  return flag;")
    out.append("}")
    out.append("// clang-format on")


# ---- register plan of the full addition acc += o (both XYZZ, loose) ------------------------------------
AQ = VB + 54              # o = (X2, Y2, ZZ2, ZZZ2): v[AQ .. AQ+51], in/out (consumed)
assert AQ % 2 == 0
O_X = [f"v{AQ + i}" for i in range(13)]
O_Y = [f"v{AQ + 13 + i}" for i in range(13)]
O_ZZ = [f"v{AQ + 26 + i}" for i in range(13)]
O_ZZZ = [f"v{AQ + 39 + i}" for i in range(13)]
A_TB = AQ + 52
AE = [[f"v{A_TB + 13 * e + i}" for i in range(13)] for e in range(4)]
A_ACCUM = A_TB + 52
A_SPL, A_SPT = A_ACCUM + 2, A_ACCUM + 4
A_TMP, A_CAR, A_FLAG = f"v{A_ACCUM + 6}", f"v{VB + 52}", f"v{A_ACCUM + 7}"
A_V_END = A_ACCUM + 8
assert A_ACCUM % 2 == 0
S_OID = SB + 20           # s[60:61]: lanes whose o is the identity


def gen_add():
    """acc += o for two loose XYZZ points (EFD add-2008-s, 11 products + 2 squares + one fused double product, 13 reductions).
    o == identity: acc stays; acc == identity: acc = o; p == 0 (mod q): exact test, then acc = 2 acc when r == 0 as well and
    acc = identity otherwise, in a cold block inside the statement (as in the mixed addition).  Complete; flag is always 0."""
    g = Gen(Plan(A_ACCUM, A_SPL, A_SPT, A_TMP, A_CAR))
    p = g.p
    for j in range(13):
        p.emit("smov", SP[j], P30[j])
    p.emit("smov", SINV, INV30)
    p.emit("mov", A_FLAG, 0)
    BX, BY, BZ = 7 * Q, 2 * Q, 2 * Q
    X1, Y1, ZZ1, ZZZ1 = V(ACC_X, BX), V(ACC_Y, BY), V(ACC_ZZ, BZ), V(ACC_ZZZ, BZ)
    X2, Y2, ZZ2, ZZZ2 = V(O_X, BX), V(O_Y, BY), V(O_ZZ, BZ), V(O_ZZZ, BZ)
    # identity masks
    for regs, sdst in ((ACC_ZZ, S_IDENT), (O_ZZ, S_OID)):
        p.emit("or3", A_TMP, regs[0], regs[1], regs[2])
        for i in range(3, 13, 2):
            p.emit("or3", A_TMP, A_TMP, regs[i], regs[i + 1])
        p.emit("cmp_eq_s", sdst, 0, A_TMP)
    p.emit("s_nop", 4)
    # acc identity (whatever o is): acc = o
    p.emit("saveexec_and", S_SAVE, S_IDENT)
    p.emit("cbranch_execz", "1")
    for a_, o_ in ((ACC_X, O_X), (ACC_Y, O_Y), (ACC_ZZ, O_ZZ), (ACC_ZZZ, O_ZZZ)):
        for i in range(13):
            p.emit("mov", a_[i], o_[i])
    p.emit("label", "1")
    # arithmetic lanes: neither is the identity
    p.emit("s_or", S_TMP, S_IDENT, S_OID)
    p.emit("exec_andn2", S_SAVE, S_TMP)
    p.emit("cbranch_execz", "9")
    U1 = g.mont([(X1, ZZ2)], AE[0], AE[0])
    U2 = g.mont([(X2, ZZ1)], AE[1], AE[1])
    P = g.sub(U2, U1, 2)                                       # E1
    PP = g.mont([(P, None)], AE[2], AE[2], sq_tmp=AE[3])       # E2
    assert PP.bound < 2 * Q
    p.emit("cmp_eq_s", S_TMP, SP[0], PP.r[0])
    p.emit("cmp_eq_s", S_TMP2, 0, PP.r[0])
    p.emit("s_nop", 4)
    p.emit("s_or", S_TMP, S_TMP, S_TMP2)
    p.emit("cbranch_s_nz", S_TMP, "6")                         # cold block behind the end of the statement
    p.emit("label", "3")
    S1 = g.mont([(Y1, ZZZ2)], AE[3], AE[3])                    # E3
    S2 = g.mont([(Y2, ZZZ1)], ACC_Y, ACC_Y)                    # Y1 dead
    R = g.sub(S2, S1, 2)                                       # ACC_Y
    ZZ12 = g.mont([(ZZ1, ZZ2)], O_X, ACC_ZZ)                   # in place; X2 dead: O_X is scratch
    ZZn = g.mont([(ZZ12, PP)], O_X, ACC_ZZ)
    PPP = g.mont([(P, PP)], O_X, O_X)                          # ppp in O_X; P dead
    ZZZ12 = g.mont([(ZZZ1, ZZZ2)], AE[1], ACC_ZZZ)
    ZZZn = g.mont([(ZZZ12, PPP)], AE[1], ACC_ZZZ)
    QQ = g.mont([(U1, PP)], AE[1], AE[1])                      # q = u1 pp (E1); U1, pp dead
    R2 = g.mont([(R, None)], ACC_X, ACC_X, sq_tmp=AE[0])       # X1 dead since U1
    X3a = g.sub(R2, PPP, 2)
    D = g.dbl(QQ, AE[0])
    X3 = g.sub(X3a, D, 3)
    T = g.sub(QQ, X3, 7)                                       # E1
    NS = g.rsub(S1, 2)                                         # E3
    Y3 = g.mont([(R, T), (PPP, NS)], AE[2], ACC_Y)             # in place on R
    assert X3.bound <= BX and Y3.bound <= BY and ZZn.bound <= BZ and ZZZn.bound <= BZ
    g.bounds = {"X": X3.bound / Q, "Y": Y3.bound / Q, "ZZ": ZZn.bound / Q, "ZZZ": ZZZn.bound / Q, "P": P.bound / Q, "T": T.bound / Q}
    p.emit("label", "9")
    p.emit("restore_exec", S_SAVE)
    p.emit("branch", "7")
    # ---- cold: u1 == u2 (mod q) on some lane: the same x.  s1 == s2 as well -> acc = 2 acc, else acc = identity.
    # (Equal partial sums are the rule when every base is the same point -- the elastic benchmark's key.)
    p.emit("label", "6")
    g.is_zero_mod_q(PP, S_EXC, S_TMP2)
    p.emit("cbranch_s_z_b", S_EXC, "3")
    p.emit("saveexec_and", S_ARITH, S_EXC)
    S1x = g.mont([(Y1, ZZZ2)], AE[2], AE[2])                   # pp is dead on these lanes
    S2x = g.mont([(Y2, ZZZ1)], AE[3], AE[3])
    Rx = g.sub(S2x, S1x, 2)
    RRx = g.mont([(Rx, None)], AE[0], AE[0], sq_tmp=AE[1])
    g.is_zero_mod_q(RRx, S_TMP, S_TMP2)
    p.emit("exec_andn2", S_EXC, S_TMP)
    p.emit("cbranch_execz", "4")
    for i in range(13):
        p.emit("mov", ACC_ZZ[i], 0)
    p.emit("label", "4")
    p.emit("exec_and", S_EXC, S_TMP)
    p.emit("cbranch_execz", "5")
    g.double_acc(X1, Y1, ZZ1, ZZZ1, E=AE)
    p.emit("label", "5")
    p.emit("exec_andn2", S_ARITH, S_EXC)
    p.emit("cbranch_execz_b", "9")
    p.emit("branch_b", "3")
    p.emit("label", "7")
    return g


def model_add(a, o):
    Ri = pow(R390, -1, Q)
    mm = lambda x, y: x * y * Ri % Q
    X1, Y1, ZZ1, ZZZ1 = a
    X2, Y2, ZZ2, ZZZ2 = o
    u1, u2, s1, s2 = mm(X1, ZZ2), mm(X2, ZZ1), mm(Y1, ZZZ2), mm(Y2, ZZZ1)
    p_, r = (u2 - u1) % Q, (s2 - s1) % Q
    if p_ == 0:
        return "flag"
    pp = mm(p_, p_)
    ppp = mm(p_, pp)
    qq = mm(u1, pp)
    x3 = (mm(r, r) - ppp - 2 * qq) % Q
    y3 = (mm(r, (qq - x3) % Q) - mm(s1, ppp)) % Q
    return (x3, y3, mm(mm(ZZ1, ZZ2), pp), mm(mm(ZZZ1, ZZZ2), ppp))


def selftest_add(ncases=300):
    g = gen_add()
    prog = g.p
    rnd = random.Random(11)
    n_mad = sum(1 for i in prog.ins if i[0] == "mad64")
    print(f"add30: {len(prog.ins)} instructions, {n_mad} v_mad_u64_u32; bounds leaving (units of q): {g.bounds}", file=sys.stderr)
    groups = (ACC_X, ACC_Y, ACC_ZZ, ACC_ZZZ, O_X, O_Y, O_ZZ, O_ZZZ)

    def limbs(v):
        return [(v >> (30 * i)) & MASK30 for i in range(12)] + [v >> 360]

    def run(vals):
        regs = {f"v{i}": rnd.getrandbits(32) for i in range(A_V_END + 4)}
        regs.update({f"s{i}": rnd.getrandbits(1) for i in range(S_END + 2)})
        for regs_, v in zip(groups, vals):
            for r, l in zip(regs_, limbs(v)):
                regs[r] = l
        keep = {k: regs[k] for grp in groups for k in grp}
        prog.run(regs)
        out = [sum(regs[r] << (30 * i) for i, r in enumerate(regs_)) for regs_ in groups[:4]]
        for regs_ in groups[:4]:
            assert all(regs[r] <= MASK30 for r in regs_[:12]), "result limb not normalised"
        return regs[A_FLAG], out, all(regs[k] == keep[k] for k in keep), all(regs[k] == keep[k] for grp in groups[:4] for k in grp)

    mult = (7, 2, 2, 2)
    nflag = 0
    for case in range(ncases):
        kind = case % 8
        res = [rnd.randrange(Q) for _ in range(8)]
        for i in (2, 3, 6, 7):
            res[i] = res[i] or 1
        if kind == 1:
            vals = [r + (m - 1) * Q for r, m in zip(res, mult + mult)]
        else:
            vals = [r + rnd.randrange(m) * Q for r, m in zip(res, mult + mult)]
        if kind == 2:      # acc identity: result = o exactly (limbs copied)
            vals[2] = 0
            flag, out, _, _ = run(vals)
            assert flag == 0 and out == vals[4:], "acc identity"
            continue
        if kind == 3:      # o identity: acc unchanged
            vals[6] = 0
            flag, out, _, acc_same = run(vals)
            assert flag == 0 and acc_same, "o identity"
            continue
        if kind == 4:      # both identity
            vals[2] = vals[6] = 0
            flag, out, _, _ = run(vals)
            assert flag == 0 and out[2] == 0, "both identity"
            continue
        if kind == 5:      # same x: u1 == u2  ->  X2 = X1 ZZ2 / ZZ1
            res[4] = res[0] * res[6] * pow(res[2], -1, Q) % Q
            vals[4] = res[4] + rnd.randrange(7) * Q
        if kind == 6:      # the same point: x and y agree  ->  doubling
            res[4] = res[0] * res[6] * pow(res[2], -1, Q) % Q
            res[5] = res[1] * res[7] * pow(res[3], -1, Q) % Q
            vals[4] = res[4] + rnd.randrange(7) * Q
            vals[5] = res[5] + rnd.randrange(2) * Q
        exp = model_add(res[:4], res[4:])
        flag, out, untouched, _ = run(vals)
        assert flag == 0
        if exp == "flag":
            nflag += 1
            if kind == 6:
                exp = model_dbl(res[:4])
            else:
                assert out[2] == 0, "same x, different y: the sum is the identity"
                continue
        for i in range(4):
            assert out[i] % Q == exp[i], ("coordinate", i, kind)
    assert nflag > 0
    print(f"add30: {ncases} cases ok ({nflag} exceptional)", file=sys.stderr)


def emit_add(out):
    g = gen_add()
    prog = g.p
    out.append(f"// XYZZ + XYZZ on loose limbs (add-2008-s): {len(prog.ins)} instructions, {sum(1 for i in prog.ins if i[0] == 'mad64')} v_mad_u64_u32.")
    out.append(f"// acc = v[{VB}:{VB + 51}], o = v[{AQ}:{AQ + 51}] (CONSUMED); registers below v{A_V_END}.  Complete: identity operands, doubling, cancellation.")
    out.append(f"constexpr int GM_ADD30_VGPRS = {A_V_END};")
    out.append("// the return value is always 0 (kept for the wrapper's shape)")
    out.append("__device__ __forceinline__ uint32_t g1_add30_asm(Acc30& A, Acc30& O) {")
    out.append("  uint32_t flag;")
    out.append("  asm volatile(")
    for line in prog.text():
        out.append(f'      "{line}\\n\\t"')
    outs, ins = [], []
    for base, name in ((VB, "A"), (AQ, "O")):
        for k in range(6):
            outs.append(f'"={{v[{base + 8 * k}:{base + 8 * k + 7}]}}"({name}.a{k})')
            ins.append(f'"{{v[{base + 8 * k}:{base + 8 * k + 7}]}}"({name}.a{k})')
        outs.append(f'"={{v[{base + 48}:{base + 51}]}}"({name}.a6)')
        ins.append(f'"{{v[{base + 48}:{base + 51}]}}"({name}.a6)')
    outs.append(f'"={{{A_FLAG}}}"(flag)')
    out.append("      : " + ", ".join(outs))
    out.append("      : " + ", ".join(ins))
    pinned = set(range(VB, VB + 52)) | set(range(AQ, AQ + 52)) | {int(A_FLAG[1:])}
    clob = ['"vcc"', '"scc"'] + [f'"s{i}"' for i in range(SB, S_END)] + [f'"v{i}"' for i in range(VB, A_V_END) if i not in pinned]
    out.append("      : " + ", ".join(clob) + ");")
    out.append("  return flag;")
    out.append("}")


def render() -> str:
    """the whole g1_madd30_gen.inc"""
    out = []
    emit(out)
    assert out[-1] == "// clang-format on"
    out.pop()
    emit_add(out)
    out.append("// clang-format on")
    return "\n".join(out) + "\n"


def main():
    if "--selftest" in sys.argv:
        selftest(2000 if "--long" in sys.argv else 400)
        selftest_add(1500 if "--long" in sys.argv else 300)
        return
    sys.stdout.write(render())


if __name__ == "__main__":
    main()

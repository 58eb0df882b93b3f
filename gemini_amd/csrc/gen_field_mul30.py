#!/usr/bin/env python3
"""Emits field_mul30_gen.inc: the production Fq multiplier / squarer for gfx950.

Elements are canonical 12 x 32-bit records (field.cuh) holding a * 2^390 mod q.  The PRODUCT runs in
radix 2^30: both operands are unpacked to 13 x 30-bit limbs, so a partial product (< 2^60) is ONE
v_mad_u64_u32 into a 64-bit column accumulator -- no carry instruction (v_addc_co_u32 is half rate on
gfx950 like the mad itself, tools/ubench_isa.hip; the 32-bit-limb multiplier of gen_field_mul.py pays
both per partial product: 600 half-rate instructions against 338 here).  Product scanning with the
Montgomery reduction interleaved (R' = 2^390 = 2^(30*13)); per column one v_mul_lo_u32 for m_k, two
full-rate instructions for the 30-bit shift.  The three columns whose worst-case sum exceeds 2^64
(k = 10, 11, 12; bounds are tracked below for canonical inputs and the real limbs of q) split their
accumulator once between the a.b and the m.q halves.  Result: 13 limbs < 1.002 q, repacked to 12 words
and conditionally reduced.

Each function is ONE asm statement on physical registers.  The operands arrive in v0..v23 by the
calling convention of a __noinline__ device function and the result leaves in v0..v11; every
temporary is a caller-saved register (v24-v39, v48-v55, v64-v71, s4-s17), so the function has no
prologue, no epilogue and no compiler-inserted moves or s_nops.  hipcc's per-statement padding and its
64-bit shift / move sequences were what held the compiler-scheduled radix-2^30 product
(field30.cuh::fq30_mul) at 61.6 Gmul/s.

The generator also INTERPRETS the instruction list it emits (integer semantics of every opcode used)
against big-integer Montgomery arithmetic -- `python3 gen_field_mul30.py --selftest` -- so index and
bound mistakes show up without a GPU; tools/fqmul_check.hip compares the assembled code with the
32-bit-limb multiplier on the device.

Run:  python3 gen_field_mul30.py > field_mul30_gen.inc
"""
import random
import sys

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
MASK30 = (1 << 30) - 1
P30 = [(Q >> (30 * i)) & MASK30 for i in range(13)]
Q32 = [(Q >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
INV30 = (-pow(Q, -1, 1 << 30)) % (1 << 30)
M32 = 0xFFFFFFFF

# ---- register plan (all caller-saved) ------------------------------------------------------------
A_H = [f"v{i}" for i in range(24, 37)]
B_H = ["v37", "v38", "v39", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67"]
ACC = 48      # v[48:49]
SPL_H = 68    # v[68:69]: the a.b half of a split column, shifted
TMP = "v70"
MT = [f"v{i}" for i in range(12)] + ["v71"]  # m_k, later t_k (t_k replaces m_k once column k + 12 is done); v12..v23 (operand b) stay intact:
# an asm statement must not modify an input-only operand -- the caller (IPRA) assumes b survives the call
SP = [f"s{4 + i}" for i in range(13)]
SINV = "s17"
QV = [f"v{i}" for i in range(24, 36)]                                    # q (32-bit limbs) for the final subtraction
DV = ["v36", "v37", "v38", "v39", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65"]
CLOBBER_V = list(range(24, 40)) + list(range(48, 56)) + list(range(64, 72))
CLOBBER_S = list(range(4, 18))


class Prog:
    def __init__(self):
        self.ins = []

    def emit(self, op, *args):
        self.ins.append((op,) + args)

    # -- text ------------------------------------------------------------------------------------
    @staticmethod
    def _o(x):
        if isinstance(x, int):
            return str(x) if -16 <= x <= 64 else hex(x)
        return x

    def text(self):
        out = []
        o = self._o
        for ins in self.ins:
            op = ins[0]
            if op == "mad64":
                d, x, y = ins[1:]
                out.append(f"v_mad_u64_u32 v[{d}:{d + 1}], vcc, {o(x)}, {o(y)}, v[{d}:{d + 1}]")
            elif op in ("mul_lo", "and", "lshr", "lshl", "add", "sub"):
                name = {"mul_lo": "v_mul_lo_u32", "and": "v_and_b32", "lshr": "v_lshrrev_b32", "lshl": "v_lshlrev_b32",
                        "add": "v_add_u32", "sub": "v_sub_u32"}[op]
                out.append(f"{name} {ins[1]}, {o(ins[2])}, {o(ins[3])}")
            elif op == "alignbit":
                out.append(f"v_alignbit_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}, {o(ins[4])}")
            elif op == "lshl_or":
                out.append(f"v_lshl_or_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}, {o(ins[4])}")
            elif op == "lshr64":
                d, sh, a = ins[1:]
                out.append(f"v_lshrrev_b64 v[{d}:{d + 1}], {o(sh)}, v[{a}:{a + 1}]")
            elif op == "mov":
                out.append(f"v_mov_b32 {ins[1]}, {o(ins[2])}")
            elif op == "smov":
                out.append(f"s_mov_b32 {ins[1]}, {o(ins[2])}")
            elif op in ("add_co", "sub_co"):
                out.append(f"v_{op}_u32 {ins[1]}, vcc, {o(ins[2])}, {o(ins[3])}")
            elif op in ("addc_co", "subb_co"):
                out.append(f"v_{op}_u32 {ins[1]}, vcc, {o(ins[2])}, {o(ins[3])}, vcc")
            elif op == "cndmask":
                out.append(f"v_cndmask_b32 {ins[1]}, {o(ins[2])}, {o(ins[3])}, vcc")
            elif op == "cmp_le":
                out.append(f"v_cmp_le_u32 vcc, {o(ins[1])}, {o(ins[2])}")
            elif op == "branch_vccz":
                out.append("s_nop 4")
                out.append(f"s_cbranch_vccz {ins[1]}")
            elif op == "label":
                out.append(f"{ins[1]}:")
            else:
                raise ValueError(op)
        return out

    # -- interpreter (one lane) --------------------------------------------------------------------
    def run(self, regs):
        def g(x):
            return x & M32 if isinstance(x, int) else regs[x]

        vcc = 0
        pc = 0
        labels = {ins[1]: i for i, ins in enumerate(self.ins) if ins[0] == "label"}
        while pc < len(self.ins):
            ins = self.ins[pc]
            pc += 1
            op = ins[0]
            if op == "mad64":
                d, x, y = ins[1:]
                acc = regs[f"v{d}"] | (regs[f"v{d + 1}"] << 32)
                acc += g(x) * g(y)
                assert acc < (1 << 64), "64-bit column accumulator overflow"
                regs[f"v{d}"] = acc & M32
                regs[f"v{d + 1}"] = acc >> 32
            elif op == "mul_lo":
                regs[ins[1]] = (g(ins[2]) * g(ins[3])) & M32
            elif op == "and":
                regs[ins[1]] = g(ins[2]) & g(ins[3])
            elif op == "lshr":
                regs[ins[1]] = g(ins[3]) >> (g(ins[2]) & 31)
            elif op == "lshl":
                regs[ins[1]] = (g(ins[3]) << (g(ins[2]) & 31)) & M32
            elif op == "add":
                regs[ins[1]] = (g(ins[2]) + g(ins[3])) & M32
            elif op == "sub":
                regs[ins[1]] = (g(ins[2]) - g(ins[3])) & M32
            elif op == "alignbit":
                regs[ins[1]] = (((g(ins[2]) << 32) | g(ins[3])) >> (g(ins[4]) & 31)) & M32
            elif op == "lshl_or":
                regs[ins[1]] = ((g(ins[2]) << (g(ins[3]) & 31)) | g(ins[4])) & M32
            elif op == "lshr64":
                d, sh, a = ins[1:]
                v = (regs[f"v{a}"] | (regs[f"v{a + 1}"] << 32)) >> (g(sh) & 63)
                regs[f"v{d}"], regs[f"v{d + 1}"] = v & M32, v >> 32
            elif op in ("mov", "smov"):
                regs[ins[1]] = g(ins[2])
            elif op == "add_co":
                s = g(ins[2]) + g(ins[3])
                regs[ins[1]], vcc = s & M32, s >> 32
            elif op == "addc_co":
                s = g(ins[2]) + g(ins[3]) + vcc
                regs[ins[1]], vcc = s & M32, s >> 32
            elif op == "sub_co":
                s = g(ins[2]) - g(ins[3])
                regs[ins[1]], vcc = s & M32, int(s < 0)
            elif op == "subb_co":
                s = g(ins[2]) - g(ins[3]) - vcc
                regs[ins[1]], vcc = s & M32, int(s < 0)
            elif op == "cndmask":
                regs[ins[1]] = g(ins[3]) if vcc else g(ins[2])
            elif op == "cmp_le":
                vcc = int(g(ins[1]) <= g(ins[2]))
            elif op == "branch_vccz":
                if not vcc:
                    pc = labels[ins[1]]
            elif op == "label":
                pass
            else:
                raise ValueError(op)
        return regs


def unpack(p, src, dst):
    """12 x 32-bit words -> 13 x 30-bit limbs (24 instructions)"""
    p.emit("and", dst[0], MASK30, src[0])
    for i in range(1, 12):
        p.emit("alignbit", dst[i], src[i], src[i - 1], 32 - 2 * i)
        p.emit("and", dst[i], MASK30, dst[i])
    p.emit("lshr", dst[12], 8, src[11])


def gen(square, loose=False):
    """loose = False: packed canonical operands in v0..v11 / v12..v23, packed canonical result in v0..v11.
    loose = True: 13 normalised 30-bit limbs per operand in v0..v12 / v13..v25 (value < 2^386, limbs 0..11 < 2^30),
    result limbs in v0..v12 (normalised, value < q + ab / 2^390): no unpack, no repack, no conditional subtraction."""
    p = Prog()
    for j in range(13):
        p.emit("smov", SP[j], P30[j])
    p.emit("smov", SINV, INV30)
    if loose:
        A = [f"v{i}" for i in range(13)]
        B = [f"v{i}" for i in range(13, 26)]
        M = [f"v{i}" for i in range(26, 39)]  # m_k
        T = A                                  # t_j replaces A_j (last read in column j + 12)
        SPL = 50
        if square:
            for j in range(1, 13):
                p.emit("lshl", B[j], 1, A[j])
        LB = [MASK30] * 12 + [(1 << 26) - 1]
    else:
        A, B, M, T, SPL = A_H, B_H, MT, MT, SPL_H
        a_w = [f"v{i}" for i in range(12)]
        b_w = [f"v{i}" for i in range(12, 24)]
        unpack(p, a_w, A)
        if square:
            # D_j = 2 A_j (31 bits): off-diagonal products are taken once against the doubled limb
            for j in range(1, 13):
                p.emit("lshl", B[j], 1, A[j])
        else:
            unpack(p, b_w, B)
        # worst-case bound of the accumulator for canonical inputs (limb 12 of a value < q is < 2^21)
        LB = [MASK30] * 12 + [(Q >> 360)]
    p.emit("mov", f"v{ACC}", 0)
    p.emit("mov", f"v{ACC + 1}", 0)
    bound = 0

    def mad(x, y, bx, by):
        nonlocal bound
        bound += bx * by
        assert bound < (1 << 64), "worst-case accumulator bound exceeds 2^64"
        p.emit("mad64", ACC, x, y)

    for k in range(25):
        lo_i, hi_i = max(0, k - 12), min(k, 12)
        n_ab = hi_i - lo_i + 1
        red = [(i, k - i) for i in (range(0, k) if k < 13 else range(k - 12, 13))]
        worst = bound + sum(LB[i] * LB[k - i] for i in range(lo_i, hi_i + 1)) + sum(MASK30 * P30[j] for _, j in red) + (MASK30 * P30[0] if k < 13 else 0)
        split = worst >= (1 << 64)
        # a.b half
        if square:
            for i in range(lo_i, hi_i + 1):
                j = k - i
                if i < j:
                    mad(A[i], B[j], LB[i], 2 * LB[j])
                elif i == j:
                    mad(A[i], A[i], LB[i], LB[i])
        else:
            for i in range(lo_i, hi_i + 1):
                mad(A[i], B[k - i], LB[i], LB[k - i])
        if split:
            p.emit("lshr64", SPL, 30, ACC)
            p.emit("and", f"v{ACC}", MASK30, f"v{ACC}")
            p.emit("mov", f"v{ACC + 1}", 0)
            spl_bound = bound >> 30
            bound = MASK30
        # m.q half
        for i, j in red:
            mad(M[i], SP[j], MASK30, P30[j])
        if k < 13:
            p.emit("mul_lo", M[k], f"v{ACC}", SINV)
            p.emit("and", M[k], MASK30, M[k])
            mad(M[k], SP[0], MASK30, P30[0])
        else:
            p.emit("and", T[k - 13], MASK30, f"v{ACC}")
        p.emit("lshr64", ACC, 30, ACC)  # one half-rate instruction; v_alignbit_b32 (half rate too) + v_lshrrev_b32 cost 1.5
        bound >>= 30
        if split:
            p.emit("add_co", f"v{ACC}", f"v{ACC}", f"v{SPL}")
            p.emit("addc_co", f"v{ACC + 1}", f"v{ACC + 1}", f"v{SPL + 1}")
            bound += spl_bound
    p.emit("mov", T[12], f"v{ACC}")
    if loose:
        return p
    # 13 x 30 -> 12 x 32, in place (word w needs limbs w and w + 1 only)
    p.emit("lshl_or", "v0", MT[1], 30, MT[0])
    for w in range(1, 12):
        p.emit("lshr", TMP, 2 * w, MT[w])
        p.emit("lshl_or", f"v{w}", MT[w + 1], 30 - 2 * w, TMP)
    # canonical form: the product is < q (1 + q / 2^390) < 1.002 q, so r >= q needs r_11 >= q_11; the
    # subtraction is skipped by the waves where no lane can need it (~7 of 8)
    p.emit("cmp_le", Q32[11], "v11")
    p.emit("branch_vccz", "1f")
    for i in range(12):
        p.emit("mov", QV[i], Q32[i])
    p.emit("sub_co", DV[0], "v0", QV[0])
    for i in range(1, 12):
        p.emit("subb_co", DV[i], f"v{i}", QV[i])
    for i in range(12):
        p.emit("cndmask", f"v{i}", DV[i], f"v{i}")  # borrow -> keep r
    p.emit("label", "1")
    return p


def selftest():
    rnd = random.Random(1)
    Rinv = pow(1 << 390, -1, Q)
    for square in (False, True):
        p = gen(square)
        # label/branch names for the interpreter
        for i, ins in enumerate(p.ins):
            if ins[0] == "branch_vccz":
                p.ins[i] = ("branch_vccz", "1")
        cases = [(0, 0), (1, 1), (Q - 1, Q - 1), (Q - 1, 1), ((1 << 380) - 1, (1 << 380) - 1)]
        # values whose 30-bit limbs are all ones up to the size of q, products landing just above q, random
        cases += [(Q - 1 - rnd.getrandbits(200), Q - 1 - rnd.getrandbits(200)) for _ in range(200)]
        cases += [(rnd.randrange(Q), rnd.randrange(Q)) for _ in range(3000)]
        # force results in [q, 1.002 q) before the conditional subtraction: pick a, solve for b
        for _ in range(300):
            a = rnd.randrange(1, Q)
            target = rnd.randrange(0, Q >> 9)  # the canonical result; raw result may be target + q
            b = target * pow(a, -1, Q) * (1 << 390) % Q
            cases.append((a, b))
        nsub = 0
        for a, b in cases:
            if square:
                b = a
            regs = {f"v{i}": rnd.getrandbits(32) for i in range(72)}
            regs.update({f"s{i}": rnd.getrandbits(32) for i in range(32)})
            for i in range(12):
                regs[f"v{i}"] = (a >> (32 * i)) & M32
                if not square:
                    regs[f"v{12 + i}"] = (b >> (32 * i)) & M32
            p.run(regs)
            got = sum(regs[f"v{i}"] << (32 * i) for i in range(12))
            exp = a * b * Rinv % Q
            assert got == exp, (square, hex(a), hex(b), hex(got), hex(exp))
        n_mad = sum(1 for i in p.ins if i[0] == "mad64")
        print(f"{'sqr' if square else 'mul'}: {len(cases)} cases ok; {len(p.ins)} instructions, {n_mad} v_mad_u64_u32", file=sys.stderr)


def selftest_loose():
    rnd = random.Random(2)
    Rinv = pow(1 << 390, -1, Q)
    for square in (False, True):
        p = gen(square, loose=True)
        cases = [(0, 0), ((1 << 386) - 1, (1 << 386) - 1), (Q - 1, Q - 1), (10 * Q, 10 * Q - 1)]
        cases += [((1 << 386) - 1 - rnd.getrandbits(300), (1 << 386) - 1 - rnd.getrandbits(300)) for _ in range(200)]
        cases += [(rnd.getrandbits(386), rnd.getrandbits(386)) for _ in range(3000)]
        worst = 0
        for a, b in cases:
            if square:
                b = a
            regs = {f"v{i}": rnd.getrandbits(32) for i in range(72)}
            regs.update({f"s{i}": rnd.getrandbits(32) for i in range(32)})
            for i in range(13):
                regs[f"v{i}"] = (a >> (30 * i)) & (MASK30 if i < 12 else M32)
                if not square:
                    regs[f"v{13 + i}"] = (b >> (30 * i)) & (MASK30 if i < 12 else M32)
            p.run(regs)
            assert all(regs[f"v{i}"] <= MASK30 for i in range(12)), "result limb not normalised"
            got = sum(regs[f"v{i}"] << (30 * i) for i in range(13))
            assert got == (a * b + (-(a * b) * pow(Q, -1, 1 << 390) % (1 << 390)) * Q) >> 390, "not the Montgomery quotient"
            assert got % Q == a * b * Rinv % Q and got < Q + (a * b >> 390) + 1
            worst = max(worst, got)
        n_mad = sum(1 for i in p.ins if i[0] == "mad64")
        print(f"loose {'sqr' if square else 'mul'}: {len(cases)} cases ok; {len(p.ins)} instructions, {n_mad} v_mad_u64_u32, max result {worst / Q:.3f} q", file=sys.stderr)


def emit_fn_loose(name, p, square, out):
    args = ", ".join(f"uint32_t a{i}" for i in range(13))
    if not square:
        args += ", " + ", ".join(f"uint32_t b{i}" for i in range(13))
    out.append(f"__device__ __noinline__ Fq30 {name}({args}) {{")
    out.append("  typedef uint32_t gm_u4 __attribute__((ext_vector_type(4)));")
    out.append("  gm_u4 A0 = {a0, a1, a2, a3}, A1 = {a4, a5, a6, a7}, A2 = {a8, a9, a10, a11};")
    if not square:
        out.append("  gm_u4 B0 = {b1, b2, b3, b4}, B1 = {b5, b6, b7, b8}, B2 = {b9, b10, b11, b12};")
    out.append("  gm_u4 R0, R1, R2;")
    out.append("  uint32_t R3;")
    out.append("  asm volatile(")
    for line in p.text():
        out.append(f'      "{line}\\n\\t"')
    out.append('      : "={v[0:3]}"(R0), "={v[4:7]}"(R1), "={v[8:11]}"(R2), "={v12}"(R3)')
    ins = '"{v[0:3]}"(A0), "{v[4:7]}"(A1), "{v[8:11]}"(A2), "{v12}"(a12)'
    if not square:
        ins += ', "{v13}"(b0), "{v[14:17]}"(B0), "{v[18:21]}"(B1), "{v[22:25]}"(B2)'
    out.append(f"      : {ins}")
    cv = (list(range(13, 26)) if square else []) + list(range(26, 40)) + list(range(48, 52))
    clob = ['"vcc"'] + [f'"s{i}"' for i in CLOBBER_S] + [f'"v{i}"' for i in cv]
    out.append(f"      : {', '.join(clob)});")
    out.append("  Fq30 r;")
    for i in range(12):
        out.append(f"  r.l[{i}] = R{i // 4}.{'xyzw'[i % 4]};")
    out.append("  r.l[12] = R3;")
    out.append("  return r;")
    out.append("}")


def emit_fn(name, p, nargs, out):
    args = ", ".join(f"uint32_t a{i}" for i in range(12))
    if nargs == 24:
        args += ", " + ", ".join(f"uint32_t b{i}" for i in range(12))
    out.append(f"__device__ __noinline__ Fq {name}({args}) {{")
    out.append("  typedef uint32_t gm_u4 __attribute__((ext_vector_type(4)));")
    out.append("  gm_u4 A0 = {a0, a1, a2, a3}, A1 = {a4, a5, a6, a7}, A2 = {a8, a9, a10, a11};")
    if nargs == 24:
        out.append("  gm_u4 B0 = {b0, b1, b2, b3}, B1 = {b4, b5, b6, b7}, B2 = {b8, b9, b10, b11};")
    out.append("  gm_u4 R0, R1, R2;")
    out.append("  asm volatile(")
    for line in p.text():
        out.append(f'      "{line}\\n\\t"')
    out.append('      : "={v[0:3]}"(R0), "={v[4:7]}"(R1), "={v[8:11]}"(R2)')
    ins = '"{v[0:3]}"(A0), "{v[4:7]}"(A1), "{v[8:11]}"(A2)'
    if nargs == 24:
        ins += ', "{v[12:15]}"(B0), "{v[16:19]}"(B1), "{v[20:23]}"(B2)'
    out.append(f"      : {ins}")
    clob = ['"vcc"'] + [f'"s{i}"' for i in CLOBBER_S] + [f'"v{i}"' for i in ([] if nargs == 24 else list(range(12, 24))) + CLOBBER_V]
    out.append(f"      : {', '.join(clob)});")
    out.append("  Fq r;")
    for i in range(12):
        out.append(f"  r.l[{i}] = R{i // 4}.{'xyzw'[i % 4]};")
    out.append("  return r;")
    out.append("}")


def main():
    if "--selftest" in sys.argv:
        selftest()
        selftest_loose()
        return
    if "--loose" in sys.argv:
        out = ["// GENERATED by gen_field_mul30.py --loose -- do not edit; edit the generator.", "// clang-format off",
               "// Fq product / square on 13 x 30-bit limbs in registers (normalised, value < 2^386): a * b * 2^-390 mod q, < q + ab / 2^390."]
        emit_fn_loose("fq30_mul_asm", gen(False, True), False, out)
        emit_fn_loose("fq30_sqr_asm", gen(True, True), True, out)
        out.append("// clang-format on")
        sys.stdout.write("\n".join(out) + "\n")
        return
    out = ["// GENERATED by gen_field_mul30.py -- do not edit; edit the generator.", "// clang-format off",
           "// Fq product / square, a * b * 2^-390 mod q, canonical 12 x u32 in and out; radix-2^30 core on physical registers."]
    emit_fn("fq30h_mul_fn", gen(False), 24, out)
    emit_fn("fq30h_sqr_fn", gen(True), 12, out)
    out.append("// clang-format on")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

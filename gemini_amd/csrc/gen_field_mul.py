#!/usr/bin/env python3
"""Emits field_mul_gen.inc: fully unrolled product-scanning (FIPS) Montgomery multiplication
for gfx950, one inline-asm statement per column half.

Why generated asm and not C: on gfx950 v_mad_u64_u32 and v_addc_co_u32 both issue at half rate
(~4.5 cycles per wave64 instruction, measured with tools/ubench_isa), and the cheapest way to
accumulate a 32x32 product into a 96-bit column accumulator is exactly

    v_mad_u64_u32 acc[0:1], vcc, a_i, b_j, acc[0:1]     ; 64-bit accumulate, carry-out in vcc
    v_addc_co_u32 acc2, vcc, 0, acc2, vcc               ; fold the carry

hipcc does not produce that pair from C (it emits a 64-bit add plus two moves per product, or a
64-bit compare), and pads every asm statement with an s_nop, so a whole column goes into one
statement.  The operand limit of an asm statement (30) is why a.b and m.p halves are separate.

Run:  python3 gen_field_mul.py > field_mul_gen.inc   (the .inc is committed; this is its source)
"""
import sys


def mac_str(xi, yi):
    return f"v_mad_u64_u32 %0, vcc, %{xi}, %{yi}, %0\\n\\tv_addc_co_u32 %1, vcc, 0, %1, vcc"


def emit_macs(pairs, xs_kind, ys_kind, out):
    """pairs: list of (x_expr, y_expr).  Distinct operands get distinct asm operand numbers."""
    ops = []  # (expr, constraint)
    index = {}

    def opnum(expr, cons):
        key = (expr, cons)
        if key not in index:
            index[key] = len(ops) + 2
            ops.append(key)
        return index[key]

    lines = []
    for x, y in pairs:
        xi = opnum(x, xs_kind)
        if y == "one":  # lo += x: the inline constant 1 as the multiplier
            lines.append(f"v_mad_u64_u32 %0, vcc, %{xi}, 1, %0\\n\\tv_addc_co_u32 %1, vcc, 0, %1, vcc")
            continue
        yi = opnum(y, ys_kind)
        lines.append(mac_str(xi, yi))
    assert len(ops) + 2 <= 30, "asm operand limit"
    body = "\\n\\t".join(lines)
    ins = ", ".join(f'"{c}"({e})' for e, c in ops)
    out.append(f'  asm("{body}" : "+v"(lo), "+v"(hi) : {ins} : "vcc");')


def gen_mul(name, params, n, square=False):
    out = []
    out.append(f"// ---- {name}: N = {n} limbs, {2 * n * n + n} v_mad_u64_u32 ----")
    out.append(f"template <> GM_DEV Fp<{params}> fp_mul<{params}>(const Fp<{params}>& a, const Fp<{params}>& b) {{")
    out.append(f"  using P = {params};")
    out.append(f"  uint32_t m[{n}];")
    out.append(f"  uint32_t t[{n}];")
    out.append("  uint64_t lo = 0;")
    out.append("  uint32_t hi = 0;")
    for k in range(2 * n - 1):
        i0 = max(0, k - n + 1)
        i1 = min(k, n - 1)
        out.append(f"  // column {k}")
        emit_macs([(f"a.l[{i}]", f"b.l[{k - i}]") for i in range(i0, i1 + 1)], "v", "v", out)
        # reduction terms m[i] * p[k-i]: for k < n: i in [0, k-1]; for k >= n: i in [k-n+1, n-1]
        if k < n:
            red = [(f"m[{i}]", f"P::MOD[{k - i}]") for i in range(0, k)]
        else:
            red = [(f"m[{i}]", f"P::MOD[{k - i}]") for i in range(k - n + 1, n)]
        if red:
            emit_macs(red, "v", "s", out)
        if k < n:
            out.append(f"  m[{k}] = (uint32_t)lo * P::INV;")
            emit_macs([(f"m[{k}]", "P::MOD[0]")], "v", "s", out)
        else:
            out.append(f"  t[{k - n}] = (uint32_t)lo;")
        out.append("  lo = (lo >> 32) | ((uint64_t)hi << 32);")
        out.append("  hi = 0;")
    out.append(f"  t[{n - 1}] = (uint32_t)lo;")
    out.append(f"  Fp<{params}> r;")
    out.append("#pragma unroll")
    out.append(f"  for (int i = 0; i < {n}; i++) r.l[i] = t[i];")
    out.append(f"  fp_cond_sub<{params}>(r, (uint32_t)(lo >> 32));")
    out.append("  return r;")
    out.append("}")
    return out


def gen_wide(params, n):
    """Lazy reduction (src/misc.rs:235-266 `ip_unsafe` does the same on the CPU): acc += a * b as INTEGERS into 2n limbs -- the
    product half of the multiplication alone, {n*n} multiply-adds -- and ONE Montgomery reduction for the sum of up to 16 products
    -- r^2 = 0.205 x 2^512, so the accumulator has ONE MORE limb than a product (2n + 1: room for 2^32 / 0.205 products) and the
    reduction folds that limb back in through one ordinary product.  The inner products of a sumcheck message
    (time_prover.rs:105-118) are sums of such products."""
    out = []
    out.append(f"// ---- {params}: wide accumulate ({n * n + 2 * n} v_mad_u64_u32) and its reduction ({n * n + n}) ----")
    out.append(f"struct FpWide_{params} {{ uint32_t l[{2 * n + 1}]; }};")
    out.append(f"GM_DEV void fp_mac_wide(FpWide_{params}& acc, const Fp<{params}>& a, const Fp<{params}>& b) {{")
    out.append("  uint64_t lo = 0;")
    out.append("  uint32_t hi = 0;")
    for k in range(2 * n - 1):
        i0 = max(0, k - n + 1)
        i1 = min(k, n - 1)
        out.append(f"  // column {k}")
        emit_macs([(f"acc.l[{k}]", "one")] + [(f"a.l[{i}]", f"b.l[{k - i}]") for i in range(i0, i1 + 1)], "v", "v", out)
        out.append(f"  acc.l[{k}] = (uint32_t)lo;")
        out.append("  lo = (lo >> 32) | ((uint64_t)hi << 32);")
        out.append("  hi = 0;")
    out.append(f"  // column {2 * n - 1}")
    emit_macs([(f"acc.l[{2 * n - 1}]", "one")], "v", "v", out)
    out.append(f"  acc.l[{2 * n - 1}] = (uint32_t)lo;")
    out.append(f"  acc.l[{2 * n}] += (uint32_t)(lo >> 32);  // the extra limb: no carry out of it below 2^32 / 0.205 products")
    out.append("}")
    # reduction: T (2n limbs) -> T / R mod p, result < 2^(32n) + p before the conditional subtractions
    out.append(f"GM_DEV Fp<{params}> fp_redc_wide(const FpWide_{params}& T) {{")
    out.append(f"  using P = {params};")
    out.append(f"  uint32_t m[{n}];")
    out.append(f"  uint32_t t[{n}];")
    out.append("  uint64_t lo = 0;")
    out.append("  uint32_t hi = 0;")
    for k in range(2 * n):
        out.append(f"  // column {k}")
        emit_macs([(f"T.l[{k}]", "one")], "v", "v", out)
        if k < n:
            red = [(f"m[{i}]", f"P::MOD[{k - i}]") for i in range(0, k)]
        else:
            red = [(f"m[{i}]", f"P::MOD[{k - i}]") for i in range(k - n + 1, n)]
        if red:
            emit_macs(red, "v", "s", out)
        if k < n:
            out.append(f"  m[{k}] = (uint32_t)lo * P::INV;")
            emit_macs([(f"m[{k}]", "P::MOD[0]")], "v", "s", out)
        else:
            out.append(f"  t[{k - n}] = (uint32_t)lo;")
        out.append("  lo = (lo >> 32) | ((uint64_t)hi << 32);")
        out.append("  hi = 0;")
    out.append(f"  Fp<{params}> r;")
    out.append("#pragma unroll")
    out.append(f"  for (int i = 0; i < {n}; i++) r.l[i] = t[i];")
    out.append("  // the low 2n limbs are below 2^(64 n): their quotient is below 2^(32 n) + p -- the carry word and up to two more multiples of p go")
    out.append(f"  fp_cond_sub<{params}>(r, (uint32_t)lo);")
    out.append(f"  fp_cond_sub<{params}>(r, 0u);")
    out.append(f"  fp_cond_sub<{params}>(r, 0u);")
    out.append(f"  // the extra limb h stands for h 2^(64 n): times R^-1 that is h R = h R^2 R^-1, one ordinary product")
    out.append(f"  Fp<{params}> h = Fp<{params}>::zero();")
    out.append(f"  h.l[0] = T.l[{2 * n}];")
    out.append(f"  return fp_add<{params}>(r, fp_mul<{params}>(h, Fp<{params}>::r2()));")
    out.append("}")
    return out


def main():
    out = ["// GENERATED by gen_field_mul.py -- do not edit; edit the generator.", "// clang-format off"]
    out += gen_mul("Fq", "FqParams", 12)
    out += gen_mul("Fr", "FrParams", 8)
    out += gen_wide("FrParams", 8)
    out.append("// clang-format on")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

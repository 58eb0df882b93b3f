"""The lazy-reduction pair of gemini_amd/csrc/field_mul_gen.inc (fp_mac_wide: acc += a * b as integers into 17 limbs; fp_redc_wide:
one Montgomery reduction of the sum) checked WITHOUT a GPU: every generated asm statement is a chain of v_mad_u64_u32 / v_addc_co_u32
into a 96-bit column accumulator, which this test rewrites as C on unsigned __int128, compiles for the host and holds against Python
integers -- random operands, (r - 1)^2 repeated thousands of times (the sum passes 2^512 at the fifth product: r^2 = 0.205 x 2^512,
the reason for the seventeenth limb), and the ordinary product the reduction uses for that limb.  Also: the committed .inc is what
the generator prints.  Reference for the technique: src/misc.rs:235-266 (`ip_unsafe`)."""
import os
import random
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gemini_amd", "csrc")
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def test_committed_inc_is_the_generators_output():
    out = subprocess.run([sys.executable, os.path.join(CSRC, "gen_field_mul.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(CSRC, "field_mul_gen.inc")).read()


def _host_program(tmp_path):
    src = open(os.path.join(CSRC, "field_mul_gen.inc")).read()
    fr_mul = src[src.index("// ---- Fr: N = 8"):src.index("// ---- FrParams: wide")]
    wide = src[src.index("struct FpWide_FrParams"):src.index("// clang-format on")]

    def conv(m):
        operands = [o.strip() for o in re.findall(r'"[vs]"\(([^)]*(?:\([^)]*\))?[^)]*)\)', m.group(2))]
        val = lambda t: operands[int(t[1:]) - 2] if t.startswith("%") else t
        lines = []
        for ins in m.group(1).split("\\n\\t"):
            if ins.startswith("v_mad_u64_u32"):
                a = [x.strip() for x in ins.split(",")]
                assert a[0].endswith("%0") and a[4] == "%0" and a[1] == "vcc"
                lines.append(f"{{ unsigned __int128 t_ = (unsigned __int128)lo + (uint64_t)({val(a[2])}) * (uint64_t)({val(a[3])}); lo = (uint64_t)t_; hi += (uint32_t)(t_ >> 64); }}")
            else:
                assert ins == "v_addc_co_u32 %1, vcc, 0, %1, vcc", ins
        return "  " + " ".join(lines)

    rx = r'  asm\("(.*?)" : "\+v"\(lo\), "\+v"\(hi\) : (.*?) : "vcc"\);'
    code = """#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define GM_DEV inline
struct FrParams { static constexpr int N = 8;
  static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr uint32_t INV = 0xffffffffu;
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u}; };
constexpr uint32_t FrParams::MOD[8]; constexpr uint32_t FrParams::R2[8];
template <class P> struct Fp { uint32_t l[8]; static Fp zero() { Fp r; memset(&r, 0, sizeof r); return r; } static Fp r2() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = P::R2[i]; return r; } };
template <class P> GM_DEV void fp_cond_sub(Fp<P>& a, uint32_t extra) { uint32_t t[8]; uint64_t b = 0; for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.l[i] - P::MOD[i] - b; t[i] = (uint32_t)d; b = (d >> 63) & 1; } bool ge = (extra != 0) | (b == 0); for (int i = 0; i < 8; i++) a.l[i] = ge ? t[i] : a.l[i]; }
template <class P> GM_DEV Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) { Fp<P> r; uint64_t c = 0; for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)a.l[i] + b.l[i] + c; r.l[i] = (uint32_t)s; c = s >> 32; } fp_cond_sub<P>(r, (uint32_t)c); return r; }
template <class P> GM_DEV Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b);
""" + re.sub(rx, conv, fr_mul) + re.sub(rx, conv, wide) + """
int main() { FpWide_FrParams acc; memset(&acc, 0, sizeof acc); Fp<FrParams> a, b; int n; if (scanf("%d", &n) != 1) return 1;
  for (int k = 0; k < n; k++) { for (int i = 0; i < 8; i++) if (scanf("%x", &a.l[i]) != 1) return 1; for (int i = 0; i < 8; i++) if (scanf("%x", &b.l[i]) != 1) return 1; fp_mac_wide(acc, a, b); }
  for (int i = 0; i < 17; i++) printf("%08x ", acc.l[i]); printf("\\n"); Fp<FrParams> r = fp_redc_wide(acc); for (int i = 0; i < 8; i++) printf("%08x ", r.l[i]); printf("\\n");
  Fp<FrParams> p = fp_mul<FrParams>(a, b); for (int i = 0; i < 8; i++) printf("%08x ", p.l[i]); printf("\\n"); return 0; }
"""
    assert "asm(" not in code  # every statement was understood
    cpp, exe = os.path.join(tmp_path, "wide_host.cpp"), os.path.join(tmp_path, "wide_host")
    open(cpp, "w").write(code)
    subprocess.check_call(["g++", "-O1", "-std=c++17", cpp, "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host compiler")
def test_wide_accumulate_and_reduce_against_big_integers(tmp_path):
    exe = _host_program(str(tmp_path))
    rng = random.Random(5)
    words = lambda v: " ".join("%x" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(8))
    value = lambda line: sum(int(h, 16) << (32 * i) for i, h in enumerate(line.split()))
    rinv = pow(1 << 256, -1, R)
    for n in (1, 2, 4, 5, 19, 20, 64, 1000, 4096):
        for kind in ("random", "top", "mixed"):
            if kind == "random":
                xs = [(rng.randrange(R), rng.randrange(R)) for _ in range(n)]
            elif kind == "top":
                xs = [(R - 1, R - 1)] * n
            else:
                xs = [(rng.choice([0, 1, R - 1, rng.randrange(R)]), rng.choice([0, 1, R - 1, rng.randrange(R)])) for _ in range(n)]
            inp = str(n) + "\n" + "\n".join(words(a) + " " + words(b) for a, b in xs) + "\n"
            out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
            total = sum(a * b for a, b in xs)
            assert value(out[0]) == total, (n, kind)  # the integer sum, all 17 limbs
            assert value(out[1]) == total * rinv % R, (n, kind)  # its Montgomery reduction, canonical
            assert value(out[2]) == xs[-1][0] * xs[-1][1] * rinv % R  # the ordinary product beside it

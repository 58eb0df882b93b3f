#!/usr/bin/env python3
"""Generates tools/_build/ubench_regs.hip: instruction-mix microbenchmarks on PHYSICAL registers (bank / operand-kind effects). Dev tool."""
import sys
tests = {}
def rep(lines, n=32):
    out = []
    i = 0
    while len(out) < n:
        out.append(lines[i % len(lines)]); i += 1
    return out
tests["mad_dep_bank01"] = rep(["v_mad_u64_u32 v[48:49], vcc, v24, v37, v[48:49]"])            # src banks 0,1 = acc banks
tests["mad_dep_bank23"] = rep(["v_mad_u64_u32 v[48:49], vcc, v26, v39, v[48:49]"])            # src banks 2,3
tests["mad_dep_bank22"] = rep(["v_mad_u64_u32 v[48:49], vcc, v26, v30, v[48:49]"])            # both src bank 2
tests["mad_dep_sgpr"] = rep(["v_mad_u64_u32 v[48:49], vcc, v26, s4, v[48:49]"])
tests["mad_4acc"] = rep([f"v_mad_u64_u32 v[{a}:{a+1}], vcc, v26, v39, v[{a}:{a+1}]" for a in (48, 50, 52, 54)])
tests["mad_varsrc"] = rep([f"v_mad_u64_u32 v[48:49], vcc, v{24+i}, v{36-i}, v[48:49]" for i in range(13)])
tests["alignbit"] = rep([f"v_alignbit_b32 v{50+i}, v{24+i}, v{25+i}, 30" for i in range(6)])
tests["lshl_or"] = rep([f"v_lshl_or_b32 v{50+i}, v{24+i}, 30, v{25+i}" for i in range(6)])
tests["and_lit"] = rep([f"v_and_b32 v{50+i}, 0x3fffffff, v{24+i}" for i in range(6)])
tests["lshr"] = rep([f"v_lshrrev_b32 v{50+i}, 30, v{24+i}" for i in range(6)])
tests["lshr_b64"] = rep([f"v_lshrrev_b64 v[{50+2*i}:{51+2*i}], 30, v[{24+2*i}:{25+2*i}]" for i in range(3)])
tests["bfe_u32"] = rep([f"v_bfe_u32 v{50+i}, v{24+i}, 3, 30" for i in range(6)])
tests["and_or_b32"] = rep([f"v_and_or_b32 v{50+i}, v{24+i}, v{30+i}, v{31+i}" for i in range(6)])
tests["add3_u32"] = rep([f"v_add3_u32 v{50+i}, v{24+i}, v{30+i}, v{31+i}" for i in range(6)])
tests["mul_lo_sgpr"] = rep([f"v_mul_lo_u32 v{50+i}, v{24+i}, s4" for i in range(6)])
tests["mov_lit"] = rep([f"v_mov_b32 v{50+i}, 0x12345678" for i in range(6)])
# one reduction-style column: 8 mads then the m / shift tail
col = [f"v_mad_u64_u32 v[48:49], vcc, v{24+i}, v{37 if i%2 else 38}, v[48:49]" for i in range(4)] + \
      [f"v_mad_u64_u32 v[48:49], vcc, v{i}, s{4+i}, v[48:49]" for i in range(4)] + \
      ["v_mul_lo_u32 v8, v48, s17", "v_and_b32 v8, 0x3fffffff, v8", "v_mad_u64_u32 v[48:49], vcc, v8, s4, v[48:49]",
       "v_alignbit_b32 v48, v49, v48, 30", "v_lshrrev_b32 v49, 30, v49"]
tests["column_mix13"] = rep(col, 26)
clob = ", ".join(f'"v{i}"' for i in list(range(0, 13)) + list(range(24, 40)) + list(range(48, 56))) + ', "vcc", ' + ", ".join(f'"s{i}"' for i in range(4, 18))
print("#include <hip/hip_runtime.h>\n#include <stdint.h>\n#include <stdio.h>")
for name, lines in tests.items():
    body = "\\n\\t".join(lines)
    print(f'__global__ void k_{name}(uint32_t* out, int iters) {{\n  asm volatile("s_mov_b32 s4, 0x3fffaaab\\n\\ts_mov_b32 s5, 0x27fbffff\\n\\ts_mov_b32 s6, 0x153ffffb\\n\\ts_mov_b32 s7, 0x2affffac\\n\\ts_mov_b32 s17, 0x3ffcfffd" ::: {clob});')
    print(f'  for (int it = 0; it < iters; it++) asm volatile("{body}" ::: {clob});')
    print('  uint32_t r; asm volatile("v_mov_b32 %0, v48" : "=v"(r) :: ' + clob + ');\n  out[blockIdx.x * blockDim.x + threadIdx.x] = r;\n}')
print("""template <class F> float timeit(F f) { hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms; }
int main() { uint32_t* out; (void)hipMalloc(&out, 4 * 256 * 8 * 256);
  for (int wps : {8, 2, 1}) { const int blocks = 256 * wps, iters = 2000; printf("waves/SIMD = %d\\n", wps);""")
for name, lines in tests.items():
    print(f'    {{ float ms = timeit([&] {{ k_{name}<<<blocks, 256>>>(out, iters); }}); double n = (double)blocks * 4 * iters * {len(lines)}; printf("  %-18s %8.3f ms  %.2f cyc/wave-instr/SIMD\\n", "{name}", ms, 1024.0 * 2.4e9 * ms * 1e-3 / n); }}')
print("  }\n  return 0; }")

// BLS12-381 Fq Montgomery product for x86-64 with BMI2 + ADX: CIOS without the extra carry word (q < 2^381 leaves the top
// bit of the top limb free), two carry chains (adcx / adox) fed by flag-preserving mulx -- the arrangement of gnark-crypto /
// blst, written out for six limbs.  Operands below q, result below q.  1.75-2 x the rate of the portable loop in
// host_field.hpp, which stays as the path of every other host (and of the Fr field) and as this one's cross-check
// (tests/test_host_field_cpu.py).  The host tail of a one-call MSM is ~2000 of these in sequence (window Horner).
#pragma once
#include <cstdint>
#include <cstdlib>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define GM_HAVE_FQ_ADX 1
namespace gmh {
// one round: t += a * b[i]; m = t0 * inv; t = (t + m q) / 2^64
#define GM_ADX_ROUND(OFF)                                                                                      \
  "xorq %%rax, %%rax\n\t"                                                                                      \
  "movq " #OFF "(%[b]), %%rdx\n\t"                                                                             \
  "mulxq 0(%[a]), %%rax, %%r14\n\t"                                                                            \
  "adoxq %%rax, %%r8\n\t"                                                                                      \
  "adcxq %%r14, %%r9\n\t"                                                                                      \
  "mulxq 8(%[a]), %%rax, %%r14\n\t"                                                                            \
  "adoxq %%rax, %%r9\n\t"                                                                                      \
  "adcxq %%r14, %%r10\n\t"                                                                                     \
  "mulxq 16(%[a]), %%rax, %%r14\n\t"                                                                           \
  "adoxq %%rax, %%r10\n\t"                                                                                     \
  "adcxq %%r14, %%r11\n\t"                                                                                     \
  "mulxq 24(%[a]), %%rax, %%r14\n\t"                                                                           \
  "adoxq %%rax, %%r11\n\t"                                                                                     \
  "adcxq %%r14, %%r12\n\t"                                                                                     \
  "mulxq 32(%[a]), %%rax, %%r14\n\t"                                                                           \
  "adoxq %%rax, %%r12\n\t"                                                                                     \
  "adcxq %%r14, %%r13\n\t"                                                                                     \
  "mulxq 40(%[a]), %%rax, %%r14\n\t"                                                                           \
  "adoxq %%rax, %%r13\n\t"                                                                                     \
  "movq $0, %%rax\n\t"                                                                                         \
  "adcxq %%rax, %%r14\n\t"                                                                                     \
  "adoxq %%rax, %%r14\n\t"                                                                                     \
  "movq %[inv], %%rdx\n\t"                                                                                     \
  "imulq %%r8, %%rdx\n\t"                                                                                      \
  "xorq %%rax, %%rax\n\t"                                                                                      \
  "mulxq 0(%[q]), %%rax, %%rbx\n\t"                                                                            \
  "adcxq %%r8, %%rax\n\t"                                                                                      \
  "movq %%rbx, %%r8\n\t"                                                                                       \
  "adcxq %%r9, %%r8\n\t"                                                                                       \
  "mulxq 8(%[q]), %%rax, %%r9\n\t"                                                                             \
  "adoxq %%rax, %%r8\n\t"                                                                                      \
  "adcxq %%r10, %%r9\n\t"                                                                                      \
  "mulxq 16(%[q]), %%rax, %%r10\n\t"                                                                           \
  "adoxq %%rax, %%r9\n\t"                                                                                      \
  "adcxq %%r11, %%r10\n\t"                                                                                     \
  "mulxq 24(%[q]), %%rax, %%r11\n\t"                                                                           \
  "adoxq %%rax, %%r10\n\t"                                                                                     \
  "adcxq %%r12, %%r11\n\t"                                                                                     \
  "mulxq 32(%[q]), %%rax, %%r12\n\t"                                                                           \
  "adoxq %%rax, %%r11\n\t"                                                                                     \
  "adcxq %%r13, %%r12\n\t"                                                                                     \
  "mulxq 40(%[q]), %%rax, %%r13\n\t"                                                                           \
  "adoxq %%rax, %%r12\n\t"                                                                                     \
  "movq $0, %%rax\n\t"                                                                                         \
  "adcxq %%rax, %%r13\n\t"                                                                                     \
  "adoxq %%r14, %%r13\n\t"

__attribute__((target("bmi2,adx"), noinline)) static void fq_mul_adx(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* q, uint64_t inv) {
  uint64_t t0, t1, t2, t3, t4, t5;
  asm volatile(
      "xorq %%r8, %%r8\n\txorq %%r9, %%r9\n\txorq %%r10, %%r10\n\txorq %%r11, %%r11\n\txorq %%r12, %%r12\n\txorq %%r13, %%r13\n\t"
      GM_ADX_ROUND(0) GM_ADX_ROUND(8) GM_ADX_ROUND(16) GM_ADX_ROUND(24) GM_ADX_ROUND(32) GM_ADX_ROUND(40)
      "movq %%r8, %[t0]\n\tmovq %%r9, %[t1]\n\tmovq %%r10, %[t2]\n\tmovq %%r11, %[t3]\n\tmovq %%r12, %[t4]\n\tmovq %%r13, %[t5]\n\t"
      : [t0] "=&m"(t0), [t1] "=&m"(t1), [t2] "=&m"(t2), [t3] "=&m"(t3), [t4] "=&m"(t4), [t5] "=&m"(t5)
      : [a] "r"(a), [b] "r"(b), [q] "r"(q), [inv] "m"(inv)
      : "rax", "rbx", "rdx", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "cc", "memory");
  // t < 2q: one conditional subtraction
  const uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
  uint64_t s[6];
  uint64_t borrow = 0;
  for (int i = 0; i < 6; i++) {
    const unsigned __int128 d = (unsigned __int128)t[i] - q[i] - borrow;
    s[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  for (int i = 0; i < 6; i++) r[i] = borrow ? t[i] : s[i];
}
// decided once: both instruction sets present and not switched off (GM_HOST_ADX=0, for A/B runs and odd hosts)
static inline bool fq_adx_usable() {
  static const bool ok = [] {
    const char* e = getenv("GM_HOST_ADX");
    if (e && e[0] == '0') return false;
    __builtin_cpu_init();
    return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
  }();
  return ok;
}
#undef GM_ADX_ROUND
}  // namespace gmh
#endif

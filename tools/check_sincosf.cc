// Exhaustive pin of sdslam_amd/csrc/sd_sincosf.h against the host libm (glibc 2.35):
// every float in [0, 6.2832] (1.09e9 values) for both sinf and cosf.
//   g++ -O2 -mfma -ffp-contract=off tools/check_sincosf.cc -o /tmp/check_sincosf -lm && /tmp/check_sincosf
// Optional args: stride (default 1 = exhaustive).
#include "../sdslam_amd/csrc/sd_sincosf.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
  float hi = 6.2832f;
  uint32_t hb = sdsc::f2u(hi);
  unsigned long long n = 0, ms = 0, mc = 0;
  for (uint64_t b = 0; b <= hb; b += stride) {
    uint32_t bb = (uint32_t)b;
    float x;
    memcpy(&x, &bb, 4);
    float s = sinf(x), c = cosf(x);
    float s2 = sdsc::sinf_glibc(x), c2 = sdsc::cosf_glibc(x);
    if (sdsc::f2u(s) != sdsc::f2u(s2)) { if (ms < 5) printf("sin x=%a libm %a mine %a\n", x, s, s2); ms++; }
    if (sdsc::f2u(c) != sdsc::f2u(c2)) { if (mc < 5) printf("cos x=%a libm %a mine %a\n", x, c, c2); mc++; }
    n++;
  }
  printf("checked %llu floats: sin mismatches %llu, cos mismatches %llu\n", n, ms, mc);
  return (ms || mc) ? 1 : 0;
}

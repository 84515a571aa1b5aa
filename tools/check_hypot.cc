// Pin of sdslam_amd/csrc/sd_hypot.h against the host libm hypot (glibc 2.35) on random inputs.
//   g++ -O2 -ffp-contract=off tools/check_hypot.cc -o /tmp/check_hypot -lm && /tmp/check_hypot [count]
#include "../sdslam_amd/csrc/sd_hypot.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  long count = argc > 1 ? atol(argv[1]) : 200000000L;
  uint64_t st = 88172645463325252ull;
  long mism = 0;
  for (long i = 0; i < count; i++) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    uint64_t a = st;
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    uint64_t b = st;
    int ea = (int)(a % 180) - 120, eb = (int)((a >> 20) % 180) - 120;
    if (i % 3 == 0) eb = ea + (int)((b >> 50) % 7) - 3;
    if (i % 1000 == 7) { ea -= 400; eb -= 400; }
    if (i % 1000 == 9) { ea += 500; eb += 500; }
    double x = ldexp(1.0 + (double)(a >> 12) / 4503599627370496.0, ea), y = ldexp(1.0 + (double)(b >> 12) / 4503599627370496.0, eb);
    if (b & 1) x = -x;
    if (i % 100000 == 3) y = 0;
    if (hypot(x, y) != sdsc::hypot_glibc(x, y)) {
      if (mism < 5) printf("x=%a y=%a libm=%a mine=%a\n", x, y, hypot(x, y), sdsc::hypot_glibc(x, y));
      mism++;
    }
  }
  printf("checked %ld inputs: hypot mismatches %ld\n", count, mism);
  return mism ? 1 : 0;
}

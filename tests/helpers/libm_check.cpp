/* Compares the device-side restatements in agarcl_amd/csrc/agar_libm.inl (compiled for the host) with
 * the host libm.  Usage: libm_check <start_hex> <count> <stride>  -> prints mismatch counts. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define AG_DEV static inline
#include "../../agarcl_amd/csrc/agar_libm.inl"
static int same(float a, float b) { if (a != a && b != b) return 1; return ag_asuint(a) == ag_asuint(b); }
int main(int argc, char **argv) {
  unsigned long long start = strtoull(argv[1], 0, 16), count = strtoull(argv[2], 0, 10), stride = strtoull(argv[3], 0, 10);
  unsigned long long bs = 0, bc = 0, ba = 0; unsigned fs = 0, fc = 0, fa = 0;
  for (unsigned long long k = 0; k < count; k++) {
    uint32_t u = (uint32_t)(start + k * stride);
    float x = ag_asfloat(u);
    if (!same(sinf(x), ag_sinf(x))) { if (!bs) fs = u; bs++; }
    if (!same(cosf(x), ag_cosf(x))) { if (!bc) fc = u; bc++; }
    if (!same(atanf(x), ag_atanf(x))) { if (!ba) fa = u; ba++; }
  }
  printf("%llu %llu %llu %08x %08x %08x\n", bs, bc, ba, fs, fc, fa);
  return 0;
}

// fh_udiv.hpp — division of a cell number by a divisor that is fixed for a launch (a map's row and slice sizes) or for a segment
// (the sub-block of the unknown lattice inside its local box).
//
// The compiler expands a 32-bit integer division into 20+ vector instructions whatever the operands; the path search splits every
// popped cell index into (x, y, z) and the decomposition every candidate voxel number — twice each.  With inv = floor(2^32 / d):
//   q' = mulhi(n, inv) = floor(n inv / 2^32),   n inv / 2^32 = n / d - n r / (d 2^32)   (r = 2^32 mod d < d)
// so for n < 2^28 the estimate is short of n / d by less than 1/16: q' is the quotient or one less, and ONE correction step gives the
// quotient exactly.  d = 1: inv = 2^32 - 1 gives q' = n - 1 (n >= 1), corrected to n.
// inverse_fp computes the same inv with one double division (the device has no cheap 64-bit integer division): the fraction of
// 2^32 / d is 0 or at least 1/d >= 2^-20 for d <= 2^20, far above the rounding error of the quotient (< 2^-21), so the truncation of
// the rounded quotient is the floor.  Plain C++ (host and device); tests/cpp/test_udiv.cpp checks both claims.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FH_UDIV_FN __host__ __device__ inline
#else
#define FH_UDIV_FN inline
#endif

namespace fhu {

FH_UDIV_FN unsigned inverse(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }
FH_UDIV_FN unsigned inverse_fp(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(4294967296.0 / (double)d); }  // d <= 2^20
FH_UDIV_FN int div(int n, int d, unsigned inv) {  // 0 <= n < 2^28, d >= 1
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned q = __umulhi((unsigned)n, inv);
#else
  unsigned q = (unsigned)(((uint64_t)(unsigned)n * inv) >> 32);
#endif
  if ((unsigned)n - q * (unsigned)d >= (unsigned)d) q++;
  return (int)q;
}

}  // namespace fhu

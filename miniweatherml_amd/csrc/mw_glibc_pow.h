// =====================================================================================================
// mw_glibc_pow.h -- pow(x, y) with the bits of the host's glibc, for the STRICT kernel path (mw_dycore_set_strict(h, 1)).
//
// The strict path runs the reference's operation order with FMA contraction off; +, -, *, /, sqrt are IEEE on both sides, so what
// was left between it and the CPU oracle was the last bit of pow (device library vs glibc): the upwind selectors and limiter
// switches of the scheme turn such a bit into differences of 1e-8 of a (tiny) field on the thermal-bubble cases.  This header
// restates glibc's algorithm (glibc >= 2.28, sysdeps/ieee754/dbl-64/e_pow.c; S. Nagy, "optimized-routines" pow):
//     log(x)  = k ln2 + log(c) + log1p(z/c - 1)        x = 2^k z, c from a 128-entry table, degree-7 polynomial, result hi + lo
//     y log x = ehi + elo                                (an FMA recovers the product's rounding error)
//     exp     = 2^(k/128) from a 128-entry table * (1 + degree-5 polynomial of the reduced argument)
// with exactly the contractions of the x86-64 FMA build (`__pow_fma`, what the ifunc resolves to on every FMA-capable host):
// each statement below is one instruction of that routine, in its operand order.  The tables are the C library's
// (mw_glibc_pow_tables.h, read from the installed libm by tools/gen_glibc_pow_tables.py).  Only the main path is restated
// (x positive normal, 2^-65 < |y| < 2^63, result neither overflowing nor subnormal) -- everything the dycore produces; other
// arguments return false and the caller uses the device library's pow.
// Checked bit for bit against the running libm: tests/test_glibc_pow.py (host build on CPU; the device routine on the GPU).
// =====================================================================================================
#pragma once
#include "mw_glibc_pow_tables.h"
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define MW_GP_HD __host__ __device__ __forceinline__
#else
#define MW_GP_HD inline
#endif

namespace mw {

struct GpLogEntry { double invc, logc, logctail; };
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const GpLogEntry gp_log_tab[128] = MW_GP_LOG_TABLE;
__device__ const unsigned long long gp_exp_tab[256] = MW_GP_EXP_TABLE;
#else
static const GpLogEntry gp_log_tab[128] = MW_GP_LOG_TABLE;
static const unsigned long long gp_exp_tab[256] = MW_GP_EXP_TABLE;
#endif

MW_GP_HD uint64_t gp_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
MW_GP_HD double gp_double(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

// true: *res = glibc's pow(x, y).  false: an argument outside the restated main path.
MW_GP_HD bool glibc_pow_main(double x, double y, double *res) {
#pragma clang fp contract(off)
  const uint64_t ix = gp_bits(x), iy = gp_bits(y);
  const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
  if (topx - 1u > 0x7fdu) return false;                                   // x zero, subnormal, negative, inf or nan
  if ((topy & 0x7ffu) - 0x3beu > 0x7fu) return false;                     // |y| tiny or huge
  // ---- log_inline
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  const double z = gp_double(iz), kd = (double)k;
  const double invc = gp_log_tab[i].invc, logc = gp_log_tab[i].logc, logctail = gp_log_tab[i].logctail;
  const double t1 = __builtin_fma(kd, MW_GP_LN2HI, logc);
  const double r = __builtin_fma(z, invc, -1.0);
  const double ar = r * MW_GP_A0;
  const double lo1 = __builtin_fma(kd, MW_GP_LN2LO, logctail);
  const double p12 = __builtin_fma(r, MW_GP_A2, MW_GP_A1);
  const double p34 = __builtin_fma(r, MW_GP_A4, MW_GP_A3);
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double d12 = t1 - t2;
  const double ar3 = r * ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo2 = d12 + r;
  const double p56 = __builtin_fma(r, MW_GP_A6, MW_GP_A5);
  const double hi = t2 + ar2;
  const double d2h = t2 - hi;
  const double p36 = __builtin_fma(p56, ar2, p34);
  const double lo4 = d2h + ar2;
  const double pp = __builtin_fma(ar2, p36, p12);
  double lo = lo1 + lo2;
  lo = lo + lo3;
  lo = lo + lo4;
  lo = __builtin_fma(ar3, pp, lo);
  const double loghi = hi + lo;
  const double logtail = (hi - loghi) + lo;
  // ---- y * log(x) = ehi + elo
  const double ehi = y * loghi;
  const double e1 = __builtin_fma(loghi, y, -ehi);
  const double elo = __builtin_fma(y, logtail, e1);
  // ---- exp_inline(ehi, elo)
  const uint32_t abstop = (uint32_t)(gp_bits(ehi) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u > 0x3eu) return false;                              // |y log x| < 2^-54 or >= 512: tiny / overflow / underflow paths
  const double kdS = __builtin_fma(ehi, MW_GP_INVLN2N, MW_GP_SHIFT);
  const uint64_t ki = gp_bits(kdS);
  const double kd2 = kdS - MW_GP_SHIFT;
  double rr = __builtin_fma(kd2, MW_GP_NEGLN2HIN, ehi);
  rr = __builtin_fma(kd2, MW_GP_NEGLN2LON, rr);
  const uint64_t idx = 2 * (ki & 127);
  const uint64_t sbits = gp_exp_tab[idx + 1] + (ki << 45);
  rr = elo + rr;
  const double c23 = __builtin_fma(rr, MW_GP_C3, MW_GP_C2);
  const double tr = rr + gp_double(gp_exp_tab[idx]);
  const double r2 = rr * rr;
  const double c45 = __builtin_fma(rr, MW_GP_C5, MW_GP_C4);
  const double q1 = __builtin_fma(c23, r2, tr);
  const double r4 = r2 * r2;
  const double q2 = __builtin_fma(c45, r4, q1);
  const double scale = gp_double(sbits);
  *res = __builtin_fma(q2, scale, scale);
  return true;
}

// exp(x) with the bits of glibc's exp (e_exp.c, the x86-64 FMA build `__exp_fma`): the same table and polynomial as pow's last step.
// Main path: 2^-54 <= |x| < 512.  (Used by the strict Kessler path: the saturation vapour pressure, microphysics_kessler.h:304.)
MW_GP_HD bool glibc_exp_main(double x, double *res) {
#pragma clang fp contract(off)
  const uint32_t abstop = (uint32_t)(gp_bits(x) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u > 0x3eu) return false;
  const double kdS = __builtin_fma(x, MW_GP_INVLN2N, MW_GP_SHIFT);
  const uint64_t ki = gp_bits(kdS);
  const double kd = kdS - MW_GP_SHIFT;
  double r = __builtin_fma(kd, MW_GP_NEGLN2HIN, x);
  r = __builtin_fma(kd, MW_GP_NEGLN2LON, r);
  const uint64_t idx = 2 * (ki & 127);
  const uint64_t sbits = gp_exp_tab[idx + 1] + (ki << 45);
  const double c23 = __builtin_fma(r, MW_GP_C3, MW_GP_C2);
  const double tr = r + gp_double(gp_exp_tab[idx]);
  const double r2 = r * r;
  const double c45 = __builtin_fma(r, MW_GP_C5, MW_GP_C4);
  const double q1 = __builtin_fma(c23, r2, tr);
  const double r4 = r2 * r2;
  const double q2 = __builtin_fma(r4, c45, q1);
  const double scale = gp_double(sbits);
  *res = __builtin_fma(scale, q2, scale);
  return true;
}

} // namespace mw

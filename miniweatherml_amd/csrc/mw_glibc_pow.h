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
__device__ const double gp_sincos_tab[440] = MW_GP_SINCOS_TABLE;
#else
static const GpLogEntry gp_log_tab[128] = MW_GP_LOG_TABLE;
static const unsigned long long gp_exp_tab[256] = MW_GP_EXP_TABLE;
static const double gp_sincos_tab[440] = MW_GP_SINCOS_TABLE;
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

// cos(x) with the bits of glibc's cos (sysdeps/ieee754/dbl-64/s_sin.c, the x86-64 FMA build `__cos_fma`) for |x| < 105414350:
//   |x| < 2^-27: 1;   |x| < 0.855469: do_cos -- |x| = k/128 + d, cos from table entry k and short polynomials of d;
//   |x| < 2.426265: cos x = sin(pi/2 - |x|) with pi/2 as hp0 + hp1;   else reduce_sincos (x = n pi/2 + a + da, four-part pi/2) and
//   do_cos / do_sin by quadrant.  do_sin: a Taylor polynomial for |a| < 0.126, else the table.
// (The cosine bells of the initial states -- perturb_temperature.h:63, dynamics_euler_stratified_wenofv.h:1131 -- take arguments in
// [0, pi/2]; the sponges -- sponge_layer.h:70, horizontal_sponge.h -- in [0, pi].)  The polynomial constants are usncs.h's, the
// table is the library's (mw_glibc_pow_tables.h).  Larger arguments (the multi-precision reductions) return false.
MW_GP_HD double gp_do_cos(double a, double da) {                            // do_cos(a, da): a = |.| k/128 + d
#pragma clang fp contract(off)
  const double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
  const double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10, big = 0x1.8p45;
  const double aa = gp_double(gp_bits(a) & 0x7fffffffffffffffull);
  const double dx = (a < 0.0) ? -da : da;
  const double u = aa + big;
  const int k = (int)(uint32_t)gp_bits(u) << 2;
  double d = aa - (u - big);
  const double cs = gp_sincos_tab[k + 2];
  d = d + dx;
  const double xx = d * d;
  const double ps = __builtin_fma(xx, sn5, sn3);
  const double x3 = d * xx;
  const double s = __builtin_fma(x3, ps, d);
  double pc = __builtin_fma(xx, cs6, cs4);
  pc = __builtin_fma(xx, pc, cs2);
  const double c = xx * pc;
  double cor = __builtin_fma(-s, gp_sincos_tab[k + 1], gp_sincos_tab[k + 3]);        // ccs - s * ssn
  cor = __builtin_fma(-c, cs, cor);                                                    // ... - cs * c
  cor = __builtin_fma(-s, gp_sincos_tab[k], cor);                                      // ... - sn * s
  return cs + cor;
}
MW_GP_HD double gp_do_sin(double a, double da) {                            // sin(a + da): TAYLOR_SIN for |a| < 0.126, else do_sin
#pragma clang fp contract(off)
  const double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
  const double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10, big = 0x1.8p45;
  const double aa = gp_double(gp_bits(a) & 0x7fffffffffffffffull);
  if (aa < 0x1.020c49ba5e354p-3) {
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19,
                 s5 = -0x1.addffc2fcdf59p-26;
    const double xx = a * a;
    double t = __builtin_fma(xx, s5, s4);
    t = __builtin_fma(xx, t, s3);
    t = __builtin_fma(xx, t, s2);
    t = __builtin_fma(xx, t, s1);
    const double hd = da * cs2;
    t = __builtin_fma(t, a, -hd);
    const double r = __builtin_fma(xx, t, da);
    return a + r;
  }
  if (!(0.0 < a)) da = -da;
  const double u = aa + big;
  const int k = (int)(uint32_t)gp_bits(u) << 2;
  const double x = aa - (u - big);
  const double xx = x * x;
  const double ps = __builtin_fma(xx, sn5, sn3);
  const double x3 = x * xx;
  const double sd = __builtin_fma(x3, ps, da);
  double pc = __builtin_fma(xx, cs6, cs4);
  pc = __builtin_fma(xx, pc, cs2);
  const double s = x + sd;
  const double c0 = xx * pc;
  const double c = __builtin_fma(x, da, c0);
  const double sn = gp_sincos_tab[k];
  double cor = __builtin_fma(s, gp_sincos_tab[k + 3], gp_sincos_tab[k + 1]);           // ssn + s * ccs
  cor = __builtin_fma(-c, sn, cor);                                                    // ... - sn * c
  const double t = __builtin_fma(s, gp_sincos_tab[k + 2], cor);                       // cs * s + cor
  const double r = sn + t;
  return gp_double((gp_bits(r) & 0x7fffffffffffffffull) | (gp_bits(a) & 0x8000000000000000ull));
}
MW_GP_HD bool glibc_cos_main(double xin, double *res) {
#pragma clang fp contract(off)
  const uint32_t kx = (uint32_t)(gp_bits(xin) >> 32) & 0x7fffffffu;
  if (kx > 0x419921fau) return false;                                     // |x| >= 105414350 (or inf / nan): the multi-precision reductions
  if (kx <= 0x3e3fffffu) { *res = 1.0; return true; }                     // |x| < 2^-27
  const double ax = gp_double(gp_bits(xin) & 0x7fffffffffffffffull);
  if (kx <= 0x3feb5fffu) { *res = gp_do_cos(ax, (xin >= 0.0) ? 0.0 : -0.0); return true; }
  if (kx <= 0x400368fcu) {                                                // cos x = sin(pi/2 - |x|)
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const double y = hp0 - ax;
    const double a = y + hp1;
    const double da = (y - a) + hp1;
    *res = gp_do_sin(a, da);
    return true;
  }
  // reduce_sincos: x = n pi/2 + a + da, pi/2 = mp1 + mp2 + pp3 + pp4
  const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
  const double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27, pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
  const double t = __builtin_fma(xin, hpinv, toint);
  const double xn = t - toint;
  const int n = (int)(gp_bits(t) & 3);
  double y = __builtin_fma(-xn, mp1, xin);
  y = __builtin_fma(-xn, mp2, y);
  const double t2 = __builtin_fma(-xn, pp3, y);
  double db = y - t2;
  db = __builtin_fma(-pp3, xn, db);
  const double a = __builtin_fma(-xn, pp4, t2);
  double e = t2 - a;
  e = __builtin_fma(-xn, pp4, e);
  const double da = db + e;
  const int m = n + 1;
  const double r = (m & 1) ? gp_do_cos(a, da) : gp_do_sin(a, da);
  *res = (m & 2) ? -r : r;
  return true;
}

} // namespace mw

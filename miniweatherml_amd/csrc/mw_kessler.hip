// =====================================================================================================
// mw_kessler.hip -- Microphysics_Kessler::time_step for gfx950
// reference: model/modules/microphysics_kessler.h:99-162 (time_step) and :234-339 (kessler()).
//
// Three kernels, no host synchronisation (rainsplit is read from device memory by the kernels themselves, so the
// reference's device->host minval sync (:276) disappears; the kernel that does not apply returns at once):
//   k_kessler_prep   [K1,K2,K3]  per cell: r, rhalf, velqr, CFL limit; block min -> one 64-bit atomicMin.  Also stores the
//                                pre-update rain flux r*qr*velqr of the levels that sit just above a z chunk.
//   k_kessler_chunks [K4,K5]     rainsplit == 1 (the normal case: the rain CFL step is ~16 s, the dycore's < 1 s): the
//                                reference's "all sed(k) first, then adjust" (:288-335) only needs the PRE-update value of
//                                r*qr*velqr at k+1, so a column is cut into z chunks that are swept top-down
//                                independently: thread = (column, chunk), the flux from above is carried in a register
//                                and taken from prep's boundary array at the chunk top.  Every field is read once and
//                                written once (72 B/cell); ~10x more threads than columns hide the latency.
//   k_kessler_column [K4,K5]     rainsplit > 1: one thread per column, all sub-cycles, top-down.
// Arrays are the (nz,ncol) views of the coupler fields (DataManager::get_lev_col), column index fastest:
// thread i walks k with perfectly coalesced accesses.
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include "mw_glibc_pow.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>

namespace mw {

// ---------------------------------------------------------------------------------------------------------------
// log / exp / sqrt for this module.  With rain in every wavefront the sweep is fp64-transcendental bound (4 log + 6 exp +
// 1 sqrt per cell); the device library's log is 92 VALU instructions, its exp 36, its sqrt 22 (they also serve denormals,
// infinities and NaNs to < 1 ulp).  The arguments here are positive, finite density / pressure ratios, so:
//   kes_log  : fdlibm's scheme -- x = 2^k m, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), log m = 2s + s^3 R(s^2) with the 7-term
//              minimax R (|error| < 2^-58.45), ln 2 split in two: 36 instructions, <= 1 ulp (tests: <= 2.3e-16 relative
//              against the host libm over 1e-130 .. 150).  x == 0 -> -inf (so that exp(a log 0) = 0 = pow(0, a)); tiny x is scaled.
//   kes_exp  : k = rint(x / ln 2), r = x - k ln 2 (two-part), degree-13 Taylor polynomial (|r| <= 0.347: remainder 4e-18),
//              ldexp: 21 instructions, <= 2.3e-16 relative.  The argument is clamped to [-746, 710]: -inf -> 0, overflow -> inf.
//   kes_sqrt : v_rsq_f64 + two coupled Newton steps (arguments in [1e-3, 1e3]).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double kes_rcp(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}
// one Newton step: 2e-15 relative (used where the quotient only feeds a rate term; the temperature <-> theta round trip and the
// mixing ratios keep the fully rounded reciprocal)
__device__ __forceinline__ double kes_rcp1(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  return r + r * (1.0 - x * r);
}
__device__ __forceinline__ double kes_log(double x) {
#pragma clang fp contract(fast)
  const bool zero = (x == 0.0);
  const bool tiny = (x < 0x1p-1000);
  const double xs = tiny ? x * 0x1p+200 : x;
  int e = __builtin_amdgcn_frexp_exp(xs);
  double m = __builtin_amdgcn_frexp_mant(xs);                 // [0.5, 1)
  const bool lo = (m < 0.70710678118654752440);
  m = lo ? m + m : m;                                         // [sqrt(1/2), sqrt(2))
  e = (lo ? e - 1 : e) - (tiny ? 200 : 0);
  const double f = m - 1.0;
  const double s = f * kes_rcp(2.0 + f);
  const double hfsq = 0.5 * f * f, z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1, dk = (double)e;
  const double r = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
  return zero ? -__builtin_huge_val() : r;
}
__device__ __forceinline__ double kes_exp(double x) {
#pragma clang fp contract(fast)
  x = fmin(fmax(x, -746.0), 710.0);
  const double k = __builtin_rint(x * 1.44269504088896340736);
  double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;                              // 1/13!
  p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0; p = p * r + 1.0 / 3628800.0; p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;     p = p * r + 1.0 / 5040.0;     p = p * r + 1.0 / 720.0;     p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;        p = p * r + 1.0 / 6.0;        p = p * r + 0.5;             p = p * r + 1.0;
  p = p * r + 1.0;
  return __builtin_amdgcn_ldexp(p, (int)k);
}
__device__ __forceinline__ double kes_sqrt(double a) {
#pragma clang fp contract(fast)
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = 0.5 - h * g;
  g = g + g * r; h = h + h * r;
  r = 0.5 - h * g;
  g = g + g * r; h = h + h * r;
  return g + h * (a - g * g) * 1.0;                           // one correction of the residual (h ~ 1/(2 sqrt a))
}
// x^a for x >= 0 as exp(a log x): device pow is ~230 fp64-VALU instructions, kes_log + kes_exp 57, and the powers of
// r*qr in the evaporation formula share one log.  |a log x| <= ~25 here, so the result is within ~3e-15 relative
// of pow (x = 0 gives exp(-inf) = 0 = pow(0, a) for a > 0).
__device__ __forceinline__ double pow_pos(double x, double a) { return kes_exp(a * kes_log(x)); }
// 1/x to full fp64 accuracy (v_rcp_f64 + two Newton steps, 5 instructions; an IEEE division is ~25).  A rain-free cell of the
// reference's formulas holds ~17 divisions against one exp/log pair and one exp: they dominate, so a/b is evaluated as a*rcp(b)
// (<= 1.5 ulp from the quotient; the parity tolerance of this module is 1e-12).
__device__ __forceinline__ double rcp64(double x) { return kes_rcp(x); }
// The same where x is a rain quantity: rain-free wavefronts (most of the domain) skip the log/exp pair.  exp(a log 0) =
// exp(-inf) = 0, so the short cut returns exactly what the formula returns; mixed wavefronts evaluate the formula.
__device__ __forceinline__ double pow_rain(double x, double a) {
  if (!__any(x != 0.0)) return 0.0;
  return kes_exp(a * kes_log(x));
}

struct KesP {
  int nz; long long ncol;
  double dz, dt;
  double R_d, cp_d, R_v, p0;       // module constants, microphysics_kessler.h:29-41
};

// workspace layout (doubles): [0..15] unused | velqr(nz,ncol) | theta | qv | qc | qr
// The minimum of dt2d over the rank (:276) is one 64-bit word that every block of the CFL pass atomicMin's into.  It lives in a
// library-owned pair of words per (device, stream) -- kessler_min_words below -- used alternately: call n accumulates into word n & 1
// while block (0, 0) of its CFL pass resets word (n + 1) & 1 to +inf for call n + 1 (whose readers finished with it one call ago,
// in stream order).  No separate initialisation launch (round 5; rounds 1-4 spent a 5 us launch per call on it).
#define MW_KES_INF 0x7FF0000000000000ull

// Terminal velocity of rain, :256-260, from the density fields (one definition for the CFL pass, the chunk sweep and the
// column sweep: the 16 bytes per cell of a stored copy cost more than recomputing it, and a rain-free wavefront skips it).
__device__ __forceinline__ double kessler_velqr(double rho_r, double rd, double ird, double rho0) {      // ird = 1 / rd
#pragma clang fp contract(off)
  if (!__any(rho_r != 0.0)) return 0.0;                       // wave-uniform; exact: 36.34 * 0^0.1364 * rhalf = 0
  const double qr = rho_r * ird;                              // :140
  const double r = 0.001 * rd;                                // :256
  const double rhalf = kes_sqrt(rho0 * ird);                  // :257  rho(0,i)/rho(k,i)
  return 36.34 * pow_rain(qr * r, 0.1364) * rhalf;            // :260
}

// K2 + K3 (:255-279).  thread = (column i, a range of levels).  What the pass must deliver is (a) rainsplit = ceil(dt / min dt2d) and
// (b) the pre-update rain flux of the level just above every z chunk of the sweep.  dt2d = 0.8 dz / velqr is below dt -- i.e. can
// raise rainsplit above 1 -- only where velqr > 0.8 dz / dt (573 m/s at the dycore's CFL step: never); every other cell contributes a
// value >= dt, and min(.., dt) gives the same rainsplit (cells without rain contribute exactly dt in the reference, :266).  A cell
// is PROVEN harmless without any transcendental: for 0.001 rho_r <= 1 the power (qr r)^0.1364 is <= 1, so
// velqr <= 36.34 sqrt(rho0 / rho) < limit  <=>  36.34^2 rho0 (1 + 1e-9) < limit^2 rho.  Only wavefronts with an unproven cell -- or on a
// chunk-boundary level, whose flux the sweep needs -- evaluate the fall speed (:260).  The pass is then a stream over two fields.
__global__ __launch_bounds__(256) void k_kessler_prep(KesP p, const double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                      double *__restrict__ flux_top, int chunk, int klevels,
                                                      unsigned long long *dtmax_bits, unsigned long long *next_bits) {
#pragma clang fp contract(off)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *next_bits = MW_KES_INF;      // the NEXT call's word (see MW_KES_INF)
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * klevels, k1 = min(k0 + klevels, p.nz);
  double dtc = p.dt;                                          // every cell's contribution is capped at dt (see above)
  if (i < p.ncol) {
    const double rho0 = rho_d[i];
    const double lim = 0.8 * p.dz / p.dt;
    const double lhs = (36.34 * 36.34) * rho0 * (1.0 + 1.0e-9), lim2 = lim * lim;
    constexpr int KL = 5;                                     // (= klevels of the launch) all loads of the thread first
    double rdv[KL], rrv[KL];
    bool rain = false;
#pragma unroll
    for (int m = 0; m < KL; m++) { const long long idx = (long long)min(k0 + m, p.nz - 1) * p.ncol + i; rrv[m] = rho_r[idx]; }
#pragma unroll
    for (int m = 0; m < KL; m++) rain = rain || (rrv[m] != 0.0);
    // (round 5) a wavefront without rain in any of its cells -- most of the domain -- needs no density either: a rain-free cell's fall speed
    // is 36.34 * 0^0.1364 * rhalf = 0 whatever its density, it contributes exactly dt (:266) and its flux r qr velqr is 0: the pass reads
    // ONE field there, and the chunk-boundary fluxes are written as the zeros they are.
    if (!__any(rain)) {
#pragma unroll
      for (int m = 0; m < KL; m++) {
        const int k = k0 + m;
        if (k < k1 && k > 0 && k % chunk == 0) flux_top[(long long)(k / chunk - 1) * p.ncol + i] = 0.0;
      }
    } else {
#pragma unroll
    for (int m = 0; m < KL; m++) { const long long idx = (long long)min(k0 + m, p.nz - 1) * p.ncol + i; rdv[m] = rho_d[idx]; }
#pragma unroll
    for (int m = 0; m < KL; m++) {
      const int k = k0 + m;
      if (k >= k1) break;
      const double rd = rdv[m], rr = rrv[m];
      const bool boundary = (k > 0 && k % chunk == 0);        // block-uniform
      const bool proven = (0.001 * rr <= 1.0) && (lhs < lim2 * rd);
      if (boundary || __any(!proven)) {
        const double ird = rcp64(rd);
        const double qr = rr * ird, r = 0.001 * rd;           // :140, :256
        const double velqr = kessler_velqr(rr, rd, ird, rho0);   // :257-260
        if (boundary) flux_top[(long long)(k / chunk - 1) * p.ncol + i] = r * qr * velqr;   // flux entering the chunk below
        if (k < p.nz - 1) {                                   // :262-268
          const double zk = (k + 0.5) * p.dz, zk1 = (k + 1 + 0.5) * p.dz;   // zmid, :137
          const double c = (velqr > 1.e-10) ? 0.8 * (zk1 - zk) / velqr : p.dt;
          dtc = fmin(dtc, c);
        }
      }
    }
    }
  }
  // block min (wave shuffle, then LDS across the 4 waves); positive doubles order like their bit patterns
  for (int off = 32; off > 0; off >>= 1) { double o = __shfl_down(dtc, off, 64); dtc = fmin(dtc, o); }
  __shared__ double smin[4];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) smin[wv] = dtc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    if (m < p.dt || (blockIdx.x == 0 && blockIdx.y == 0))                    // one word, one atomic per block that has something to say
      atomicMin(dtmax_bits, (unsigned long long)__double_as_longlong(m));     // :276 minval(dt2d), capped at dt
  }
}

// One cell of one rain sub-cycle (:288-335): sedimentation from the pre-update fluxes, then the adjustment terms.
// In: T, qv, qc, qr, velqr (pre-update), flux_above = r*qr*velqr of level k+1 (pre-update).  Out: updated T..velqr;
// returns this cell's pre-update flux (the next lower cell's flux_above).
// TEMPERATURE FORM.  The reference converts temp -> theta = temp / pk with the Exner function pk = (p/p0)^(R_d/cp) (:142-143), works
// on theta, and converts back temp = theta pk (:160).  pk is constant during the call and enters the column physics only as
// pk * theta (= temp, :303-305) and as the latent-heating factor lv / (cp pk) on theta (:321) -- which is lv / cp on temp.  The Exner
// function therefore cancels exactly; evaluating in temperature form drops one log, one exp and two reciprocals per cell and
// differs from the reference's operation order by rounding only (tests: 1e-12 against the oracle, which keeps the theta form; a
// vapour-free state now comes back bit for bit).  pc (:258) needs p/p0 itself, not pk.
__device__ __forceinline__ double kessler_cell(const KesP &p, int k, double rd, double rho0, double pp0, double dt0, double flux_above,
                                               double &T, double &qv, double &qc, double &qr, double &velqr, double &precl_acc) {
#pragma clang fp contract(off)
  const double Rd = p.R_d, cp = p.cp_d;
  const double psl = p.p0 / 100;                              // :246
  const double rhoqr = 1000., lv = 2.5e6;                     // :247-248
  const int nz = p.nz;
  double r = 0.001 * rd;                                              // :256
  double rhalf = kes_sqrt(rho0 * rcp64(rd));                          // :257
  double pc = 3.8 * kes_rcp1(pp0 * psl);                                      // :258  pow(pk, cp/Rd) = pressure/p0 (pk = (pressure/p0)^(Rd/cp))
  double zk = (k + 0.5) * p.dz;
  // sedimentation (:288-299) from pre-update values
  if (k == 0) precl_acc = precl_acc + rho0 * qr * velqr / rhoqr;      // :292 (rho(0,i) qr(0,i) velqr(0,i))
  double flux_here = r * qr * velqr;
  double sed;
  if (k == nz - 1) {
    double zm = (k - 1 + 0.5) * p.dz;
    sed = -dt0 * qr * velqr * kes_rcp1(0.5 * (zk - zm));                 // :295
  } else {
    double zp = (k + 1 + 0.5) * p.dz;
    sed = dt0 * (flux_above - flux_here) * kes_rcp1(r * (zp - zk));      // :297-298
  }
  // adjustment terms (:302-335)
  double qrprod = qc - (qc - dt0 * fmax(0.001 * (qc - 0.001), 0.0)) * kes_rcp1(1 + dt0 * 2.2 * pow_rain(qr, 0.875));
  qc = fmax(qc - qrprod, 0.0);
  qr = fmax(qr + qrprod + sed, 0.0);
  double tmp = T - 36.;                                               // :303  pk * theta - 36
  const double rtmp = kes_rcp1(tmp);
  double qvs = pc * kes_exp(17.27 * (T - 273.) * rtmp);
  double prod = (qv - qvs) * kes_rcp1(1. + qvs * (4093. * lv / cp) * (rtmp * rtmp));
  double tmp1 = 0.0;                                                   // rain-free wavefront: (1.6 + 0) * 0 / (..) * (..) = 0
  if (__any(r * qr != 0.0)) {
    const double lrq = kes_log(r * qr);                                // one log for the two powers of r*qr
    const double rqvs = kes_rcp1(qvs);
    tmp1 = dt0 * (((1.6 + 124.9 * kes_exp(0.2046 * lrq)) * kes_exp(0.525 * lrq)) * kes_rcp1(2550000. * pc * ((1.0 / 3.8) * rqvs) + 540000.)) *
           (fmax(qvs - qv, 0.0) * (kes_rcp1(r) * rqvs));
  }
  double tmp2 = fmax(-prod - qc, 0.0);
  double tmp3 = qr;
  double ern = fmin(tmp1, fmin(tmp2, tmp3));
  double cond = fmax(prod, -qc);
  T = T + (lv / cp) * (cond - ern);                                   // :321  theta += lv / (cp pk) (..), times pk
  qv = fmax(qv - cond + ern, 0.0);
  qc = qc + cond;
  qr = qr - ern;
  velqr = 36.34 * pow_rain(qr * r, 0.1364) * rhalf;                   // :331 (qr changed by ern: its own log)
  return flux_here;
}

// rainsplit == 1: thread = (column i, z chunk c), top-down over the chunk's levels.
__device__ __forceinline__ void kessler_chunks_body(const KesP &p, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                    double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                    double *__restrict__ temp, double *__restrict__ precl,
                                                    const double *__restrict__ flux_top, int chunk) {
#pragma clang fp contract(off)
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.ncol) return;
  const int c = blockIdx.y;
  const int k_lo = c * chunk, k_hi = min(k_lo + chunk, p.nz) - 1;
  const double dt0 = p.dt / 1.0;                              // :280
  const double rho0 = rho_d[i];
  double precl_acc = 0;
  double flux_above = (k_hi < p.nz - 1) ? flux_top[(long long)c * p.ncol + i] : 0.0;
  // software prefetch: the next level's five inputs are in flight while this level is computed
  long long idx = (long long)k_hi * p.ncol + i;
  double rd = rho_d[idx], T_in = temp[idx], rv_in = rho_v[idx], rc_in = rho_c[idx], rr_in = rho_r[idx];
  for (int k = k_hi; k >= k_lo; k--) {
    idx = (long long)k * p.ncol + i;
    const long long nidx = (long long)max(k - 1, k_lo) * p.ncol + i;
    const double rd_n = rho_d[nidx], T_n = temp[nidx], rv_n = rho_v[nidx], rc_n = rho_c[nidx], rr_n = rho_r[nidx];
    double pressure = p.R_d * rd * T_in + p.R_v * rv_in * T_in;          // :141
    const double pp0 = pressure * (1.0 / p.p0);
    const double ird = rcp64(rd);
    double qv = rv_in * ird, qc = rc_in * ird, qr = rr_in * ird;         // :138-140
    double T = T_in;                                                     // (temperature form: see kessler_cell)
    double velqr = kessler_velqr(rr_in, rd, ird, rho0);                  // :260 (as k_kessler_prep saw it)
    flux_above = kessler_cell(p, k, rd, rho0, pp0, dt0, flux_above, T, qv, qc, qr, velqr, precl_acc);
    // :154-161 [K5].  (Round 5: cloud / rain that came in as zero and go out as zero over a whole wavefront -- most of the domain -- are not
    //  stored again: 16 of the 72 bytes a cell costs.  Wave-uniform, so that no cache line is written in part.)
    const double rc_out = qc * rd, rr_out = qr * rd;
    rho_v[idx] = qv * rd;
    if (__any(rc_in != 0.0 || rc_out != 0.0)) rho_c[idx] = rc_out;
    if (__any(rr_in != 0.0 || rr_out != 0.0)) rho_r[idx] = rr_out;
    if (__any(T != T_in)) temp[idx] = T;                                 // (no phase change anywhere in the wavefront: the temperature comes back bit for bit)
    rd = rd_n; T_in = T_n; rv_in = rv_n; rc_in = rc_n; rr_in = rr_n;
  }
  if (c == 0) precl[i] = precl_acc / 1.0;                                // :332-334
}

// rainsplit > 1: thread = column, all sub-cycles, top-down.
__device__ __forceinline__ void kessler_column_body(const KesP &p, int rainsplit, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                    double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                    double *__restrict__ temp, double *__restrict__ precl, double *__restrict__ ws) {
#pragma clang fp contract(off)
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.ncol) return;
  const long long n = (long long)p.nz * p.ncol;
  double *w_velqr = ws, *w_theta = ws + n, *w_qv = ws + 2 * n, *w_qc = ws + 3 * n, *w_qr = ws + 4 * n;
  const double dt0 = p.dt / (double)rainsplit;                // :280
  const int nz = p.nz;
  const double rho0 = rho_d[i];
  double precl_acc = 0;                                       // :270-272
  for (int nt = 0; nt < rainsplit; nt++) {
    const bool first = (nt == 0), lastp = (nt == rainsplit - 1);
    double flux_above = 0;                                    // r(k+1)*qr(k+1)*velqr(k+1), pre-update
    for (int k = nz - 1; k >= 0; k--) {
      long long idx = (long long)k * p.ncol + i;
      double rd = rho_d[idx];
      double T_in = temp[idx], rv_in = rho_v[idx];
      double pressure = p.R_d * rd * T_in + p.R_v * rv_in * T_in;        // :141
      const double pp0 = pressure / p.p0;                                // (of the call's input state, like the reference's pk, :141-142)
      double T, qv, qc, qr, velqr;
      if (first) {
        const double ird = rcp64(rd);
        qv = rv_in * ird; qc = rho_c[idx] * ird; qr = rho_r[idx] * ird;   // :138-140
        T = T_in;                                                         // (temperature form: see kessler_cell)
        velqr = kessler_velqr(rho_r[idx], rd, ird, rho0);                 // :260 (as k_kessler_prep saw it)
      } else { T = w_theta[idx]; qv = w_qv[idx]; qc = w_qc[idx]; qr = w_qr[idx]; velqr = w_velqr[idx]; }
      flux_above = kessler_cell(p, k, rd, rho0, pp0, dt0, flux_above, T, qv, qc, qr, velqr, precl_acc);
      if (lastp) {                                                        // :154-161 [K5]
        rho_v[idx] = qv * rd; rho_c[idx] = qc * rd; rho_r[idx] = qr * rd;
        temp[idx] = T;
      } else { w_theta[idx] = T; w_qv[idx] = qv; w_qc[idx] = qc; w_qr[idx] = qr; w_velqr[idx] = velqr; }
    }
  }
  precl[i] = precl_acc / (double)rainsplit;                               // :332-334
}

// K4 + K5 in ONE launch (round 5; two launches before, one of them a no-op): the grid of the z-chunk sweep; with rainsplit == 1 every
// workgroup sweeps its chunk, with rainsplit > 1 (wave-uniform: one word) the workgroups of chunk 0 run whole columns and the others leave.
#ifndef MW_KES_SWEEP
#define MW_KES_SWEEP 0          // 0: k_kessler_chunks + k_kessler_column (one of them returns at once); 1 / 2: one merged launch, capped at 128 VGPRs / uncapped (A/B)
#endif
#if MW_KES_SWEEP == 1
__global__ __launch_bounds__(256, 4)
#else
__global__ __launch_bounds__(256)
#endif
void k_kessler_sweep(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                       double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                       double *__restrict__ temp, double *__restrict__ precl,
                                                       const unsigned long long *dtmax_bits, double *__restrict__ ws,
                                                       const double *__restrict__ flux_top, int chunk) {
#pragma clang fp contract(off)
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  const int rainsplit = (int)ceil(p.dt / dt_max);             // :279
  if (rainsplit == 1) kessler_chunks_body(p, rho_v, rho_c, rho_r, rho_d, temp, precl, flux_top, chunk);
  else if (blockIdx.y == 0) kessler_column_body(p, rainsplit, rho_v, rho_c, rho_r, rho_d, temp, precl, ws);
}

__global__ __launch_bounds__(256) void k_kessler_chunks(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c, double *__restrict__ rho_r,
                                                        const double *__restrict__ rho_d, double *__restrict__ temp, double *__restrict__ precl,
                                                        const unsigned long long *dtmax_bits, const double *__restrict__ flux_top, int chunk) {
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  if ((int)ceil(p.dt / dt_max) != 1) return;                  // k_kessler_column handles rainsplit > 1
  kessler_chunks_body(p, rho_v, rho_c, rho_r, rho_d, temp, precl, flux_top, chunk);
}
__global__ __launch_bounds__(256) void k_kessler_column(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c, double *__restrict__ rho_r,
                                                        const double *__restrict__ rho_d, double *__restrict__ temp, double *__restrict__ precl,
                                                        const unsigned long long *dtmax_bits, double *__restrict__ ws) {
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  const int rainsplit = (int)ceil(p.dt / dt_max);             // :279
  if (rainsplit == 1) return;                                 // done by k_kessler_chunks
  kessler_column_body(p, rainsplit, rho_v, rho_c, rho_r, rho_d, temp, precl, ws);
}

// ---------------------------------------------------------------------------------------------------------------
// STRICT path (mw_kessler_set_strict(1)): the reference's formulas in the reference's operation order -- theta form, IEEE divisions,
// no contraction -- with glibc's pow and exp (mw_glibc_pow.h): BIT-identical to the CPU oracle (and, with it, to the reference on a
// glibc host).  Not a performance path: two launches, thread = cell for K1-K3 and thread = column for the sub-cycles.
//   workspace: w_velqr | w_theta | w_qv | w_qc | w_qr, (nz, ncol) each.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double kes_pow_ref(double x, double y) { double r; if (glibc_pow_main(x, y, &r)) return r; return pow(x, y); }
__device__ __forceinline__ double kes_exp_ref(double x) { double r; if (glibc_exp_main(x, &r)) return r; return exp(x); }

// K1 (:136-144), K2 (:255-273), K3's minimum (:276)
__global__ __launch_bounds__(256) void k_kessler_strict_prep(KesP p, const double *__restrict__ rho_v, const double *__restrict__ rho_c,
                                                             const double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                             const double *__restrict__ temp, double *__restrict__ ws,
                                                             unsigned long long *dtmax_bits, unsigned long long *next_bits) {
#pragma clang fp contract(off)
  if (blockIdx.x == 0 && threadIdx.x == 0) *next_bits = MW_KES_INF;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)p.nz * p.ncol;
  double dtc = __builtin_huge_val();
  if (t < n) {
    const int k = (int)(t / p.ncol);
    const long long i = t - (long long)k * p.ncol;
    double *w_velqr = ws, *w_theta = ws + n, *w_qv = ws + 2 * n, *w_qc = ws + 3 * n, *w_qr = ws + 4 * n;
    const double rho = rho_d[t];
    const double qv = rho_v[t] / rho, qc = rho_c[t] / rho, qr = rho_r[t] / rho;                    // :138-140
    const double pressure = p.R_d * rho * temp[t] + p.R_v * rho_v[t] * temp[t];                     // :141
    const double exner = kes_pow_ref(pressure / p.p0, p.R_d / p.cp_d);                              // :142
    const double theta = temp[t] / exner;                                                           // :143
    const double r = 0.001 * rho;                                                                   // :256
    const double rhalf = sqrt(rho_d[i] / rho);                                                      // :257
    const double velqr = 36.34 * kes_pow_ref(qr * r, 0.1364) * rhalf;                               // :260
    w_velqr[t] = velqr; w_theta[t] = theta; w_qv[t] = qv; w_qc[t] = qc; w_qr[t] = qr;
    if (k < p.nz - 1) {                                                                             // :262-268
      const double zk = (k + 0.5) * p.dz, zk1 = (k + 1 + 0.5) * p.dz;
      dtc = (velqr > 1.e-10) ? 0.8 * (zk1 - zk) / velqr : p.dt;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { double o = __shfl_down(dtc, off, 64); dtc = fmin(dtc, o); }
  __shared__ double smin[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) smin[wv] = dtc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    if (m < __builtin_huge_val()) atomicMin(dtmax_bits, (unsigned long long)__double_as_longlong(m));   // positive doubles order like their bits
  }
}

// K4 (:285-335) + K5 (:154-161): thread = column, all sub-cycles, top-down (sed(k) needs the pre-update flux of level k+1 only)
__global__ __launch_bounds__(256) void k_kessler_strict_column(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                               double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                               double *__restrict__ temp, double *__restrict__ precl,
                                                               const unsigned long long *dtmax_bits, double *__restrict__ ws) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.ncol) return;
  const long long n = (long long)p.nz * p.ncol;
  double *w_velqr = ws, *w_theta = ws + n, *w_qv = ws + 2 * n, *w_qc = ws + 3 * n, *w_qr = ws + 4 * n;
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  const int rainsplit = (int)ceil(p.dt / dt_max);                                                   // :279
  const double dt0 = p.dt / (double)rainsplit;                                                      // :280
  const int nz = p.nz;
  const double Rd = p.R_d, cp = p.cp_d;
  const double psl = p.p0 / 100, rhoqr = 1000., lv = 2.5e6;                                          // :246-248
  const double rho0 = rho_d[i];
  double pr = 0;                                                                                    // precl(i), :270-272
  for (int nt = 0; nt < rainsplit; nt++) {
    double flux_above = 0;                                  // r(k+1) qr(k+1) velqr(k+1), pre-update
    for (int k = nz - 1; k >= 0; k--) {
      const long long idx = (long long)k * p.ncol + i;
      const double rho = rho_d[idx];
      // exner of the call's INPUT state (:141-142; temp / rho_v of this level are only overwritten below, in the last sub-cycle)
      const double pressure = Rd * rho * temp[idx] + p.R_v * rho_v[idx] * temp[idx];
      const double pk = kes_pow_ref(pressure / p.p0, Rd / cp);
      const double r = 0.001 * rho;                                                                 // :256
      const double rhalf = sqrt(rho0 / rho);                                                        // :257
      const double pc = 3.8 / (kes_pow_ref(pk, cp / Rd) * psl);                                     // :258
      double theta = w_theta[idx], qv = w_qv[idx], qc = w_qc[idx], qr = w_qr[idx];
      const double velqr = w_velqr[idx];
      const double zk = (k + 0.5) * p.dz;
      if (k == 0) pr = pr + rho0 * qr * velqr / rhoqr;                                              // :292
      const double flux_here = r * qr * velqr;
      double sed;
      if (k == nz - 1) { const double zm = (k - 1 + 0.5) * p.dz; sed = -dt0 * qr * velqr / (0.5 * (zk - zm)); }       // :295
      else             { const double zp = (k + 1 + 0.5) * p.dz; sed = dt0 * (flux_above - flux_here) / (r * (zp - zk)); }   // :297-298
      flux_above = flux_here;
      // :302-335
      const double qrprod = qc - (qc - dt0 * fmax(0.001 * (qc - 0.001), 0.)) / (1 + dt0 * 2.2 * kes_pow_ref(qr, 0.875));
      qc = fmax(qc - qrprod, 0.);
      qr = fmax(qr + qrprod + sed, 0.);
      const double tmp = pk * theta - 36.;
      const double qvs = pc * kes_exp_ref(17.27 * (pk * theta - 273.) / tmp);
      const double prod = (qv - qvs) / (1. + qvs * (4093. * lv / cp) / (tmp * tmp));
      const double tmp1 = dt0 * (((1.6 + 124.9 * kes_pow_ref(r * qr, 0.2046)) * kes_pow_ref(r * qr, 0.525)) /
                                 (2550000. * pc / (3.8 * qvs) + 540000.)) * (fmax(qvs - qv, 0.) / (r * qvs));
      const double tmp2 = fmax(-prod - qc, 0.);
      const double tmp3 = qr;
      const double ern = fmin(tmp1, fmin(tmp2, tmp3));
      theta = theta + lv / (cp * pk) * (fmax(prod, -qc) - ern);
      qv = fmax(qv - fmax(prod, -qc) + ern, 0.);
      qc = qc + fmax(prod, -qc);
      qr = qr - ern;
      const double velqr_new = 36.34 * kes_pow_ref(qr * r, 0.1364) * rhalf;                         // :331
      if (nt == rainsplit - 1) {                                                                    // :154-161 [K5]
        rho_v[idx] = qv * rho; rho_c[idx] = qc * rho; rho_r[idx] = qr * rho;
        temp[idx] = theta * pk;
      } else { w_theta[idx] = theta; w_qv[idx] = qv; w_qc[idx] = qc; w_qr[idx] = qr; w_velqr[idx] = velqr_new; }
    }
  }
  precl[i] = pr / (double)rainsplit;                                                                // :332-334
}

// diagnostic: the module's log / exp / sqrt on caller-supplied arguments (tests compare them with the host libm)
__global__ __launch_bounds__(256) void k_kessler_math_probe(long long n, const double *__restrict__ x, double *__restrict__ y, int fn) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  y[i] = fn == 0 ? kes_log(x[i]) : fn == 1 ? kes_exp(x[i]) : fn == 2 ? kes_sqrt(x[i]) : kes_rcp(x[i]);
}

} // namespace mw

using namespace mw;

extern "C" {

long long mw_kessler_workspace_bytes(int nz, long long ncol) { return (long long)sizeof(double) * (16 + 5ll * nz * ncol + ((long long)nz / 4 + 1) * ncol); }

int mw_kessler_math_probe(long long n, const double *x, double *y, int fn, void *stream) {
  if (n < 1 || !x || !y || fn < 0 || fn > 3) MW_FAIL("kessler_math_probe: bad argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipLaunchKernelGGL(k_kessler_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, x, y, fn);
  MW_LAUNCH_CHECK();
  return 0;
}

static thread_local int g_kessler_strict = 0;        // per calling thread: a rank harness with one host thread per rank may use different modes side by side
// 1: the strict path (reference operation order, glibc's pow / exp: bit-identical to the CPU oracle); 0: the production kernels
// The pair of min words of a caller: library-owned device memory, allocated and set to +inf once, then kept consistent by the kernels
// themselves (see MW_KES_INF).  -> this call's word and the one its CFL pass resets for the next call.
// Keyed by (device, stream, WORKSPACE): a workspace is one caller's scratch -- two host threads that run Kessler on the same device and
// stream (the thread-per-rank harness on the null stream) bring their own workspaces and so get their own pairs and their own call
// parity; round 5 keyed the pair by (device, stream) alone, and two such threads could interleave get / prep / commit on ONE word
// (a reset between the other thread's CFL pass and its sweep: rainsplit = ceil(dt / inf) = 0).  The words are not IN the workspace:
// the caller may free it and an allocator may hand the same address out again with other contents -- the pair of that key is still
// consistent then (one word holds the last call's minimum, the other +inf).
// (commit = true, after the CFL pass was launched: the call counts -- a call that failed before its CFL pass must not flip the parity,
//  its word for the next call would never have been reset)
static int kessler_min_words(hipStream_t st, const void *workspace, unsigned long long **cur, unsigned long long **next, bool commit = false) {
  struct Entry { unsigned long long *w = nullptr; unsigned long long calls = 0; };
  static std::mutex mu;
  static std::map<std::tuple<int, hipStream_t, const void *>, Entry> tab;
  int dev = 0;
  MW_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  Entry &e = tab[std::make_tuple(dev, st, workspace)];
  if (commit) { e.calls++; return 0; }
  if (!e.w) {
    const unsigned long long init[2] = {MW_KES_INF, MW_KES_INF};
    MW_HIP(hipMalloc(&e.w, 16));
    if (hipMemcpy(e.w, init, 16, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(e.w); e.w = nullptr; MW_FAIL("kessler: initialising the min words failed"); }
  }
  *cur = e.w + (e.calls & 1); *next = e.w + ((e.calls + 1) & 1);
  return 0;
}

int mw_kessler_set_strict(int strict) { g_kessler_strict = strict ? 1 : 0; return 0; }

int mw_kessler_time_step(int nz, long long ncol, double dz, double dt, double *rho_v, double *rho_c, double *rho_r,
                         const double *rho_d, double *temp, double *precl, void *workspace, int *rainsplit_out, void *stream) {
  if (nz < 2 || ncol < 1) MW_FAIL("kessler: need nz >= 2 and ncol >= 1");
  if (dt <= 0) MW_FAIL("kessler.f90 called with nonpositive dt");          // :242
  if (!rho_v || !rho_c || !rho_r || !rho_d || !temp || !precl || !workspace) MW_FAIL("kessler: null pointer");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  KesP p; p.nz = nz; p.ncol = ncol; p.dz = dz; p.dt = dt; p.R_d = 287.; p.cp_d = 1003.; p.R_v = 461.; p.p0 = 1.e5;
  unsigned long long *bits = nullptr, *next_bits = nullptr;
  if (kessler_min_words(st, workspace, &bits, &next_bits)) return 1;
  double *ws = (double *)workspace + 16;
  if (g_kessler_strict) {
    const long long n = (long long)nz * ncol;
    hipLaunchKernelGGL(k_kessler_strict_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, rho_v, rho_c, rho_r, rho_d, temp, ws, bits, next_bits); MW_LAUNCH_CHECK();
    if (kessler_min_words(st, workspace, nullptr, nullptr, true)) return 1;
    hipLaunchKernelGGL(k_kessler_strict_column, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, st, p, rho_v, rho_c, rho_r, rho_d, temp,
                       precl, bits, ws); MW_LAUNCH_CHECK();
    if (rainsplit_out) {
      unsigned long long hb = 0;
      MW_HIP(hipMemcpyAsync(&hb, bits, 8, hipMemcpyDeviceToHost, st));
      MW_HIP(hipStreamSynchronize(st));
      double dt_max; memcpy(&dt_max, &hb, 8);
      *rainsplit_out = (int)std::ceil(dt / dt_max);
    }
    return 0;
  }
  // z chunks of the rainsplit == 1 path: enough (column, chunk) threads to fill the chip, at least 4 levels per chunk
  int chunk = nz;
  for (int c : {25, 20, 16, 12, 10, 8, 5, 4}) if (c < nz) { chunk = c; if (((ncol + 63) / 64) * ((nz + c - 1) / c) >= 16384) break; }
  const int nchunks = (nz + chunk - 1) / chunk;
  double *flux_top = ws + 5ll * nz * ncol;                    // (nchunks-1, ncol)
  const int klevels = 5;                                        // levels per thread of the CFL pass (k_kessler_prep: KL)
  hipLaunchKernelGGL(k_kessler_prep, dim3((unsigned)((ncol + 255) / 256), (unsigned)((nz + klevels - 1) / klevels)), dim3(256), 0, st, p,
                     rho_r, rho_d, flux_top, chunk, klevels, bits, next_bits); MW_LAUNCH_CHECK();
  if (kessler_min_words(st, workspace, nullptr, nullptr, true)) return 1;
#if MW_KES_SWEEP
  hipLaunchKernelGGL(k_kessler_sweep, dim3((unsigned)((ncol + 255) / 256), (unsigned)nchunks), dim3(256), 0, st, p, rho_v, rho_c, rho_r,
                     rho_d, temp, precl, bits, ws, flux_top, chunk); MW_LAUNCH_CHECK();
#else
  hipLaunchKernelGGL(k_kessler_chunks, dim3((unsigned)((ncol + 255) / 256), (unsigned)nchunks), dim3(256), 0, st, p, rho_v, rho_c, rho_r,
                     rho_d, temp, precl, bits, flux_top, chunk); MW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_kessler_column, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, st, p, rho_v, rho_c, rho_r, rho_d, temp,
                     precl, bits, ws); MW_LAUNCH_CHECK();
#endif
  if (rainsplit_out) {
    unsigned long long hb = 0;
    MW_HIP(hipMemcpyAsync(&hb, bits, 8, hipMemcpyDeviceToHost, st));
    MW_HIP(hipStreamSynchronize(st));
    double dt_max; memcpy(&dt_max, &hb, 8);
    *rainsplit_out = (int)std::ceil(dt / dt_max);
  }
  return 0;
}

} // extern "C"

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
#include <cmath>
#include <cstring>

namespace mw {

// x^a for x >= 0 as exp(a log x): device pow is ~230 fp64-VALU instructions, log + exp ~140, and the three powers of
// r*qr in the evaporation/fall-speed formulas share one log.  |a log x| <= ~25 here, so the result is within ~3e-15 relative
// of pow (x = 0 gives exp(-inf) = 0 = pow(0, a) for a > 0).
__device__ __forceinline__ double pow_pos(double x, double a) { return exp(a * log(x)); }
// 1/x to full fp64 accuracy (v_rcp_f64 + two Newton steps, 5 instructions; an IEEE division is ~25).  A rain-free cell of the
// reference's formulas holds ~17 divisions against one exp/log pair and one exp: they dominate, so a/b is evaluated as a*rcp(b)
// (<= 1.5 ulp from the quotient; the parity tolerance of this module is 1e-12).
__device__ __forceinline__ double rcp64(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}
// The same where x is a rain quantity: rain-free wavefronts (most of the domain) skip the log/exp pair.  exp(a log 0) =
// exp(-inf) = 0, so the short cut returns exactly what the formula returns; mixed wavefronts evaluate the formula.
__device__ __forceinline__ double pow_rain(double x, double a) {
  if (!__any(x != 0.0)) return 0.0;
  return exp(a * log(x));
}

struct KesP {
  int nz; long long ncol;
  double dz, dt;
  double R_d, cp_d, R_v, p0;       // module constants, microphysics_kessler.h:29-41
};

// workspace layout (doubles): [0] dt_max bits (as unsigned long long) | velqr(nz,ncol) | theta | qv | qc | qr
__global__ __launch_bounds__(256) void k_kessler_init_min(unsigned long long *dtmax_bits) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dtmax_bits = 0x7FF0000000000000ull;   // +inf
}

// Terminal velocity of rain, :256-260, from the density fields (one definition for the CFL pass, the chunk sweep and the
// column sweep: the 16 bytes per cell of a stored copy cost more than recomputing it, and a rain-free wavefront skips it).
__device__ __forceinline__ double kessler_velqr(double rho_r, double rd, double rho0) {
#pragma clang fp contract(off)
  if (!__any(rho_r != 0.0)) return 0.0;                       // wave-uniform; exact: 36.34 * 0^0.1364 * rhalf = 0
  const double qr = rho_r / rd;                               // :140
  const double r = 0.001 * rd;                                // :256
  const double rhalf = sqrt(rho0 / rd);                       // :257  rho(0,i)/rho(k,i)
  return 36.34 * pow_rain(qr * r, 0.1364) * rhalf;            // :260
}

__global__ __launch_bounds__(256) void k_kessler_prep(KesP p, const double *__restrict__ rho_v, const double *__restrict__ rho_r,
                                                      const double *__restrict__ rho_d, const double *__restrict__ temp,
                                                      double *__restrict__ velqr_out, double *__restrict__ flux_top, int chunk,
                                                      unsigned long long *dtmax_bits) {
#pragma clang fp contract(off)
  // grid-stride over all (k, column) cells: a few thousand workgroups -> a few thousand atomics on the one min word
  // (62500 single-address atomics cost 0.7 ms on MI355X: ~88 per microsecond per address)
  double dtc = __longlong_as_double(0x7FF0000000000000ll);
  const long long n = (long long)p.nz * p.ncol;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx / p.ncol);
    const long long i = idx - (long long)k * p.ncol;
    double rd = rho_d[idx];
    const double rr = rho_r[idx];
    double qr = rr / rd;                                      // :140
    double r = 0.001 * rd;                                    // :256
    double velqr = kessler_velqr(rr, rd, rho_d[i]);           // :257-260
    if (k > 0 && k % chunk == 0) flux_top[(long long)(k / chunk - 1) * p.ncol + i] = r * qr * velqr;   // flux entering the chunk below
    if (k < p.nz - 1) {                                       // :262-268
      double zk = (k + 0.5) * p.dz, zk1 = (k + 1 + 0.5) * p.dz;   // zmid, :137
      double c = (velqr > 1.e-10) ? 0.8 * (zk1 - zk) / velqr : p.dt;
      dtc = fmin(dtc, c);
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
    atomicMin(dtmax_bits, (unsigned long long)__double_as_longlong(m));     // :276 minval(dt2d)
  }
}

// One cell of one rain sub-cycle (:288-335): sedimentation from the pre-update fluxes, then the adjustment terms.
// In: theta, qv, qc, qr, velqr (pre-update), flux_above = r*qr*velqr of level k+1 (pre-update).  Out: updated theta..velqr;
// returns this cell's pre-update flux (the next lower cell's flux_above).
__device__ __forceinline__ double kessler_cell(const KesP &p, int k, double rd, double rho0, double pk, double pp0, double dt0, double flux_above,
                                               double &theta, double &qv, double &qc, double &qr, double &velqr, double &precl_acc) {
#pragma clang fp contract(off)
  const double Rd = p.R_d, cp = p.cp_d;
  const double psl = p.p0 / 100;                              // :246
  const double rhoqr = 1000., lv = 2.5e6;                     // :247-248
  const int nz = p.nz;
  double r = 0.001 * rd;                                              // :256
  double rhalf = sqrt(rho0 * rcp64(rd));                              // :257
  double pc = 3.8 * rcp64(pp0 * psl);                                      // :258  pow(pk, cp/Rd) = pressure/p0 (pk = (pressure/p0)^(Rd/cp))
  double zk = (k + 0.5) * p.dz;
  // sedimentation (:288-299) from pre-update values
  if (k == 0) precl_acc = precl_acc + rho0 * qr * velqr / rhoqr;      // :292 (rho(0,i) qr(0,i) velqr(0,i))
  double flux_here = r * qr * velqr;
  double sed;
  if (k == nz - 1) {
    double zm = (k - 1 + 0.5) * p.dz;
    sed = -dt0 * qr * velqr * rcp64(0.5 * (zk - zm));                 // :295
  } else {
    double zp = (k + 1 + 0.5) * p.dz;
    sed = dt0 * (flux_above - flux_here) * rcp64(r * (zp - zk));      // :297-298
  }
  // adjustment terms (:302-335)
  double qrprod = qc - (qc - dt0 * fmax(0.001 * (qc - 0.001), 0.0)) * rcp64(1 + dt0 * 2.2 * pow_rain(qr, 0.875));
  qc = fmax(qc - qrprod, 0.0);
  qr = fmax(qr + qrprod + sed, 0.0);
  double tmp = pk * theta - 36.;
  const double rtmp = rcp64(tmp);
  double qvs = pc * exp(17.27 * (pk * theta - 273.) * rtmp);
  double prod = (qv - qvs) * rcp64(1. + qvs * (4093. * lv / cp) * (rtmp * rtmp));
  double tmp1 = 0.0;                                                   // rain-free wavefront: (1.6 + 0) * 0 / (..) * (..) = 0
  if (__any(r * qr != 0.0)) {
    const double lrq = log(r * qr);                                    // one log for the two powers of r*qr
    const double rqvs = rcp64(qvs);
    tmp1 = dt0 * (((1.6 + 124.9 * exp(0.2046 * lrq)) * exp(0.525 * lrq)) * rcp64(2550000. * pc * ((1.0 / 3.8) * rqvs) + 540000.)) *
           (fmax(qvs - qv, 0.0) * (rcp64(r) * rqvs));
  }
  double tmp2 = fmax(-prod - qc, 0.0);
  double tmp3 = qr;
  double ern = fmin(tmp1, fmin(tmp2, tmp3));
  double cond = fmax(prod, -qc);
  theta = theta + lv * rcp64(cp * pk) * (cond - ern);
  qv = fmax(qv - cond + ern, 0.0);
  qc = qc + cond;
  qr = qr - ern;
  velqr = 36.34 * pow_rain(qr * r, 0.1364) * rhalf;                   // :331 (qr changed by ern: its own log)
  return flux_here;
}

// rainsplit == 1: thread = (column i, z chunk c), top-down over the chunk's levels.
__global__ __launch_bounds__(256) void k_kessler_chunks(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                        double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                        double *__restrict__ temp, double *__restrict__ precl,
                                                        const unsigned long long *dtmax_bits, const double *__restrict__ velqr_in,
                                                        const double *__restrict__ flux_top, int chunk) {
#pragma clang fp contract(off)
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  if ((int)ceil(p.dt / dt_max) != 1) return;                  // k_kessler_column handles rainsplit > 1
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
    double pk = pow_pos(pp0, p.R_d / p.cp_d);                            // :142 exner
    const double ird = rcp64(rd);
    double qv = rv_in * ird, qc = rc_in * ird, qr = rr_in * ird;         // :138-140
    double theta = T_in * rcp64(pk);                                     // :143
    double velqr = kessler_velqr(rr_in, rd, rho0);                       // :260 (as k_kessler_prep saw it)
    flux_above = kessler_cell(p, k, rd, rho0, pk, pp0, dt0, flux_above, theta, qv, qc, qr, velqr, precl_acc);
    rho_v[idx] = qv * rd; rho_c[idx] = qc * rd; rho_r[idx] = qr * rd;    // :154-161 [K5]
    temp[idx] = theta * pk;
    rd = rd_n; T_in = T_n; rv_in = rv_n; rc_in = rc_n; rr_in = rr_n;
  }
  if (c == 0) precl[i] = precl_acc / 1.0;                                // :332-334
}

__global__ __launch_bounds__(256) void k_kessler_column(KesP p, double *__restrict__ rho_v, double *__restrict__ rho_c,
                                                        double *__restrict__ rho_r, const double *__restrict__ rho_d,
                                                        double *__restrict__ temp, double *__restrict__ precl,
                                                        const unsigned long long *dtmax_bits, double *__restrict__ ws) {
#pragma clang fp contract(off)
  const double dt_max = __longlong_as_double((long long)*dtmax_bits);
  const int rainsplit = (int)ceil(p.dt / dt_max);             // :279
  if (rainsplit == 1) return;                                 // done by k_kessler_chunks
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
      const double pp0 = pressure / p.p0;
      double pk = pow_pos(pp0, p.R_d / p.cp_d);                          // :142 exner
      double theta, qv, qc, qr, velqr;
      if (first) {
        qv = rv_in / rd; qc = rho_c[idx] / rd; qr = rho_r[idx] / rd;      // :138-140
        theta = T_in / pk;                                                // :143
        velqr = kessler_velqr(rho_r[idx], rd, rho0);                      // :260 (as k_kessler_prep saw it)
      } else { theta = w_theta[idx]; qv = w_qv[idx]; qc = w_qc[idx]; qr = w_qr[idx]; velqr = w_velqr[idx]; }
      flux_above = kessler_cell(p, k, rd, rho0, pk, pp0, dt0, flux_above, theta, qv, qc, qr, velqr, precl_acc);
      if (lastp) {                                                        // :154-161 [K5]
        rho_v[idx] = qv * rd; rho_c[idx] = qc * rd; rho_r[idx] = qr * rd;
        temp[idx] = theta * pk;
      } else { w_theta[idx] = theta; w_qv[idx] = qv; w_qc[idx] = qc; w_qr[idx] = qr; w_velqr[idx] = velqr; }
    }
  }
  precl[i] = precl_acc / (double)rainsplit;                               // :332-334
}

} // namespace mw

using namespace mw;

extern "C" {

long long mw_kessler_workspace_bytes(int nz, long long ncol) { return (long long)sizeof(double) * (16 + 5ll * nz * ncol + ((long long)nz / 4 + 1) * ncol); }

int mw_kessler_time_step(int nz, long long ncol, double dz, double dt, double *rho_v, double *rho_c, double *rho_r,
                         const double *rho_d, double *temp, double *precl, void *workspace, int *rainsplit_out, void *stream) {
  if (nz < 2 || ncol < 1) MW_FAIL("kessler: need nz >= 2 and ncol >= 1");
  if (dt <= 0) MW_FAIL("kessler.f90 called with nonpositive dt");          // :242
  if (!rho_v || !rho_c || !rho_r || !rho_d || !temp || !precl || !workspace) MW_FAIL("kessler: null pointer");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  KesP p; p.nz = nz; p.ncol = ncol; p.dz = dz; p.dt = dt; p.R_d = 287.; p.cp_d = 1003.; p.R_v = 461.; p.p0 = 1.e5;
  unsigned long long *bits = (unsigned long long *)workspace;
  double *ws = (double *)workspace + 16;
  hipLaunchKernelGGL(k_kessler_init_min, dim3(1), dim3(64), 0, st, bits); MW_LAUNCH_CHECK();
  long long nb = ((long long)nz * ncol + 255) / 256;
  if (nb > 4096) nb = 4096;
  // z chunks of the rainsplit == 1 path: enough (column, chunk) threads to fill the chip, at least 4 levels per chunk
  int chunk = nz;
  for (int c : {25, 20, 16, 12, 10, 8, 5, 4}) if (c < nz) { chunk = c; if (((ncol + 63) / 64) * ((nz + c - 1) / c) >= 16384) break; }
  const int nchunks = (nz + chunk - 1) / chunk;
  double *flux_top = ws + 5ll * nz * ncol;                    // (nchunks-1, ncol)
  hipLaunchKernelGGL(k_kessler_prep, dim3((unsigned)nb), dim3(256), 0, st, p, rho_v, rho_r, rho_d, temp, ws, flux_top, chunk, bits); MW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_kessler_chunks, dim3((unsigned)((ncol + 255) / 256), (unsigned)nchunks), dim3(256), 0, st, p, rho_v, rho_c, rho_r,
                     rho_d, temp, precl, bits, ws, flux_top, chunk); MW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_kessler_column, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, st, p, rho_v, rho_c, rho_r, rho_d, temp,
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

} // extern "C"

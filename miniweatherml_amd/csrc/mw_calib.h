// =====================================================================================================
// mw_calib.h -- calibration kernels (no reference counterpart; SURVEY.md 8(d): "fp64 vector 78.6 TFLOP/s ... absent from the
// local guide -- calibrate with an FMA microbenchmark").  Included by mw_dycore.hip behind mw_march.h.
//
//   k_calib_fma64       independent v_fma_f64 chains, W wavefronts per SIMD: the SUSTAINED fp64 issue rate of this part under its own
//                       power management (wave-instructions per second) and the shader clock it ran at (cycle counter / real-time
//                       counter) -- the measured ceiling that "fp64-VALU fraction" statements are made against.
//   k_calib_stage_arith the ARITHMETIC of one RK stage and nothing else: per cell 24 weno5_edges_fast + 3 riemann_primary + the 15
//                       passive fluxes, on register windows fed from a table that stays in L2 (no HBM traffic, one store per thread
//                       at the end), with the register budget and occupancy of k_xz_state (256 threads, 2 workgroups per CU).
//                       Its time for N cells is the floor of a stage of N cells for ANY schedule of this arithmetic on this chip.
//   k_spin / k_scale    test aids: a kernel that occupies a stream for a given time (delay fuzz of the exchange tests) and
//                       buf *= f (the self-loop transport's sum over identical blocks).
// =====================================================================================================
#pragma once

namespace mw {

// 8 independent chains x 8 links per loop trip = 64 v_fma_f64 per trip and lane; asm so that nothing is folded or re-associated.
__global__ __launch_bounds__(256) void k_calib_fma64(long long trips, double seed, double *__restrict__ sink, long long *__restrict__ clocks) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 0.999999, c = 1.0e-6;
  const long long t0 = clock64(), r0 = wall_clock64();
  for (long long i = 0; i < trips; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                   "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  const long long t1 = clock64(), r1 = wall_clock64();
  const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 1.2345e300) sink[0] = s;                             // (never true: keeps the chains alive)
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

// riemann_primary<1> (mw_march.h) without the cold branch to the out-of-line pow (|(rho theta)'| > 5 % of the background: never taken
// on the table's data): three call sites of a real function in one loop body cost this kernel 41-84 spilled VGPRs, the production
// kernels have at most two.  The executed instructions are the same.
__device__ __forceinline__ FaceState riemann_floor(double rL, double rR, double uL, double uR, double eTL, double eTR, double hyt, double p0,
                                                   double ihyt, double &f_nrm, double &f_T) {
#pragma clang fp contract(fast)
  const double cs = 350;
  const double mL = uL * rL, mR = uR * rR;
  const double dL = eTL * ihyt, dR = eTR * ihyt;
  double sL, sR;
  pressure_series_pair(dL, dR, sL, sR);
  const double p_L = p0 + p0 * (sL * dL), p_R = p0 + p0 * (sR * dR);
  const double w1 = 0.5 * (p_R - cs * mR), w2 = 0.5 * (p_L + cs * mL);
  const double p_upw = w1 + w2;
  FaceState fs;
  fs.m_upw = (w2 - w1) * (1.0 / 350.0);
  fs.ind = (__dadd_rn(mL, mR) > 0) ? 0 : 1;
  const double r_upw = fs.ind ? rR : rL, u_upw = fs.ind ? uR : uL;
  f_nrm = fs.m_upw * u_upw + p_upw;
  f_T = fs.m_upw * ((fs.ind ? eTR : eTL) + hyt) * fast_rcp(r_upw);
  return fs;
}

// One stage's arithmetic per cell.  tab: (nlev, 8, 64) doubles -- level k of a 64-lane row of the eight reconstruction variables
// (rho', u, v, w, (rho theta)', three tracers); every wave reads the same rows (L2 hits).  bg: hyr, hyt, p0, 1/hyt of the level.
// NV = 8: all three tracers are reconstructed (24 reconstructions per cell); NV = 6: cloud and rain are exactly zero and take the production
// kernels' zero short-cut (mw_march.h, MW_ZERO_SKIP) -- 18 reconstructions per cell: the floor of a stage on a cloud-free state.
template <int NV>
__global__ __launch_bounds__(256, 2) void k_calib_stage_arith(const double *__restrict__ tab, int nlev, int levels, double hyr, double hyt,
                                                              double p0, double ihyt, double *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const double *col = tab + lane;
  double w[NV][5], nxt[NV], ct[NV];
  int kt = (int)((blockIdx.x * 7u + (threadIdx.x >> 6)) % (unsigned)nlev);     // (every wave starts somewhere else in the table)
#pragma unroll
  for (int v = 0; v < NV; v++) {
    ct[v] = 0;
#pragma unroll
    for (int s = 0; s < 5; s++) w[v][s] = col[((long long)((kt + s) % nlev) * 8 + v) * 64];
  }
  kt = (kt + 5) % nlev;
  double acc = 0;
#pragma unroll
  for (int v = 0; v < NV; v++) landed(w[v]);
  for (int k = 0; k < levels; k++) {
#pragma unroll
    for (int v = 0; v < NV; v++) nxt[v] = col[((long long)kt * 8 + v) * 64];
    kt = (kt + 1 == nlev) ? 0 : kt + 1;
    // z direction: the window as it is;  "x" and "y": the same five values in two other orders (nothing is shared between the three
    // reconstructions of a variable: their first differences all differ).  One direction at a time, its Riemann solve right behind it
    // (the production kernels' order; all 48 edge values at once would not fit the register file).
    double f = 0;
    {   // z face: lower cell's top edge (carried) against this cell's bottom edge
      double be[NV], te[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { weno5_edges_fast(w[v][0], w[v][1], w[v][2], w[v][3], w[v][4], be[v], te[v]); if (v & 1) MW_SCHED_FENCE(); }
      double fn, fT;
      const FaceState fs = riemann_floor(ct[idR] + hyr, be[idR] + hyr, ct[idW], be[idW], ct[idT], be[idT], hyt, p0, ihyt, fn, fT);
      f += fs.m_upw + fn + fT;
      f += fs.m_upw * (fs.ind ? be[idU] : ct[idU]) + fs.m_upw * (fs.ind ? be[idV] : ct[idV]);
#pragma unroll
      for (int t = 5; t < NV; t++) f += fs.m_upw * (fs.ind ? be[t] : ct[t]);
#pragma unroll
      for (int v = 0; v < NV; v++) ct[v] = te[v];
    }
    MW_SCHED_FENCE();
    {   // "x" face
      double we[NV], ee[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { weno5_edges_fast(w[v][1], w[v][0], w[v][2], w[v][4], w[v][3], we[v], ee[v]); if (v & 1) MW_SCHED_FENCE(); }
      double fn, fT;
      const FaceState fs = riemann_floor(ee[idR] + hyr, we[idR] + hyr, ee[idU], we[idU], ee[idT], we[idT], hyt, p0, ihyt, fn, fT);
      f += fs.m_upw + fn + fT;
      f += fs.m_upw * (fs.ind ? we[idV] : ee[idV]) + fs.m_upw * (fs.ind ? we[idW] : ee[idW]);
#pragma unroll
      for (int t = 5; t < NV; t++) f += fs.m_upw * (fs.ind ? we[t] : ee[t]);
    }
    MW_SCHED_FENCE();
    {   // "y" face
      double se[NV], ne[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { weno5_edges_fast(w[v][3], w[v][1], w[v][2], w[v][0], w[v][4], se[v], ne[v]); if (v & 1) MW_SCHED_FENCE(); }
      double fn, fT;
      const FaceState fs = riemann_floor(ne[idR] + hyr, se[idR] + hyr, ne[idV], se[idV], ne[idT], se[idT], hyt, p0, ihyt, fn, fT);
      f += fs.m_upw + fn + fT;
      f += fs.m_upw * (fs.ind ? se[idU] : ne[idU]) + fs.m_upw * (fs.ind ? se[idW] : ne[idW]);
#pragma unroll
      for (int t = 5; t < NV; t++) f += fs.m_upw * (fs.ind ? se[t] : ne[t]);
    }
    acc += f;
    landed(nxt);
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
      for (int s = 0; s < 4; s++) w[v][s] = w[v][s + 1];
      w[v][4] = nxt[v];
    }
  }
  sink[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_spin(long long ticks) {                        // wall_clock64: the constant 100 MHz counter
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ __launch_bounds__(256) void k_scale(double *__restrict__ buf, long long n, double f) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) buf[t] *= f;
}

int launch_spin(long long usec, hipStream_t st) {
  if (usec <= 0) return 0;
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, usec * 100);
  MW_LAUNCH_CHECK();
  return 0;
}
int launch_scale(double *buf, long long n, double f, hipStream_t st) {
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, n, f);
  MW_LAUNCH_CHECK();
  return 0;
}

} // namespace mw

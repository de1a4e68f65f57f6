// =====================================================================================================
// mw_dycore.hip -- MI355X (gfx950) implementation of Dynamics_Euler_Stratified_WenoFV::time_step
// (reference: model/modules/dynamics_euler_stratified_wenofv.h:81-552,1891-2015) behind include/mw_cdna4.h.
//
// Data layout in HBM:
//   * two prognostic slabs S0 (q^n) and S1 (q*), each (V, nz+2*HZ, ny+2*HY, (nx+2*HX)*nens) fp64, x(+ens) fastest,
//     V = 5 + T, HX = HY = 3 (HY = 0 in 2-D), HZ = 2.  They hold the *reconstruction variables*
//     (rho', u, v, w, (rho theta)', q_t) -- i.e. the reference's `state`/`tracers` AFTER its in-place divide by
//     density (:248-255); the conserved value is recovered as u*rho exactly like the reference's re-multiply
//     (:477-484), so no information or rounding step is lost or added.
//   * six face-flux arrays in the reference's public layout (:1671-1676), state and tracer parts contiguous.
//   * no `limits` arrays (6*V*N doubles in the reference, :260-265): edge values live in registers only.
//
// Per RK stage:   halo fill (3-cell, replaces halo_exchange+edge_exchange)  ->  k_flux (D6+D9 fused)
//                 ->  k_fct (D10)  ->  k_update (D11 + D12 + the next stage's D2, or D13 on the last stage)
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include "mw_weno.h"
#include "mw_glibc_pow.h"
#include <vector>
#include <map>
#include <set>
#include <tuple>
#include <mutex>
#include <cmath>
#include <cstring>
#include <random>
#include <algorithm>

// Every kernel launch of this file goes through MW_KLAUNCH: besides launching it notes the kernel's host function in a process-wide
// registry, so that a test session can ask which INSTANTIATIONS of the dispatcher's kernels its oracle comparisons really exercised
// (mw_debug_launched_kernels; tests/conftest.py checks them against the list of everything compiled into this object).  Host-side only:
// one uncontended mutex and a set insert per launch, nothing in the kernels.
namespace mw {
static std::mutex g_launch_mu;
static std::set<const void *> g_launched;
static inline void note_launch(const void *fn) { std::lock_guard<std::mutex> lk(g_launch_mu); g_launched.insert(fn); }
}
#define MW_KLAUNCH(kern, ...) do { mw::note_launch((const void *)&kern); hipLaunchKernelGGL(kern, __VA_ARGS__); } while (0)

namespace mw {

enum { idR = 0, idU = 1, idV = 2, idW = 3, idT = 4 };
static constexpr int HXc = 3;     // x/y halo: 2 for the stencil + 1 so that the neighbour's edge value is rebuilt locally
static constexpr int HZc = 2;     // z halo: z faces at the domain boundary use the edge-value BC rule, not ghost cells

// A row / level / variable stride in elements.  It fits 32 bits on any handle that fits the GPU (one variable of a slab with 2^31 doubles is 16 GB,
// and a handle keeps 4 slabs of >= 6 variables; strides_fit() refuses anything else at create), and every use is a
// product with a 32-bit index: held as an int that converts to long long, `(long long)idx * stride` is a 32 x 32 -> 64 multiply (s_mul_i32 +
// s_mul_hi_i32) instead of the 64 x 64 one (7 scalar instructions) -- the marching kernels derive a dozen such offsets from the level index in
// every iteration (no SGPRs to keep them), and at two waves per SIMD a wave's scalar instructions delay its own vector ones.  No int arithmetic can
// overflow through it: the only way out is the conversion.
struct Stride32 {
  int v;
  __host__ __device__ __forceinline__ operator long long() const { return (long long)v; }
  __host__ __device__ __forceinline__ Stride32 &operator=(long long x) { v = (int)x; return *this; }
};

struct DyP {                      // kernel parameter block (by value)
  int nz, ny, nx, nens, nt, V;
  int HX, HY, HZ;
  int NXE;                        // (nx+2HX)*nens
  Stride32 sJ, sK, sV;            // row / level / variable strides of the prognostic slabs
  Stride32 nC;                    // nz*ny*nx*nens
  Stride32 fxJ, fxK, fxV, fyJ, fyK, fyV, fzJ, fzK, fzV;   // flux strides
  int sim2d, bc_x, bc_y, bc_z, px, py, nproc_x, nproc_y;
  int v0;                         // halo/pack kernels: index of the first variable of the group being processed
  int cst, ce;                    // coupler-side stride / member offset (1, 0; member-major mode: nens, member) -- see cpl() in mw_march.h
  int wrap_x, wrap_y;             // production path, periodic direction owned by one rank: the marching kernels wrap their x / row index
                                  // instead of reading halo cells, and that halo is not filled
  int enable_gravity, use_immersed, idWV;
  int zero_skip;                  // marching kernels: skip the reconstructions of a tracer that is exactly zero over a wavefront's stencil (mw_march.h)
  // zero-row map of the current RK stage (mw_march.h: k_zero_rows), nullptr = none: one word per (level, row), bit v = "tracer v may be
  // non-zero in what iterations k-3 .. k of the row's marching wave touch"; word of (k, j) at [k * zq_ld + j + HY]
  const unsigned *zq;
  const unsigned *zqk;            // ... the converting y launch: the rows of the slab it writes that hold zeros already
  const unsigned *zqp, *zqc;      // ... "the row an iteration stores to holds zeros already": the previous sub-cycle's map of this stage / the coupler's rows (mw_march.h)
  int zq_ld;
  // parked column increments (mw_nudge_to_column_deferred): inc[(l * nz + k) * nens + e] for l = density_dry, uvel, vvel, temp, water_vapor, added to
  // the coupler's values while the converting y launch loads them; nullptr = none
  const double *pinc;
  unsigned pos_mask, mass_mask;
  double dx, dy, dz, rdx, rdy, rdz, C0, gamma, grav, fcor, R_d, R_v;
  const double *hyc, *hytc, *hye, *hyte;       // device (nz,nens) / (nz+1,nens)
  const double *p0c, *p0e, *ihytc, *ihyte;     // C0*hyt^gamma and 1/hyt at cells / edges (fast pressure path)
  const double *imm;                           // device (nz,ny,nx,nens)
  const double *hypk;                          // the eight profile values of level k packed as rows of 8: (hyc, hytc, p0c, ihytc, hye, hyte,
                                               // p0e, ihyte)[(k*nens+e)*8 + f], nz+1 rows: one pointer instead of eight in the hot kernels
  double bn[11];                               // binomial series coefficients C(gamma, n), n = 0..10
  int bn_default;                              // bn[] equals the literal table for gamma = 1003/716 bit for bit (the usual case)
  int an_default;                              // likewise C(1/gamma, n) of the conversion's inverse series (mw_march.h)
};

struct CouplerPtrs {
  double *rho_d, *u, *v, *w, *temp;
  double *tr[MW_MAX_TRACERS];
};

// Compile-time configuration of the marching kernels (mw_march.h).  The run-time switches of DyP that are wave-uniform and fixed
// for a whole run cost SGPRs (the marching kernels have none to spare: every SGPR spilled to a VGPR lane comes back as a
// v_readlane, a VALU instruction) and selects (v_cndmask pairs per double).  K = 0 keeps every switch at run time (any
// configuration).  K = 1 / 2 are the shipped experiments' configurations with the switches folded:
//   both: nens == 1 (or one member of a member-major handle), 3-D, periodic x and y (any rank count: the index wrap stays a
//         run-time switch), wall in z, no Coriolis term (latitude is forced to 0 at init, :1249), the default gamma (series
//         coefficients as literals), every tracer positive and mass-adding with water vapour first (idWV == 0);
//   K = 1 (supercell_example, supercell_kessler_surrogate, community_benchmark): gravity on, no immersed boundaries, the three
//         Kessler tracers;
//   K = 2 (simple_city): immersed boundaries, gravity off, water vapour only.
// marching_config() in the host part decides; anything else runs K = 0.
template <int K> struct Cf {
  static constexpr bool spec = (K != 0);
  static __device__ __forceinline__ bool x_periodic(const DyP &p) { return spec || p.bc_x == MW_BC_PERIODIC; }
  static __device__ __forceinline__ bool y_periodic(const DyP &p) { return spec || p.bc_y == MW_BC_PERIODIC; }
  static __device__ __forceinline__ bool z_wall(const DyP &p) { return spec || p.bc_z == MW_BC_WALL; }
  static __device__ __forceinline__ bool sim2d(const DyP &p) { return !spec && p.sim2d; }
  static __device__ __forceinline__ bool immersed(const DyP &p) { return K == 2 || (!spec && p.use_immersed); }
  static __device__ __forceinline__ bool gravity(const DyP &p) { return K == 1 || (!spec && p.enable_gravity); }
  static __device__ __forceinline__ bool coriolis(const DyP &p) { return !spec; }
  static __device__ __forceinline__ bool bn_default(const DyP &p) { return spec || p.bn_default; }
  static __device__ __forceinline__ bool an_default(const DyP &p) { return spec || p.an_default; }
  static __device__ __forceinline__ bool positive(const DyP &p, int t) { return spec || ((p.pos_mask >> t) & 1u); }
  static __device__ __forceinline__ bool adds_mass(const DyP &p, int t) { return spec || ((p.mass_mask >> t) & 1u); }
  static __device__ __forceinline__ bool is_wv(const DyP &p, int t) { return spec ? (t == 0) : (t == p.idWV); }
  static __device__ __forceinline__ int ntr(const DyP &p) { return K == 1 ? 3 : K == 2 ? 1 : p.nt; }   // K = 1: the three Kessler tracers; K = 2: water vapour
};

// -----------------------------------------------------------------------------------------------------
// pow(x, gamma): strict = device libm pow; fast = same for now (kept separate so it can be specialised)
// -----------------------------------------------------------------------------------------------------
// pow of the kernels that keep the reference's operation order (general path: strict and fast arithmetic; init; D1 / D13 passes):
// the bits of the host's glibc (mw_glibc_pow.h), so that the strict path equals the CPU oracle bit for bit.  Arguments outside the
// restated main path -- nothing the dycore produces -- take the device library's pow.
__device__ __forceinline__ double pow_ref(double x, double y) {
  double r;
  if (__builtin_expect(glibc_pow_main(x, y, &r), 1)) return r;
  return pow(x, y);
}
__device__ __forceinline__ double exp_ref(double x) {       // likewise exp (the thermal initial state's saturation vapour pressure, :1139)
  double r;
  if (__builtin_expect(glibc_exp_main(x, &r), 1)) return r;
  return exp(x);
}
__device__ __forceinline__ double cos_ref(double x) {       // likewise cos (the cosine bells of the initial states, :1131, perturb_temperature.h:63)
  double r;
  if (__builtin_expect(glibc_cos_main(x, &r), 1)) return r;
  return cos(x);
}
template <bool STRICT> __device__ __forceinline__ double pow_gamma(double x, double g) { return pow_ref(x, g); }

// p = C0 (hyt + e)^gamma for the fast path.  The Riemann solver needs two of these per face (6 per cell and stage,
// :401,:426,:457); the device-libm pow costs ~230 fp64-VALU instructions.  Writing (hyt + e)^gamma =
// hyt^gamma (1 + delta)^gamma with delta = e/hyt (|delta| is a few per cent: e is the reconstructed PERTURBATION of
// rho*theta) turns it into p0(k) * sum_n C(gamma,n) delta^n: 10 FMAs; truncation |C(gamma,11)| 0.05^11 ~ 1e-17 for
// |delta| <= 0.05.  Larger perturbations take the generic pow (per-lane branch).
// out of line on purpose: the libm pow body (~230 instructions, ~60 VGPRs) would otherwise be inlined twice per Riemann solve
// into kernels that sit at the register limit; it only runs for |(rho theta)'| > 5 % of the hydrostatic value.
__device__ __attribute__((noinline)) double pressure_pow(double C0, double x, double gamma) { return C0 * pow(x, gamma); }

// C(gamma, n), n = 1..10, for the default gamma = cp_d/(cp_d - R_d) = 1003/716 (the long-double recurrence of fill_params, as
// hex literals).  Literal operands are materialised by scalar moves where they are used; the same numbers read from the
// parameter block stay resident in 22 SGPRs for the whole kernel -- and the marching kernels already spill SGPRs to VGPR lanes.
__device__ __forceinline__ double pressure_series_default(double dl) {
#pragma clang fp contract(fast)
  double acc = 0x1.d587239f51368p-10;
  acc = acc * dl + -0x1.34ef19ee96d45p-9;
  acc = acc * dl + 0x1.a553bdf108378p-9;
  acc = acc * dl + -0x1.2cfe340a81e1p-8;
  acc = acc * dl + 0x1.ca1dc2cec496fp-8;
  acc = acc * dl + -0x1.7dda38e0cc64cp-7;
  acc = acc * dl + 0x1.6f48bfb7e5329p-6;
  acc = acc * dl + -0x1.cb58863e4dd29p-5;
  acc = acc * dl + 0x1.1f7e1e502b562p-2;
  acc = acc * dl + 0x1.669d5185016e2p+0;
  return acc;
}
static const double BN_DEFAULT[11] = {1.0, 0x1.669d5185016e2p+0, 0x1.1f7e1e502b562p-2, -0x1.cb58863e4dd29p-5, 0x1.6f48bfb7e5329p-6,
                                      -0x1.7dda38e0cc64cp-7, 0x1.ca1dc2cec496fp-8, -0x1.2cfe340a81e1p-8, 0x1.a553bdf108378p-9,
                                      -0x1.34ef19ee96d45p-9, 0x1.d587239f51368p-10};

template <int K = 0>
__device__ __forceinline__ double pressure_fast(const DyP &p, double e, double hyt, double p0, double ihyt) {
#pragma clang fp contract(fast)
  double dl = e * ihyt;
  if (fabs(dl) <= 0.05 && Cf<K>::bn_default(p)) return p0 + p0 * (pressure_series_default(dl) * dl);
  return pressure_pow(p.C0, hyt + e, p.gamma);                  // large perturbation, or a non-default gamma
}
// The two sides of a face at once (the Riemann solver needs both, :401): the same two Horner chains as pressure_series_default,
// as three-address v_fma_f64 interleaved in ONE asm statement.  Left to the compiler, the chain of a polynomial whose coefficients
// it keeps in registers becomes v_mov_b64 (coefficient -> accumulator) + v_fmac_f64 (two-address) per step -- nine extra VALU
// instructions per evaluation, four evaluations per level in k_xz_state (seen in round 2's gfx950 code, tools/isa_histogram.py) --
// inside one divergent block per side.  Here: two independent dependency chains in one block, no moves.  (The coefficients are
// "v" operands: twenty VGPRs for the whole kernel.  The single evaluations of D1 / D13 keep the compiler's form: k_tracers_fused
// <3, 1> has no registers for them.)
__device__ __forceinline__ void pressure_series_pair(double dlL, double dlR, double &sL, double &sR) {
  double aL, aR;
  asm("v_fma_f64 %0, %4, %2, %5\n\tv_fma_f64 %1, %4, %3, %5\n\t"
      "v_fma_f64 %0, %0, %2, %6\n\tv_fma_f64 %1, %1, %3, %6\n\t"
      "v_fma_f64 %0, %0, %2, %7\n\tv_fma_f64 %1, %1, %3, %7\n\t"
      "v_fma_f64 %0, %0, %2, %8\n\tv_fma_f64 %1, %1, %3, %8\n\t"
      "v_fma_f64 %0, %0, %2, %9\n\tv_fma_f64 %1, %1, %3, %9\n\t"
      "v_fma_f64 %0, %0, %2, %10\n\tv_fma_f64 %1, %1, %3, %10\n\t"
      "v_fma_f64 %0, %0, %2, %11\n\tv_fma_f64 %1, %1, %3, %11\n\t"
      "v_fma_f64 %0, %0, %2, %12\n\tv_fma_f64 %1, %1, %3, %12\n\t"
      "v_fma_f64 %0, %0, %2, %13\n\tv_fma_f64 %1, %1, %3, %13"
      : "=&v"(aL), "=&v"(aR)
      : "v"(dlL), "v"(dlR), "v"(0x1.d587239f51368p-10), "v"(-0x1.34ef19ee96d45p-9), "v"(0x1.a553bdf108378p-9), "v"(-0x1.2cfe340a81e1p-8),
        "v"(0x1.ca1dc2cec496fp-8), "v"(-0x1.7dda38e0cc64cp-7), "v"(0x1.6f48bfb7e5329p-6), "v"(-0x1.cb58863e4dd29p-5),
        "v"(0x1.1f7e1e502b562p-2), "v"(0x1.669d5185016e2p+0));
  sL = aL; sR = aR;
}
template <int K = 0>
__device__ __forceinline__ void pressure_fast_pair(const DyP &p, double eL, double eR, double hyt, double p0, double ihyt, double &pL, double &pR) {
#pragma clang fp contract(fast)
  const double dL = eL * ihyt, dR = eR * ihyt;
  if (__builtin_expect(fabs(dL) <= 0.05 && fabs(dR) <= 0.05 && Cf<K>::bn_default(p), 1)) {
    double sL, sR;
    pressure_series_pair(dL, dR, sL, sR);
    pL = p0 + p0 * (sL * dL);
    pR = p0 + p0 * (sR * dR);
  } else {                                                      // a large perturbation on either side, or a non-default gamma
    pL = pressure_fast<K>(p, eL, hyt, p0, ihyt);
    pR = pressure_fast<K>(p, eR, hyt, p0, ihyt);
  }
}

// -----------------------------------------------------------------------------------------------------
// D1  convert_coupler_to_dynamics (:1955-2015) fused with the stage-1 divide D2 (:248-255)
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coupler_to_state(DyP p, CouplerPtrs c, double *__restrict__ S) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  int e = ie % p.nens;
  long long ci = ((long long)k * p.ny + j) * NXI + ie;
  double rho_d = c.rho_d[ci], u = c.u[ci], v = c.v[ci], w = c.w[ci], temp = c.temp[ci];
  double rho_v = c.tr[p.idWV][ci];
  double press = rho_d * p.R_d * temp + rho_v * p.R_v * temp;
  double rho = rho_d;
  for (int tr = 0; tr < p.nt; tr++) if ((p.mass_mask >> tr) & 1u) rho += c.tr[tr][ci];
  double theta = pow_ref(press / p.C0, 1.0 / p.gamma) / rho;
  double hyc = p.hyc[k * p.nens + e], hytc = p.hytc[k * p.nens + e];
  double *s = S + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
  double rp = rho - hyc;                                  // state(idR)
  double den = rp + hyc;                                  // what D2 divides by (:249)
  s[idR * p.sV] = rp;
  s[idU * p.sV] = (rho * u) / den;
  s[idV * p.sV] = (rho * v) / den;
  s[idW * p.sV] = (rho * w) / den;
  s[idT * p.sV] = rho * theta - hytc;
  for (int tr = 0; tr < p.nt; tr++) s[(5 + tr) * p.sV] = c.tr[tr][ci] / den;
}

// -----------------------------------------------------------------------------------------------------
// Halo fill, single rank in a direction == periodic wrap onto itself (coupler.h:169-179 self neighbour),
// plus the boundary conditions of halo_exchange (:752-825).  One thread per halo cell and variable.
// Regions: 0 = x halos (k,j interior), 1 = y halos (k,i interior), 2 = z halos (j,i interior); corners are
// never read.  `do_x`/`do_y` = 0 when that direction's halos come from a neighbour.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void halo_x_body(const DyP &p, double *__restrict__ S, long long t) {
  // threads: (v, k, j, h in [0,2HX), e)
  int H2 = 2 * p.HX;
  long long n = (long long)p.V * p.nz * p.ny * H2 * p.nens;
  if (t >= n) return;
  int e = (int)(t % p.nens); t /= p.nens;
  int h = (int)(t % H2); t /= H2;
  int j = (int)(t % p.ny); t /= p.ny;
  int k = (int)(t % p.nz); int v = (int)(t / p.nz);
  double *row = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + e;
  int lo = h < p.HX;
  int ih = lo ? h : p.nx + h;                       // halo cell index in [0,HX) or [nx+HX, nx+2HX)
  double val;
  if (p.bc_x == MW_BC_PERIODIC) {
    int src = lo ? ih + p.nx : ih - p.nx;
    val = row[(long long)src * p.nens];
  } else {                                          // :782-803 (both sides, two independent ifs: `px == 0`, `px == nproc_x-1`)
    if (lo ? (p.px != 0) : (p.px != p.nproc_x - 1)) return;   // not a domain edge: this halo holds the neighbour's strip
    if (v + p.v0 == idU && p.bc_x == MW_BC_WALL) val = 0;
    else val = row[(long long)(lo ? p.HX : p.HX + p.nx - 1) * p.nens];
  }
  row[(long long)ih * p.nens] = val;
}
__device__ __forceinline__ void halo_y_body(const DyP &p, double *__restrict__ S, long long t) {
  // threads: (v, k, h in [0,2HY), ie interior)
  int H2 = 2 * p.HY, NXI = p.nx * p.nens;
  long long n = (long long)p.V * p.nz * H2 * NXI;
  if (t >= n) return;
  int ie = (int)(t % NXI); t /= NXI;
  int h = (int)(t % H2); t /= H2;
  int k = (int)(t % p.nz); int v = (int)(t / p.nz);
  double *col = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;
  int lo = h < p.HY;
  int jh = lo ? h : p.ny + h;
  double val;
  if (p.bc_y == MW_BC_PERIODIC) {
    int src = lo ? jh + p.ny : jh - p.ny;
    val = col[(long long)src * p.sJ];
  } else {                                          // :804-825
    if (lo ? (p.py != 0) : (p.py != p.nproc_y - 1)) return;
    if (v + p.v0 == idV && p.bc_y == MW_BC_WALL) val = 0;
    else val = col[(long long)(lo ? p.HY : p.HY + p.ny - 1) * p.sJ];
  }
  col[(long long)jh * p.sJ] = val;
}
__device__ __forceinline__ void halo_z_body(const DyP &p, double *__restrict__ S, long long t) {
  // threads: (v, h in [0,2HZ), j, ie interior)   :752-781
  int H2 = 2 * p.HZ, NXI = p.nx * p.nens;
  long long n = (long long)p.V * H2 * p.ny * NXI;
  if (t >= n) return;
  int ie = (int)(t % NXI); t /= NXI;
  int j = (int)(t % p.ny); t /= p.ny;
  int h = (int)(t % H2); int v = (int)(t / H2);
  double *col = S + (long long)v * p.sV + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
  int lo = h < p.HZ;
  int kh = lo ? h : p.nz + h;
  double val;
  if (p.bc_z == MW_BC_PERIODIC) val = col[(long long)(lo ? kh + p.nz : kh - p.nz) * p.sK];     // :752-763
  else if (v + p.v0 == idW && p.bc_z == MW_BC_WALL) val = 0;
  else val = col[(long long)(lo ? p.HZ : p.HZ + p.nz - 1) * p.sK];
  col[(long long)kh * p.sK] = val;
}

// All three regions in one launch (they are independent: each writes its own halo cells from interior cells, and the corners
// are never read): blocks [0, nbx) do x, [nbx, nbx+nby) do y, the rest z.  Used when no direction needs a neighbour exchange.
__global__ __launch_bounds__(256) void k_halo_xyz(DyP p, double *__restrict__ S, unsigned nbx, unsigned nby) {
  const unsigned b = blockIdx.x;
  if (b < nbx) halo_x_body(p, S, (long long)b * 256 + threadIdx.x);
  else if (b < nbx + nby) halo_y_body(p, S, (long long)(b - nbx) * 256 + threadIdx.x);
  else halo_z_body(p, S, (long long)(b - nbx - nby) * 256 + threadIdx.x);
}

// Pack / unpack for a neighbour exchange (3-cell halos of all V variables; interior rows only, like :606-631,:725-747)
// W/E buffers (V,nz,ny,HX,nens), S/N buffers (V,nz,HY,nx,nens).
__device__ __forceinline__ void pack_x_body(const DyP &p, const double *__restrict__ S, double *__restrict__ bW, double *__restrict__ bE, long long t) {
  long long n = (long long)p.V * p.nz * p.ny * p.HX * p.nens;
  if (t >= n) return;
  long long r = t;
  int e = (int)(r % p.nens); r /= p.nens;
  int h = (int)(r % p.HX); r /= p.HX;
  int j = (int)(r % p.ny); r /= p.ny;
  int k = (int)(r % p.nz); int v = (int)(r / p.nz);
  const double *row = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + e;
  bW[t] = row[(long long)(p.HX + h) * p.nens];              // my first HX interior cells -> west neighbour's east halo
  bE[t] = row[(long long)(p.nx + h) * p.nens];              // my last  HX interior cells -> east neighbour's west halo
}
__device__ __forceinline__ void unpack_x_body(const DyP &p, double *__restrict__ S, const double *__restrict__ bW, const double *__restrict__ bE, long long t) {
  long long n = (long long)p.V * p.nz * p.ny * p.HX * p.nens;
  if (t >= n) return;
  long long r = t;
  int e = (int)(r % p.nens); r /= p.nens;
  int h = (int)(r % p.HX); r /= p.HX;
  int j = (int)(r % p.ny); r /= p.ny;
  int k = (int)(r % p.nz); int v = (int)(r / p.nz);
  double *row = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + e;
  row[(long long)h * p.nens]                 = bW[t];       // received from west
  row[(long long)(p.nx + p.HX + h) * p.nens] = bE[t];       // received from east
}
__device__ __forceinline__ void pack_y_body(const DyP &p, const double *__restrict__ S, double *__restrict__ bS, double *__restrict__ bN, long long t) {
  int NXI = p.nx * p.nens;
  long long n = (long long)p.V * p.nz * p.HY * NXI;
  if (t >= n) return;
  long long r = t;
  int ie = (int)(r % NXI); r /= NXI;
  int h = (int)(r % p.HY); r /= p.HY;
  int k = (int)(r % p.nz); int v = (int)(r / p.nz);
  const double *col = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;
  bS[t] = col[(long long)(p.HY + h) * p.sJ];
  bN[t] = col[(long long)(p.ny + h) * p.sJ];
}
__device__ __forceinline__ void unpack_y_body(const DyP &p, double *__restrict__ S, const double *__restrict__ bS, const double *__restrict__ bN, long long t) {
  int NXI = p.nx * p.nens;
  long long n = (long long)p.V * p.nz * p.HY * NXI;
  if (t >= n) return;
  long long r = t;
  int ie = (int)(r % NXI); r /= NXI;
  int h = (int)(r % p.HY); r /= p.HY;
  int k = (int)(r % p.nz); int v = (int)(r / p.nz);
  double *col = S + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;
  col[(long long)h * p.sJ]                 = bS[t];
  col[(long long)(p.ny + p.HY + h) * p.sJ] = bN[t];
}

// Both directions in ONE launch (round 5; blocks [0, nbx) pack / unpack the W / E strips, the rest the S / N strips -- nbx or the rest
// may be 0): an exchange is a chain pack -> transfer -> unpack on a stream that shares the chip with the big stencil kernels, where every
// launch waits for a workgroup slot; two launches per exchange instead of four shorten the chain (tools/exchange_serial_probe.py).
__global__ __launch_bounds__(256) void k_pack_xy(DyP p, const double *__restrict__ S, double *__restrict__ bW, double *__restrict__ bE,
                                                 double *__restrict__ bS, double *__restrict__ bN, unsigned nbx) {
  if (blockIdx.x < nbx) pack_x_body(p, S, bW, bE, (long long)blockIdx.x * 256 + threadIdx.x);
  else pack_y_body(p, S, bS, bN, (long long)(blockIdx.x - nbx) * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_unpack_xy(DyP p, double *__restrict__ S, const double *__restrict__ bW, const double *__restrict__ bE,
                                                   const double *__restrict__ bS, const double *__restrict__ bN, unsigned nbx) {
  if (blockIdx.x < nbx) unpack_x_body(p, S, bW, bE, (long long)blockIdx.x * 256 + threadIdx.x);
  else unpack_y_body(p, S, bS, bN, (long long)(blockIdx.x - nbx) * 256 + threadIdx.x);
}

// -----------------------------------------------------------------------------------------------------
// Flux stencil: WENO reconstruction (D6, :271-388) + edge BCs (D8, :1008-1081) + Riemann solver (D9, :395-474)
// fused; one thread per face triple (x-, y-, z- lower faces of cell (k,j,i,e), plus the top/north/east rim).
// The two sides of a face are described by (source cell, which edge); passive variables (transverse momenta,
// tracers) are reconstructed on the upwind side only.
// -----------------------------------------------------------------------------------------------------
template <bool STRICT, int ORD = 5>
__device__ __forceinline__ double edge_value(const double *__restrict__ q, long long st, int right) {
  double l, r;
  if (ORD == 7 || ORD == 9) {
    constexpr int h = (ORD - 1) / 2;
    double s[ORD];
#pragma unroll
    for (int m = 0; m < ORD; m++) s[m] = q[(long long)(m - h) * st];
    if (STRICT) weno79_edges_strict<ORD>(s, l, r); else weno79_edges_fast<ORD>(s, l, r);
  }
  else if (ORD == 3) { if (STRICT) weno3_edges_strict(q[-st], q[0], q[st], l, r); else weno3_edges_fast(q[-st], q[0], q[st], l, r); }
  else if (STRICT) weno5_edges_strict(q[-2 * st], q[-st], q[0], q[st], q[2 * st], l, r);
  else             weno5_edges_fast  (q[-2 * st], q[-st], q[0], q[st], q[2 * st], l, r);
  return right ? r : l;
}

// Background values a face side adds to its reconstructed perturbations: hyr (density), hyt (rho theta), p0 = C0 hyt^gamma, 1/hyt.
// Both sides of a face share them -- except at the two z faces of a z-PERIODIC domain, where the reference copies the finished
// edge value of the opposite boundary face (:1008-1019), hydrostatic part of THAT level included.
struct FaceBg { double hyr, hyt, p0, ihyt; };

template <bool STRICT, int ORD>
__device__ __forceinline__ void face_flux(const DyP &p, const double *__restrict__ cL, int eL, const double *__restrict__ cR,
                                          int eR, long long st, int nrm, const FaceBg &bL, const FaceBg &bR,
                                          bool zero_nrm, double *__restrict__ f, long long fV) {
  const double cs = 350;
  double rL = edge_value<STRICT, ORD>(cL + idR * p.sV, st, eL) + bL.hyr;
  double rR = edge_value<STRICT, ORD>(cR + idR * p.sV, st, eR) + bR.hyr;
  double uL = edge_value<STRICT, ORD>(cL + nrm * p.sV, st, eL);
  double uR = edge_value<STRICT, ORD>(cR + nrm * p.sV, st, eR);
  double eTL = edge_value<STRICT, ORD>(cL + idT * p.sV, st, eL), eTR = edge_value<STRICT, ORD>(cR + idT * p.sV, st, eR);
  double tL = eTL + bL.hyt;
  double tR = eTR + bR.hyt;
  if (STRICT) {
#pragma clang fp contract(off)
    double mL = zero_nrm ? 0.0 : uL * rL;
    double mR = zero_nrm ? 0.0 : uR * rR;
    double p_L = p.C0 * pow_gamma<true>(tL, p.gamma), p_R = p.C0 * pow_gamma<true>(tR, p.gamma);
    double w1 = 0.5 * (p_R - cs * mR);
    double w2 = 0.5 * (p_L + cs * mL);
    double p_upw = w1 + w2;
    double m_upw = (w2 - w1) / cs;
    int ind = (mL + mR > 0) ? 0 : 1;
    double r_upw = ind ? rR : rL;
    const double *cU = ind ? cR : cL;  int eU = ind ? eR : eL;
    f[idR * fV] = m_upw;
    f[nrm * fV] = m_upw * (ind ? mR : mL) / r_upw + p_upw;
    f[idT * fV] = m_upw * (ind ? tR : tL) / r_upw;
    for (int l = idU; l < p.V; l++) {
      if (l == nrm || l == idT) continue;
      double val = edge_value<true, ORD>(cU + l * p.sV, st, eU) * r_upw;
      f[l * fV] = m_upw * val / r_upw;
    }
  } else {
#pragma clang fp contract(fast)
    double mL = zero_nrm ? 0.0 : uL * rL;
    double mR = zero_nrm ? 0.0 : uR * rR;
    double p_L = pressure_fast(p, eTL, bL.hyt, bL.p0, bL.ihyt), p_R = pressure_fast(p, eTR, bR.hyt, bR.p0, bR.ihyt);
    double w1 = 0.5 * (p_R - cs * mR);
    double w2 = 0.5 * (p_L + cs * mL);
    double p_upw = w1 + w2;
    double m_upw = (w2 - w1) * (1.0 / 350.0);
    int ind = (mL + mR > 0) ? 0 : 1;
    double r_upw = ind ? rR : rL;
    const double *cU = ind ? cR : cL;  int eU = ind ? eR : eL;
    double u_upw = zero_nrm ? 0.0 : (ind ? uR : uL);
    f[idR * fV] = m_upw;
    f[nrm * fV] = m_upw * u_upw + p_upw;
    f[idT * fV] = m_upw * (ind ? tR : tL) / r_upw;
    for (int l = idU; l < p.V; l++) {
      if (l == nrm || l == idT) continue;
      f[l * fV] = m_upw * edge_value<false, ORD>(cU + l * p.sV, st, eU);
    }
  }
}

template <bool STRICT, int ORD>
__global__ __launch_bounds__(256) void k_flux(DyP p, const double *__restrict__ S, double *__restrict__ FX,
                                              double *__restrict__ FY, double *__restrict__ FZ) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int NXF = (p.nx + 1) * p.nens;
  int nyf = p.sim2d ? p.ny : p.ny + 1;                       // 2-D: no y faces beyond row 0 needed
  if (t >= (long long)nyf * NXF) return;
  int j = (int)(t / NXF), ie = (int)(t - (long long)j * NXF);
  int i = ie / p.nens, e = ie - i * p.nens;
  const double *c = S + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)(i + p.HX) * p.nens + e;
  // ---------------- X face i-1/2 : left state = right edge of cell i-1, right state = left edge of cell i
  if (j < p.ny && k < p.nz) {
    const double *cL = c - p.nens, *cR = c;  int eL = 1, eR = 0;  bool zero = false;
    if (p.bc_x != MW_BC_PERIODIC) {                          // :1040-1060 (note the else-if: quirk 1)
      if (p.px == 0) {
        if (i == 0) { cL = cR; eL = eR; zero = (p.bc_x == MW_BC_WALL); }
        else if (i == p.nx && p.nproc_x == 1) { cR = c - (long long)p.nx * p.nens; eR = 0; }   // slot 1 at nx keeps what the periodic self-exchange delivered: left edge of cell 0 (:985)
      } else if (p.px == p.nproc_x - 1) {
        if (i == p.nx) { cR = cL; eR = eL; zero = (p.bc_x == MW_BC_WALL); }
      }
    }
    const FaceBg bg = {p.hyc[k * p.nens + e], p.hytc[k * p.nens + e], p.p0c[k * p.nens + e], p.ihytc[k * p.nens + e]};
    face_flux<STRICT, ORD>(p, cL, eL, cR, eR, p.nens, idU, bg, bg, zero, FX + (long long)k * p.fxK + (long long)j * p.fxJ + ie, p.fxV);
  }
  // ---------------- Y face j-1/2
  if (!p.sim2d && i < p.nx && k < p.nz) {
    const double *cL = c - p.sJ, *cR = c;  int eL = 1, eR = 0;  bool zero = false;
    if (p.bc_y != MW_BC_PERIODIC) {                          // :1061-1081
      if (p.py == 0) {
        if (j == 0) { cL = cR; eL = eR; zero = (p.bc_y == MW_BC_WALL); }
        else if (j == p.ny && p.nproc_y == 1) { cR = c - (long long)p.ny * p.sJ; eR = 0; }
      } else if (p.py == p.nproc_y - 1) {
        if (j == p.ny) { cR = cL; eR = eL; zero = (p.bc_y == MW_BC_WALL); }
      }
    }
    const FaceBg bg = {p.hyc[k * p.nens + e], p.hytc[k * p.nens + e], p.p0c[k * p.nens + e], p.ihytc[k * p.nens + e]};
    face_flux<STRICT, ORD>(p, cL, eL, cR, eR, p.sJ, idV, bg, bg, zero, FY + (long long)k * p.fyK + (long long)j * p.fyJ + ie, p.fyV);
  }
  // ---------------- Z face k-1/2 : wall / open edge-value rule at k = 0 and k = nz  (:1020-1038)
  if (i < p.nx && j < p.ny) {
    const double *cL = c - p.sK, *cR = c;  int eL = 1, eR = 0;  bool zero = false;
    int kL = k, kR = k;                                      // which face's hydrostatic edge values each side carries (:368-377)
    if (p.bc_z == MW_BC_PERIODIC) {                          // :1008-1019: slot 0 of face 0 := slot 0 of face nz, slot 1 of face nz := slot 1 of face 0
      if (k == 0)    { cL = c + (long long)(p.nz - 1) * p.sK; kL = p.nz; }       // top edge of cell nz-1, as finished at face nz
      if (k == p.nz) { cR = c - (long long)p.nz * p.sK;       kR = 0;    }       // bottom edge of cell 0, as finished at face 0
    } else {                                                 // :1020-1038 wall / open
      if (k == 0)    { cL = cR; eL = eR; zero = (p.bc_z == MW_BC_WALL); }
      if (k == p.nz) { cR = cL; eR = eL; zero = (p.bc_z == MW_BC_WALL); }
    }
    const FaceBg bL = {p.hye[kL * p.nens + e], p.hyte[kL * p.nens + e], p.p0e[kL * p.nens + e], p.ihyte[kL * p.nens + e]};
    const FaceBg bR = {p.hye[kR * p.nens + e], p.hyte[kR * p.nens + e], p.p0e[kR * p.nens + e], p.ihyte[kR * p.nens + e]};
    face_flux<STRICT, ORD>(p, cL, eL, cR, eR, p.sK, idW, bL, bR, zero,
                      FZ + (long long)k * p.fzK + (long long)j * p.fzJ + (long long)i * p.nens + e, p.fzV);
  }
}

// -----------------------------------------------------------------------------------------------------
// D10  FCT positivity (:498-516).  In-place scaling of outgoing tracer fluxes; race-free by the reference's
// sign argument (:495-497): a face is only ever rescaled by the cell it leaves.
// -----------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void k_fct(DyP p, const double *__restrict__ S, double *__restrict__ FX,
                                             double *__restrict__ FY, double *__restrict__ FZ, double dt) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y, tr = blockIdx.z;
  if (!((p.pos_mask >> tr) & 1u)) return;
  int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  int e = ie % p.nens;
  const double *s = S + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
  double rho = s[idR * p.sV] + p.hyc[k * p.nens + e];
  double tracer = s[(5 + tr) * p.sV] * rho;                 // the re-multiplied value (:482) that FCT reads
  double dx = p.dx, dy = p.dy, dz = p.dz;
  double *fx = FX + (long long)(5 + tr) * p.fxV + (long long)k * p.fxK + (long long)j * p.fxJ + ie;
  double *fy = FY + (long long)(5 + tr) * p.fyV + (long long)k * p.fyK + (long long)j * p.fyJ + ie;
  double *fz = FZ + (long long)(5 + tr) * p.fzV + (long long)k * p.fzK + (long long)j * p.fzJ + ie;
  double fxm = fx[0], fxp = fx[p.nens], fym = fy[0], fyp = fy[p.fyJ], fzm = fz[0], fzp = fz[p.fzK];
  double mass_available = fmax(tracer, 0.0) * dx * dy * dz;
  double flux_out_x, flux_out_y, flux_out_z;
  if (FAST) {   // production path: multiply by the reciprocal grid spacings (1 ulp from the reference's divisions)
    flux_out_x = (fmax(fxp, 0.0) - fmin(fxm, 0.0)) * p.rdx;
    flux_out_y = (fmax(fyp, 0.0) - fmin(fym, 0.0)) * p.rdy;
    flux_out_z = (fmax(fzp, 0.0) - fmin(fzm, 0.0)) * p.rdz;
  } else {
    flux_out_x = (fmax(fxp, 0.0) - fmin(fxm, 0.0)) / dx;
    flux_out_y = (fmax(fyp, 0.0) - fmin(fym, 0.0)) / dy;
    flux_out_z = (fmax(fzp, 0.0) - fmin(fzm, 0.0)) / dz;
  }
  double mass_out = (flux_out_x + flux_out_y + flux_out_z) * dt * dx * dy * dz;
  if (mass_out > mass_available) {
    double mult = mass_available / mass_out;
    if (fxp > 0) fx[p.nens] = fxp * mult;
    if (fxm < 0) fx[0]      = fxm * mult;
    if (fyp > 0) fy[p.fyJ]  = fyp * mult;
    if (fym < 0) fy[0]      = fym * mult;
    if (fzp > 0) fz[p.fzK]  = fzp * mult;
    if (fzm < 0) fz[0]      = fzm * mult;
  }
}

// -----------------------------------------------------------------------------------------------------
// D11 tendencies (:519-551) + D12 SSPRK3 combine (:121-174) + storage divide (next stage's D2, :248-255)
//   MODE 0: write the new slab (stored form)            STAGE 1: q* = q^n + dt L(q^n)
//   MODE 1: last stage of the last cycle: write the     STAGE 2: q* = 3/4 q^n + 1/4 q* + 1/4 dt L(q*)
//           COUPLER fields directly (D13, :1927-1950)    STAGE 3: q  = 1/3 q^n + 2/3 q* + 2/3 dt L(q*)
//   MODE 2: write tendencies only (mw_dycore_compute_tendencies)
// Sstar = slab the fluxes were computed from; Sn = q^n slab (== Sstar in stage 1).
// -----------------------------------------------------------------------------------------------------
template <int STAGE, int MODE>
__global__ __launch_bounds__(256) void k_update(DyP p, const double *Sstar, const double *Sn,   // may alias Sout (in-place stages)
                                                double *Sout, const double *__restrict__ FX,
                                                const double *__restrict__ FY, const double *__restrict__ FZ,
                                                double dt_stage, double dt_dyn, CouplerPtrs c,
                                                double *__restrict__ state_tend, double *__restrict__ tracers_tend) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  int e = ie % p.nens;
  long long so = (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
  long long ci = ((long long)k * p.ny + j) * NXI + ie;
  const double *fx = FX + (long long)k * p.fxK + (long long)j * p.fxJ + ie;
  const double *fy = FY + (long long)k * p.fyK + (long long)j * p.fyJ + ie;
  const double *fz = FZ + (long long)k * p.fzK + (long long)j * p.fzJ + ie;
  const double hyc = p.hyc[k * p.nens + e], hytc = p.hytc[k * p.nens + e];
  const double dx = p.dx, dy = p.dy, dz = p.dz;
  const double rho_s = Sstar[so + idR * p.sV] + hyc;                 // density of the stage input
  const double rho_n = (STAGE == 1) ? rho_s : Sn[so + idR * p.sV] + hyc;
  // immersed-boundary relaxation coefficients (:534-550)
  double prop = 0, imm_coef = 0;
  if (p.use_immersed) {
    double tau = 1.e3 * dt_stage;
    imm_coef = -fmin(1.0, dt_stage / tau);
    prop = p.imm[ci];
  }
  const double ru_s = Sstar[so + idU * p.sV] * rho_s;                // re-multiplied momenta (:478-480)
  const double rv_s = Sstar[so + idV * p.sV] * rho_s;
  double newR = 0, rho_new = 0;                                      // filled at l == idR (first iteration)
  double rho_dry_acc = 0, rho_v_new = 0, press_arg = 0, unew = 0, vnew = 0, wnew = 0;
  for (int l = 0; l < p.V; l++) {
    // conserved value of the stage input and of q^n
    double raw_s = Sstar[so + l * p.sV];
    double q_s = (l == idR || l == idT) ? raw_s : raw_s * rho_s;
    double q_n;
    if (STAGE == 1) q_n = q_s;
    else { double raw_n = Sn[so + l * p.sV]; q_n = (l == idR || l == idT) ? raw_n : raw_n * rho_n; }
    double tend = -(fx[l * p.fxV + p.nens] - fx[l * p.fxV]) / dx
                  -(fy[l * p.fyV + p.fyJ ] - fy[l * p.fyV]) / dy
                  -(fz[l * p.fzV + p.fzK ] - fz[l * p.fzV]) / dz;
    if (l == idW && p.enable_gravity) tend += -p.grav * rho_s;
    if (l == idU) tend += p.fcor * rv_s;
    if (l == idV) tend -= p.fcor * ru_s;
    if (l == idV && p.sim2d) tend = 0;
    if (p.use_immersed && l < 5) {
      double imm_tend = imm_coef * q_s / dt_stage;
      tend = prop * imm_tend + (1 - prop) * tend;
    }
    if (MODE == 2) {
      if (l < 5) state_tend[(long long)l * p.nC + ci] = tend;
      else       tracers_tend[(long long)(l - 5) * p.nC + ci] = tend;
      continue;
    }
    double qnew;
    if (STAGE == 1)      qnew = q_n + dt_dyn * tend;
    else if (STAGE == 2) qnew = (3.0 / 4.0) * q_n + (1.0 / 4.0) * q_s + (1.0 / 4.0) * dt_dyn * tend;
    else                 qnew = (1.0 / 3.0) * q_n + (2.0 / 3.0) * q_s + (2.0 / 3.0) * dt_dyn * tend;
    if (l >= 5 && ((p.pos_mask >> (l - 5)) & 1u)) qnew = fmax(0.0, qnew);
    if (l == idR) { newR = qnew; rho_new = newR + hyc; rho_dry_acc = rho_new; }
    if (MODE == 0) {
      Sout[so + l * p.sV] = (l == idR || l == idT) ? qnew : qnew / rho_new;
    } else {  // MODE 1: convert_dynamics_to_coupler (:1927-1950)
      if (l == idU) unew = qnew / rho_new;
      if (l == idV) vnew = qnew / rho_new;
      if (l == idW) wnew = qnew / rho_new;
      if (l == idT) { double theta = (qnew + hytc) / rho_new; press_arg = rho_new * theta; }
      if (l >= 5) {
        c.tr[l - 5][ci] = qnew;
        if (l - 5 == p.idWV) rho_v_new = qnew;
        if ((p.mass_mask >> (l - 5)) & 1u) rho_dry_acc -= qnew;
      }
    }
  }
  if (MODE == 1) {
    double press = p.C0 * pow_ref(press_arg, p.gamma);
    double temp = press / (rho_dry_acc * p.R_d + rho_v_new * p.R_v);
    c.rho_d[ci] = rho_dry_acc;  c.u[ci] = unew;  c.v[ci] = vnew;  c.w[ci] = wnew;  c.temp[ci] = temp;
  }
}

} // namespace mw
#include "mw_march.h"
#include "mw_calib.h"      // calibration kernels: fp64 FMA ceiling, the arithmetic floor of a stage (WENO + Riemann on registers)
namespace mw {

// -----------------------------------------------------------------------------------------------------
// Initial data (input construction, :1197-1683, :1687-1887).  Column profiles are built on the host
// (mw_init.cpp part below); the per-cell quadrature + convert_dynamics_to_coupler (:1656) runs here.
// -----------------------------------------------------------------------------------------------------
struct InitP {
  int init_data, ord;
  long long i_beg, j_beg;
  double xlen, ylen, cp_d, p0;
  const double *hyDensGLL, *hyDensThetaGLL, *hyDensVapGLL;     // supercell: (nz,5) device
  const double *bheights; int nbx, nby, cells_per_building, buildings_pad, nblocks_x, nblocks_y;   // city
  long long nx_glob, ny_glob;
};

__device__ __forceinline__ void d_hydro_const_theta(double z, double grav, double C0, double cp, double p0, double gamma,
                                                    double rd, double &r, double &t) {     // :1108-1117
#pragma clang fp contract(off)
  const double theta0 = 300., exner0 = 1.;
  t = theta0;
  double exner = exner0 - grav * z / (cp * theta0);
  double pr = p0 * pow_ref(exner, (cp / rd));
  double rt = pow_ref((pr / C0), (1.0 / gamma));
  r = rt / t;
}
__device__ __forceinline__ double d_sample_ellipse_cosine(double amp, double x, double y, double z, double x0, double y0,
                                                          double z0, double xrad, double yrad, double zrad) {   // :1121-1134
#pragma clang fp contract(off)
  double dist = sqrt(((x - x0) / xrad) * ((x - x0) / xrad) + ((y - y0) / yrad) * ((y - y0) / yrad) +
                     ((z - z0) / zrad) * ((z - z0) / zrad)) * M_PI / 2.;
  if (dist <= M_PI / 2.) return amp * pow_ref(cos_ref(dist), 2.0);
  return 0.;
}

__constant__ double c_gll5_pts[5] = {-0.50000000000000000000000000000000000000, -0.32732683535398857189914622812342917778,
                                     0.00000000000000000000000000000000000000, 0.32732683535398857189914622812342917778,
                                     0.50000000000000000000000000000000000000};      // TransformMatrices.h:650-656
__constant__ double c_gll5_wts[5] = {0.050000000000000000000000000000000000000, 0.27222222222222222222222222222222222222,
                                     0.35555555555555555555555555555555555556, 0.27222222222222222222222222222222222222,
                                     0.050000000000000000000000000000000000000};     // :659-665
__constant__ double c_gll3_pts[3] = {-0.50000000000000000000000000000000000000, 0.00000000000000000000000000000000000000,
                                     0.50000000000000000000000000000000000000};      // TransformMatrices.h:83-88 (MW_ORD = 3)
__constant__ double c_gll3_wts[3] = {0.16666666666666666666666666666666666667, 0.66666666666666666666666666666666666667,
                                     0.16666666666666666666666666666666666667};      // :90-95
__constant__ double c_gll9_pts[9] = {-0.50000000000000000000000000000000000000, -0.44987899770573007865617262220916897903,
                                     -0.33859313975536887672294271354567122536, -0.18155873191308907935537603435432960651,
                                     0.00000000000000000000000000000000000000, 0.18155873191308907935537603435432960651,
                                     0.33859313975536887672294271354567122536, 0.44987899770573007865617262220916897903,
                                     0.50000000000000000000000000000000000000};      // :4113-4124
__constant__ double c_gll9_wts[9] = {0.013888888888888888888888888888888888889, 0.082747680780402762523169860014604152919,
                                     0.13726935625008086764035280928968636297, 0.17321425548652317255756576606985914397,
                                     0.18575963718820861678004535147392290249, 0.17321425548652317255756576606985914397,
                                     0.13726935625008086764035280928968636297, 0.082747680780402762523169860014604152919,
                                     0.013888888888888888888888888888888888889};     // :4126-4137
__constant__ double c_gll7_pts[7] = MW_GLL7_PTS;                                    // get_gll_points / _weights(SArray<FP,1,7>): mw_weno79.h
__constant__ double c_gll7_wts[7] = MW_GLL7_WTS;
__constant__ double c_gl3_pts[3] = {0.112701665379258311482073460022, 0.500000000000000000000000000000,
                                    0.887298334620741688517926539980};                // :1349-1351
__constant__ double c_gl3_wts[3] = {0.277777777777777777777777777779, 0.444444444444444444444444444444,
                                    0.277777777777777777777777777779};                // :1353-1355

__global__ __launch_bounds__(256) void k_init_cells(DyP p, InitP q, CouplerPtrs c, double *__restrict__ imm) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  int i = ie / p.nens, e = ie - i * p.nens;
  long long ci = ((long long)k * p.ny + j) * NXI + ie;
  double sR = 0, sU = 0, sV = 0, sW = 0, sT = 0, sWV = 0;
  const double dx = p.dx, dy = p.dy, dz = p.dz;
  if (q.init_data == MW_DATA_SUPERCELL) {                     // :1843-1886  (ord GLL points per direction)
    const int no = q.ord;
    const double *gp = (no == 3) ? c_gll3_pts : (no == 7) ? c_gll7_pts : (no == 9) ? c_gll9_pts : c_gll5_pts;
    const double *gw = (no == 3) ? c_gll3_wts : (no == 7) ? c_gll7_wts : (no == 9) ? c_gll9_wts : c_gll5_wts;
    for (int kk = 0; kk < no; kk++) for (int jj = 0; jj < no; jj++) for (int ii = 0; ii < no; ii++) {
      double zloc = (k + 0.5) * dz + gp[kk] * dz;
      double dens = q.hyDensGLL[k * no + kk];
      double uvel;
      const double zs = 5000, us = 30, uc = 15;
      if (zloc < zs) uvel = us * (zloc / zs) - uc; else uvel = us - uc;
      double vvel = 0, wvel = 0;
      double dens_vap = q.hyDensVapGLL[k * no + kk], dens_theta = q.hyDensThetaGLL[k * no + kk];
      double factor = gw[ii] * gw[jj] * gw[kk];
      sR += (dens - q.hyDensGLL[k * no + kk]) * factor;
      sU += dens * uvel * factor;
      sV += dens * vvel * factor;
      sW += dens * wvel * factor;
      sT += (dens_theta - q.hyDensThetaGLL[k * no + kk]) * factor;
      sWV += dens_vap * factor;
    }
  } else {                                                     // thermal :1361-1392 ; city :1463-1503 ; building :1566-1607
    const int nq = (q.init_data == MW_DATA_THERMAL) ? 3 : 9;
    const double *qp = (q.init_data == MW_DATA_THERMAL) ? c_gl3_pts : c_gll9_pts;
    const double *qw = (q.init_data == MW_DATA_THERMAL) ? c_gl3_wts : c_gll9_wts;
    for (int kk = 0; kk < nq; kk++) for (int jj = 0; jj < nq; jj++) for (int ii = 0; ii < nq; ii++) {
      double x = (i + q.i_beg + 0.5) * dx + (qp[ii] - 0.5) * dx;
      double y = (j + q.j_beg + 0.5) * dy + (qp[jj] - 0.5) * dy;   if (p.sim2d) y = q.ylen / 2;
      double z = (k + 0.5) * dz + (qp[kk] - 0.5) * dz;
      double rho, u, v, w, theta, rho_v, hr, ht;
      if (q.init_data == MW_DATA_THERMAL) {                    // thermal(), :1086-1103
        d_hydro_const_theta(z, p.grav, p.C0, q.cp_d, q.p0, p.gamma, p.R_d, hr, ht);
        double rho_d = hr;
        u = 0.; v = 0.; w = 0.;
        double theta_d = ht + d_sample_ellipse_cosine(2.0, x, y, z, q.xlen / 2, q.ylen / 2, 2000., 2000., 2000., 2000.);
        double p_d = p.C0 * pow_ref(rho_d * theta_d, p.gamma);
        double temp = p_d / rho_d / p.R_d;
        double tc = temp - 273.15;                             // saturation_vapor_pressure, :1137-1140
        double sat_pv = 610.94 * exp_ref(17.625 * tc / (243.04 + tc));
        double sat_rv = sat_pv / p.R_v / temp;
        rho_v = d_sample_ellipse_cosine(0.8, x, y, z, q.xlen / 2, q.ylen / 2, 2000., 2000., 2000., 2000.) * sat_rv;
        double pr = rho_d * p.R_d * temp + rho_v * p.R_v * temp;
        rho = rho_d + rho_v;
        theta = pow_ref(pr / p.C0, 1.0 / p.gamma) / rho;
      } else {
        if (p.enable_gravity) d_hydro_const_theta(z, p.grav, p.C0, q.cp_d, q.p0, p.gamma, p.R_d, hr, ht);
        else { hr = 1.15; ht = 300; }
        rho = hr; u = 20; v = 0; w = 0; theta = ht; rho_v = 0;
      }
      if (p.sim2d) v = 0;
      double wt = qw[ii] * qw[jj] * qw[kk];
      sR += (rho - hr) * wt;
      sU += rho * u * wt;
      sV += rho * v * wt;
      sW += rho * w * wt;
      sT += (rho * theta - hr * ht) * wt;
      sWV += rho_v * wt;
    }
    if (q.init_data == MW_DATA_CITY) {                         // :1504-1514
      int inorm = ((int)q.i_beg + i) / q.cells_per_building - q.buildings_pad;
      int jnorm = ((int)q.j_beg + j) / q.cells_per_building - q.buildings_pad;
      if ((inorm >= 0 && inorm < q.nblocks_x * 3 && inorm % 3 < 2) && (jnorm >= 0 && jnorm < q.nblocks_y * 9 && jnorm % 9 < 8)) {
        if (k <= ceil(q.bheights[(long long)jnorm * q.nbx + inorm] / dz)) imm[ci] = 1;
      }
    } else if (q.init_data == MW_DATA_BUILDING) {              // :1608-1617
      double x0 = 0.3 * q.nx_glob, y0 = 0.5 * q.ny_glob, xr = 0.05 * q.ny_glob, yr = 0.05 * q.ny_glob;
      if (fabs((double)(q.i_beg + i) - x0) <= xr && fabs((double)(q.j_beg + j) - y0) <= yr && k <= 0.2 * p.nz) imm[ci] = 1;
    }
  }
  // convert_dynamics_to_coupler (:1927-1950); all tracers other than water vapour start at zero
  double hyc = p.hyc[k * p.nens + e], hytc = p.hytc[k * p.nens + e];
  double rho = sR + hyc;
  double u = sU / rho, v = sV / rho, w = sW / rho;
  double theta = (sT + hytc) / rho;
  double press = p.C0 * pow_ref(rho * theta, p.gamma);
  double rho_d = rho;
  for (int tr = 0; tr < p.nt; tr++) {
    double val = (tr == p.idWV) ? sWV : 0.0;
    if ((p.mass_mask >> tr) & 1u) rho_d -= val;
    c.tr[tr][ci] = val;
  }
  double temp = press / (rho_d * p.R_d + sWV * p.R_v);
  c.rho_d[ci] = rho_d; c.u[ci] = u; c.v[ci] = v; c.w[ci] = w; c.temp[ci] = temp;
}

// modules::perturb_temperature(thermal=true)   perturb_temperature.h:41-66
__global__ __launch_bounds__(256) void k_perturb_temperature(int nz, int ny, int nx, int nens, long long i_beg, long long j_beg,
                                                             double dx, double dy, double dz, double xlen, double ylen,
                                                             double *__restrict__ temp) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  long long n = (long long)nz * ny * nx * nens;
  if (t >= n) return;
  long long r = t / nens;
  int i = (int)(r % nx); r /= nx;
  int j = (int)(r % ny); int k = (int)(r / ny);
  double xloc = (i + i_beg + 0.5) * dx, yloc = (j + j_beg + 0.5) * dy, zloc = (k + 0.5) * dz;
  double x0 = xlen / 2, y0 = ylen / 2, z0 = 1500, radx = 10000, rady = 10000, radz = 1500, amp = 5;
  double xn = (xloc - x0) / radx, yn = (yloc - y0) / rady, zn = (zloc - z0) / radz;
  double rad = sqrt(xn * xn + yn * yn + zn * zn);
  if (rad < 1) temp[t] += amp * pow_ref(cos_ref(M_PI * rad / 2), 2.0);
}

// modules::perturb_temperature(random=true)   perturb_temperature.h:25-39: the lowest nz/4 levels get uniform noise in [-1, 1] * 3 K,
// fading linearly with height; every (level, column) draws from its own generator seeded with a globally unique key
// (myrank*nz*nx*ny*nens + k*ncol + i).  yakl::Random is not available (empty submodule): the same key goes through the splitmix64
// finaliser (53 random bits -> [0, 1)), the substitution the surrogate-data sampler uses (mw_output.hip); INTEGRATION.md says so.
__global__ __launch_bounds__(256) void k_perturb_temperature_random(int num_levels, long long ncol, unsigned long long seed,
                                                                    double *__restrict__ temp) {
#pragma clang fp contract(off)
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)num_levels * ncol) return;
  const int k = (int)(t / ncol);
  unsigned long long z = seed + (unsigned long long)t + 0x9E3779B97F4A7C15ull;              // t = k*ncol + i
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const double u01 = (double)(z >> 11) * (1.0 / 9007199254740992.0);
  const double rnd = u01 * 2.0 - 1.0;
  const double scaling = (num_levels - (double)k) / num_levels;
  temp[t] += rnd * 3.0 * scaling;                               // (levels are the slowest index: t addresses temp(k, column) directly)
}

// Streaming copy with this library's access shape (8 bytes per lane, consecutive lanes consecutive doubles): the known-byte
// workload used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section).
// Diagnostic: the device WENO-5 routines on caller-supplied stencils (unit test of the core arithmetic against the golden vectors)
__global__ __launch_bounds__(256) void k_weno5_edges(const double *__restrict__ st, double *__restrict__ out, long long n, int strict) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const double *s = st + t * 5;
  double l, r;
  if (strict) weno5_edges_strict(s[0], s[1], s[2], s[3], s[4], l, r);
  else        weno5_edges_fast(s[0], s[1], s[2], s[3], s[4], l, r);
  out[t * 2] = l; out[t * 2 + 1] = r;
}

// Diagnostic: the strict path's pow (mw_glibc_pow.h) on caller-supplied arguments; main[i] = 1 where the restated main path applied
__global__ __launch_bounds__(256) void k_strict_pow(const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out,
                                                    unsigned char *__restrict__ main_path, long long n) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  double r;
  const bool m = glibc_pow_main(x[t], y[t], &r);
  out[t] = m ? r : pow(x[t], y[t]);
  if (main_path) main_path[t] = m ? 1 : 0;
}

// member-major slab (nens, V, nz+2HZ, ny+2HY, nx+2HX) -> the fused layout (V, nz+2HZ, ny+2HY, (nx+2HX)*nens), halos included.
// p = the FUSED parameter block.  Used when the public flux arrays are rebuilt from a stage input of the production path.
__global__ __launch_bounds__(256) void k_member_to_fused(DyP p, const double *__restrict__ src, double *__restrict__ dst) {
  const long long n = (long long)p.V * p.sV;                  // fused elements
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int e = (int)(t % p.nens);
  const long long r = t / p.nens;                             // (v, k, j, i) with the member-major strides
  dst[t] = src[(long long)e * (n / p.nens) + r];
}

// ColumnNudger's increment (column_nudging.h:62-65): dt (column - average) / time_scale, one number per (field, level, member) -- the
// reference's expression, IEEE division, no contraction (= k_nudge_apply in mw_column.hip).
__global__ __launch_bounds__(256) void k_nudge_increments(const double *__restrict__ column, const double *__restrict__ avg, double dt, long long n, double *__restrict__ inc) {
#pragma clang fp contract(off)
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const double time_scale = 900;
  if (t < n) inc[t] = dt * (column[t] - avg[t]) / time_scale;
}
// ... and parked increments applied by a pass after all (mw_dycore_flush_pending: somebody wants to see the fields before the next time step,
// or the next time step runs a path whose conversion does not take them along): state(l,k,j,i,e) += inc(l,k,e), the same rounded addition.
struct Ptr5 { double *f[5]; };
__global__ __launch_bounds__(256) void k_apply_pending(Ptr5 fp, int nz, long long ncell_lev, int nens, const double *__restrict__ inc) {
#pragma clang fp contract(off)
  const int k = blockIdx.y, l = blockIdx.z;
  const long long n = ncell_lev * nens;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int e = (int)(t % nens);
    double *q = fp.f[l] + (long long)k * n;
    q[t] = q[t] + inc[((long long)l * nz + k) * nens + e];
  }
}

__global__ __launch_bounds__(256) void k_calib_copy(const double *__restrict__ in, double *__restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

} // namespace mw

// =====================================================================================================
// Host side
// =====================================================================================================
using namespace mw;

// Run-time options of a handle (mw_dycore_set_option / mw_dycore_get_option; rounds 1-4 read MW_* environment variables per launch
// instead -- process-global, untyped and racy under threads).  Typed integers, read where the schedule of a time step is decided.
struct DyOpts {
  int overlap = -1;            // two-stream schedule (state | tracer pipelines): -1 = automatic (with a transport installed), 0 / 1 forced
  int pipe = 1;                // with a transport: the pipelined one-stream schedule (rk_stage_pipe) where k_y_all applies
  int pipe_edge_inline = 0;    // ... its two edge strips of the y launch on the compute stream instead of the exchange stream
  int pipe_convert = 1;        // ... D1 of the inner rows inside the first k_y_all<true>
  int pipe_split_edges = 1;    // ... the next stage's edge strips split into a state part (behind the state strips) and a tracer part
  int spec = 1;                // folded configurations of the marching kernels (Cf<1>, Cf<2>)
  int wrap = 1;                // index wrap instead of halo cells in a periodic direction owned by one rank
  int y_all = 1, y_all_conv = 1;      // y faces of all variables in one launch; ... also the converting first stage
  int member_major = 1, mm_direct = 1, mm_conv = 1;   // nens > 1: member-major arrays; D13 / D1 inside the members-in-one-workgroup launches
  int fused_convert = 1, fused_convert_mm = 1;        // D1 inside the first y launch (one rank, periodic x and y)
  int chunk_y = 0, chunk_yt = 0, chunk_z = 0, chunk_f = 0;   // cells per chunk of the marching kernels (0 = the chunk model)
  int chunk_model = 1;
  int tf_rows4 = 1;            // tracer stage: workgroup = 4 rows of one x tile (0: 4 tiles of one row)
  int zero_skip = 1;           // wave-uniform short-cut for tracers that are exactly zero over a wavefront's stencil (bit-neutral; 0: A/B)
  int zero_rows = 1;           // ... and the zero-row maps on top of it: rows of a tracer that are known to be zero are not loaded (mw_march.h: k_zero_rows)
  int pipe_maps_early = 1;     // pipelined schedule, first stage: local zero-row maps in front of its y launches, two strip exchanges (0: one exchange, maps beside the y launch; A/B)
  int zero_stores = 1;         // ... and zeros are not stored over rows that hold zeros already (the coupler's arrays, slabs S1 / S2; 0: A/B)
  int zero_verify = 0;         // test aid: check the maps' claims against the data in front of every launch that relies on them (k_zero_verify; mw_debug_zero_violations)
  int rccl_lanes = 0, rccl_two_comms = -1;            // built-in RCCL transport: side streams (1 | 2), a communicator per lane (0 | 1); 0 / -1 = the
                                                      // process default (MW_RCCL_LANES); read when mw_dycore_use_rccl* installs the transport
  int rccl_prio = 1;           // ... its side streams at the highest stream priority (0: default priority; A/B)
  int rccl_inline = 1;         // ... the send / receive group on the caller's stream instead of a side stream of the transport's own
  int xchg_fuzz = 0;           // test aid: seeded random delays (spin kernels) around the built-in transport's sends / receives
  int debug_no_patch = 0;      // test aid: the y-face correction pass of the fused tracer stage is not launched (the negative control of the FCT tests)
};

struct mw_dycore_s {
  mw_grid_t g;
  DyOpts o;
  unsigned char pos[MW_MAX_TRACERS], adds[MW_MAX_TRACERS];
  hipStream_t stream;
  DyP p;
  double *S0 = nullptr, *S1 = nullptr, *S2 = nullptr, *S3 = nullptr;   // q^n and three stage slabs (rotated, never aliased)
  double *M[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};          // upwind mass flux of every x/y/z face,
  unsigned char *UP[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // upwind selector; double-buffered by stage parity
  hipStream_t tstream = nullptr;                        // tracer pipeline (runs one stage behind / beside the state pipeline)
  hipEvent_t ev_state[8] = {nullptr}, ev_tr[8] = {nullptr}, ev_misc = nullptr;
  long long gstage = 0;                                 // global stage counter (event ring index, buffer parity)
  int overlap = 1;
  int pre_lo = 0, pre_hi = 0;                // pipelined schedule: rows outside [pre_lo, pre_hi) (and the W / E strip columns) were converted up front
  int last_march = 0;                        // the last time_step ran on the marching kernels (mw_dycore_schedule)
  std::string path;                          // what the dispatcher chose for the last time_step, spelled out (mw_dycore_path)
  int pipe = 0;                              // blocks of a decomposed domain: pipelined one-stream schedule (rk_stage_pipe)
  bool pipe_ready = false;                   // ... the next stage's input strips are already on their way (event ev_pipe[2])
  bool pipe_edge_done = false;               // ... and its two edge strips of the y launch were issued behind them on the exchange stream
  hipEvent_t ev_pipe[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool entry_marked = false;                 // ev_pipe[6] was recorded at this time step's entry   // [0], [1]: compute -> exchange stream; [2]: state strips + state edge rows ready; [3]: tracer strips + tracer edge faces ready; [4]: zero-row maps ready; [5]: the block's local zero-row maps ready (exchange -> compute stream); [6]: time_step entered (the coupler's arrays are ready)
  double *tendY = nullptr;                              // (5,nz,ny,nx,nens) y part of the state tendencies
  double *FX = nullptr, *FY = nullptr, *FZ = nullptr;
  const double *flux_src = nullptr; double flux_dt = 0; // stage input + dt of the last stage (state fluxes on demand)
  int chunk_y = 0, chunk_yt = 0, chunk_z = 0, chunk_f = 0;
  bool first_cycle = false;                // the running sub-cycle is the time step's first (zero_rows_build)
  bool conv_pending = false;               // time_step: the coupler -> slab conversion is still to be done by stage 1's k_y_state
  unsigned int *dirty = nullptr;           // two words: "a y face was scaled in this / the next fused tracer launch"
  unsigned long long fused_launches = 0;
  unsigned char *flags = nullptr;          // fused tracer stage: per-cell "a y face of this cell was FCT-scaled" bits
  double *zrx = nullptr;                   // ... and the message buffers of a decomposed block's map exchange (own | rW | rE | sS | sN | rS | rN)
  const double *kz_buf[2] = {nullptr, nullptr};   // ... and, for the two slabs that take turns as q^n, "the rows the last conversion into it left zero" (maps behind MC; nullptr: unknown)
  int zr_cur = 0;                          // ... double-buffered: set zr_cur belongs to the running sub-cycle, the other one to the one before
  bool zr_prev_ok = false, zr_prev_use = false;   // the other set describes what slabs S1 / S2 hold now (the sub-cycle before ran with maps, nothing else wrote the slabs since) / ... and is handed to this sub-cycle's kernels
  unsigned long long *zviol = nullptr;     // option zero_verify: four violation counters (k_zero_verify)
  double *pinc = nullptr; bool pinc_on = false; double *pinc_fields[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // parked column increments (mw_nudge_to_column_deferred) and the five arrays they belong to
  unsigned long long pinc_lazy = 0, pinc_eager = 0;             // how often parked increments rode on the conversion / were applied by a pass
  unsigned *zr = nullptr; long long zr_msz = 0; bool zr_on = false;   // zero-row maps (M0 and the six of k_zero_dilate) of the running sub-cycle (mw_march.h: k_zero_rows), zr_msz words each
  bool strides_ok = true;                  // fill_params: every stride of DyP fits its Stride32
  int fused = 0;                           // 1: fused tracer stage (k_tracers_fused + k_tracer_patch)
  double *hy_dev = nullptr;                  // hyc | hytc | hye | hyte | p0c | ihytc | p0e | ihyte | packed rows (see DyP::hypk)
  double *imm = nullptr;
  std::vector<double> hy_host;               // same packing (the last four are derived in upload_background)
  double etime = 0;
  int strict = 0;
  int mm_direct = 0;                         // ... and D13 written from the last stage's kernels (MemberOff: 2 or 4 members per workgroup), no k_member_to_coupler pass
  int member_major = 0;                      // production path with nens > 1: the handle's arrays hold one member after the other (View)
  int ord = 5;                               // WENO order (3, 7, 9: the reference's -DMW_ORD builds; they run on the general kernels)
  int hxw = HXc, hzw = HZc;                  // halo widths of the slabs: hs + 1 in x / y, hs in z (3 / 2 up to order 5)
  // halo exchange
  mw_exchange_fn xchg = nullptr; void *xchg_ctx = nullptr;
  double *bufs[2][8] = {{nullptr}, {nullptr}};   // [group: 0 state (or all), 1 tracers][sW sE sS sN rW rE rS rN]
  long long nWE1 = 0, nSN1 = 0;                  // per variable
  // profiling
  int prof = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[12];      // kernel classes 0..7; 8 = one whole RK stage (all its launches); 9 = one whole time_step; 10 / 11 = the compute stream's waits for the state / tracer strips (pipelined schedule)
  size_t ev_used[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  void (*xchg_free)(void *) = nullptr;       // set when the handle owns xchg_ctx (the built-in RCCL transport, mw_rccl.cpp)
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Padding the blocks-per-plane count to a
// multiple of 8 puts the block that owns tile (j,i) of level k+1 on the SAME XCD as the block of level k, so the z-neighbour
// face/cell reads of the plane kernels hit that XCD's L2 instead of going out to the Infinity Cache / HBM.
static inline dim3 plane_grid(long long per_plane, int nk, int nzdim = 1) {
  unsigned nb = (unsigned)((per_plane + 255) / 256);
  nb = (nb + 7u) & ~7u;
  return dim3(nb, (unsigned)nk, (unsigned)nzdim);
}

static void fill_params(mw_dycore_s *d) {
  const mw_grid_t &g = d->g;  DyP &p = d->p;
  p.nz = g.nz; p.ny = g.ny; p.nx = g.nx; p.nens = g.nens; p.nt = g.num_tracers; p.V = 5 + g.num_tracers;
  p.sim2d = (g.ny_glob == 1);
  p.HX = d->hxw; p.HY = p.sim2d ? 0 : d->hxw; p.HZ = d->hzw;
  p.NXE = (g.nx + 2 * p.HX) * g.nens;
  p.sJ = p.NXE; p.sK = (long long)(g.ny + 2 * p.HY) * p.sJ; p.sV = (long long)(g.nz + 2 * p.HZ) * p.sK;
  p.nC = (long long)g.nz * g.ny * g.nx * g.nens;
  p.fxJ = (long long)(g.nx + 1) * g.nens; p.fxK = (long long)g.ny * p.fxJ;       p.fxV = (long long)g.nz * p.fxK;
  p.fyJ = (long long)g.nx * g.nens;       p.fyK = (long long)(g.ny + 1) * p.fyJ; p.fyV = (long long)g.nz * p.fyK;
  p.fzJ = (long long)g.nx * g.nens;       p.fzK = (long long)g.ny * p.fzJ;       p.fzV = (long long)(g.nz + 1) * p.fzK;
  // (Stride32: the largest stride is a slab's variable stride; the products above were formed from converted values, so check it in 64 bits)
  d->strides_ok = (long long)(g.nz + 2 * p.HZ) * (long long)(g.ny + 2 * p.HY) * (long long)p.NXE <= 2147483647ll &&
                  (long long)(g.nz + 1) * (long long)(g.ny + 1) * (long long)(g.nx + 1) * g.nens <= 2147483647ll;
  p.v0 = 0; p.wrap_x = 0; p.wrap_y = 0; p.cst = 1; p.ce = 0;
  p.bc_x = g.bc_x; p.bc_y = g.bc_y; p.bc_z = g.bc_z; p.px = g.px; p.py = g.py; p.nproc_x = g.nproc_x; p.nproc_y = g.nproc_y;
  p.enable_gravity = g.enable_gravity; p.use_immersed = g.use_immersed; p.idWV = g.idWV;
  p.zero_skip = d->o.zero_skip;
  p.zq = p.zqp = p.zqc = p.zqk = nullptr; p.zq_ld = 0;          // (set per RK stage by zero_rows_stage / zero_rows_conv)
  p.pinc = nullptr;                                             // (set by mw_dycore_time_step when parked increments ride on the conversion)
  p.pos_mask = 0; p.mass_mask = 0;
  for (int t = 0; t < g.num_tracers; t++) { if (d->pos[t]) p.pos_mask |= 1u << t; if (d->adds[t]) p.mass_mask |= 1u << t; }
  p.dx = g.xlen / g.nx_glob; p.dy = g.ylen / g.ny_glob; p.dz = g.zlen / g.nz;        // coupler.h:262-268
  p.rdx = 1.0 / p.dx; p.rdy = 1.0 / p.dy; p.rdz = 1.0 / p.dz;
  p.C0 = g.C0; p.gamma = g.gamma_d; p.grav = g.grav; p.R_d = g.R_d; p.R_v = g.R_v;
  p.fcor = 2 * g.earthrot * sin(g.latitude);                                           // :213
  size_t nzc = (size_t)g.nz * g.nens, nze = (size_t)(g.nz + 1) * g.nens;
  p.hyc = d->hy_dev; p.hytc = d->hy_dev + nzc; p.hye = d->hy_dev + 2 * nzc; p.hyte = d->hy_dev + 2 * nzc + nze;
  const double *ext = d->hy_dev + 2 * nzc + 2 * nze;
  p.p0c = ext; p.ihytc = ext + nzc; p.p0e = ext + 2 * nzc; p.ihyte = ext + 2 * nzc + nze;
  p.imm = d->imm;
  p.hypk = d->hy_dev + 4 * nzc + 4 * nze;
  long double bn = 1.0L;                                    // C(gamma, n) = C(gamma, n-1) (gamma - n + 1) / n
  p.bn[0] = 1.0;
  for (int n = 1; n <= 10; n++) { bn = bn * ((long double)g.gamma_d - (n - 1)) / n; p.bn[n] = (double)bn; }
  p.bn_default = 1;
  for (int n = 0; n <= 10; n++) if (p.bn[n] != BN_DEFAULT[n]) p.bn_default = 0;
  { static const double AN_DEFAULT[11] = {1.0, 0x1.6d7ed9f857ccfp-1, -0x1.a255770e765c3p-4, 0x1.66b0e7bdc9cadp-5, -0x1.9a025de3c9f2fp-6,
                                         0x1.0d783d4c75011p-6, -0x1.80febd2957c9ap-7, 0x1.22bbebca1e45p-7, -0x1.c8e61cbaa3102p-8,
                                         0x1.71e467e9895fp-8, -0x1.327f77d85aeeep-8};
    const long double a = 1.0L / (long double)g.gamma_d;
    long double an = 1.0L;                                  // C(1/gamma, n)
    p.an_default = 1;
    for (int n = 1; n <= 10; n++) { an = an * (a - (n - 1)) / n; if ((double)an != AN_DEFAULT[n]) p.an_default = 0; } }
}

static int upload_background(mw_dycore_s *d) {
  {  // derived tables of the fast pressure path: p0 = C0 hyt^gamma, 1/hyt (cells and edges)
    const mw_grid_t &g = d->g;
    size_t nzc = (size_t)g.nz * g.nens, nze = (size_t)(g.nz + 1) * g.nens;
    double *h = d->hy_host.data();
    const double *hytc = h + nzc, *hyte = h + 2 * nzc + nze;
    double *ext = h + 2 * nzc + 2 * nze;
    for (size_t n = 0; n < nzc; n++) { ext[n] = g.C0 * pow(hytc[n], g.gamma_d); ext[nzc + n] = 1.0 / hytc[n]; }
    for (size_t n = 0; n < nze; n++) { ext[2 * nzc + n] = g.C0 * pow(hyte[n], g.gamma_d); ext[2 * nzc + nze + n] = 1.0 / hyte[n]; }
    double *pk = h + 4 * nzc + 4 * nze;                      // packed rows
    const double *src[8] = {h, h + nzc, ext, ext + nzc, h + 2 * nzc, h + 2 * nzc + nze, ext + 2 * nzc, ext + 2 * nzc + nze};
    for (size_t n = 0; n < nze; n++) for (int f = 0; f < 8; f++) pk[n * 8 + f] = (f < 4 && n >= nzc) ? 0.0 : src[f][n];
    // member-major copies (View): member e's columns contiguous, so that the nens = 1 kernels index them with k alone
    double *mm = pk + 8 * nze;
    const size_t per = 4 * (size_t)g.nz + 8 * (size_t)(g.nz + 1);
    for (int e = 0; e < g.nens; e++) {
      double *m = mm + (size_t)e * per;
      for (int k = 0; k < g.nz; k++) { const size_t n = (size_t)k * g.nens + e;
        m[k] = h[n]; m[g.nz + k] = hytc[n]; m[2 * g.nz + k] = ext[n]; m[3 * g.nz + k] = ext[nzc + n]; }
      double *mpk = m + 4 * g.nz;
      for (int k = 0; k <= g.nz; k++) for (int f = 0; f < 8; f++) mpk[(size_t)k * 8 + f] = pk[((size_t)k * g.nens + e) * 8 + f];
    }
  }
  MW_HIP(hipMemcpyAsync(d->hy_dev, d->hy_host.data(), d->hy_host.size() * sizeof(double), hipMemcpyHostToDevice, d->stream));
  MW_HIP(hipStreamSynchronize(d->stream));
  return 0;
}

struct ProfScope {
  mw_dycore_s *d; int which; size_t idx; bool on; hipStream_t st;
  ProfScope(mw_dycore_s *d_, int w, hipStream_t st_ = nullptr) : d(d_), which(w), idx(0), on((d_->prof == 1 && w != 9) || (d_->prof == 2 && (w == 0 || w == 8)) || (d_->prof == 3 && w == 9)), st(st_ ? st_ : d_->stream) {
    if (!on) return;
    if (d->ev_used[which] == d->ev[which].size()) {
      hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); d->ev[which].push_back({a, b});
    }
    idx = d->ev_used[which]++;
    (void)hipEventRecord(d->ev[which][idx].first, st);
  }
  ~ProfScope() { if (on) (void)hipEventRecord(d->ev[which][idx].second, st); }
};

// ---------------------------------------------------------------------------------------------------------------------
// Member-major mode (production path, nens > 1).  With the coupler's member-fastest layout the x stencil of a wave cannot use
// the DPP lane shifts (x neighbours are nens lanes apart) and the marching kernels fall back to neighbour loads + ds_bpermute:
// k_xz_state is 36 % slower per cell at nens = 4.  The handle's INTERNAL arrays (slabs, tendY, M/UP, FY, side arrays, flags) are
// ours to lay out, so with nens > 1 they hold one member after the other and every production kernel is launched once per member
// in its nens = 1 form (View: the member's parameter block and base pointers); only the coupler-side accesses -- conversion in,
// D13 out, immersed proportion -- are strided (DyP::cst / ce, cpl() in mw_march.h).  The general kernels keep the fused layout in
// the same allocations (a handle runs one path or the other within a time_step; get_fluxes transposes the retained stage input).
// ---------------------------------------------------------------------------------------------------------------------
struct View {
  DyP p;
  long long slab, tend, m[3], f[3], cells;          // member e's offsets = e * these (doubles; UP / flags: bytes = the same counts)
  int e;
  template <class T> T *S(T *base) const { return base ? base + e * slab : base; }
};
static int n_views(const mw_dycore_s *d) { return d->member_major ? d->p.nens : 1; }
static View view(const mw_dycore_s *d, int e) {
  View v; v.e = e; v.p = d->p;
  v.slab = v.tend = v.cells = 0; for (int a = 0; a < 3; a++) v.m[a] = v.f[a] = 0;
  if (!d->member_major) { v.e = 0; return v; }
  DyP &q = v.p;
  const int n = d->p.nens;
  q.nens = 1; q.cst = n; q.ce = e;
  q.NXE = q.nx + 2 * q.HX;
  q.sJ = q.NXE; q.sK = (long long)(q.ny + 2 * q.HY) * q.sJ; q.sV = (long long)(q.nz + 2 * q.HZ) * q.sK;
  q.nC = (long long)q.nz * q.ny * q.nx;
  q.fxJ = q.nx + 1; q.fxK = (long long)q.ny * q.fxJ;       q.fxV = (long long)q.nz * q.fxK;
  q.fyJ = q.nx;     q.fyK = (long long)(q.ny + 1) * q.fyJ; q.fyV = (long long)q.nz * q.fyK;
  q.fzJ = q.nx;     q.fzK = (long long)q.ny * q.fzJ;       q.fzV = (long long)(q.nz + 1) * q.fzK;
  const size_t nzc = (size_t)q.nz * n, nze = (size_t)(q.nz + 1) * n, per = 4 * (size_t)q.nz + 8 * (size_t)(q.nz + 1);
  const double *m = d->hy_dev + 4 * nzc + 4 * nze + 8 * nze + (size_t)e * per;
  q.hyc = m; q.hytc = m + q.nz; q.p0c = m + 2 * q.nz; q.ihytc = m + 3 * q.nz; q.hypk = m + 4 * q.nz;
  q.hye = q.hyte = q.p0e = q.ihyte = nullptr;              // (edge tables: only through hypk on this path)
  if (q.zq) {                                                  // the member's own zero-row maps (zero_rows_build)
    const long long mstride = (2 * MW_ZR_MAPS + 3) * d->zr_msz;
    q.zq += e * mstride; if (q.zqp) q.zqp += e * mstride;
    q.zqc = q.zqk = nullptr;
  }
  v.slab = (long long)q.V * q.sV; v.tend = 5 * q.nC; v.cells = q.nC;
  v.m[0] = q.fxV; v.m[1] = q.fyV; v.m[2] = q.fzV;
  v.f[0] = (long long)q.V * q.fxV; v.f[1] = (long long)q.V * q.fyV; v.f[2] = (long long)q.V * q.fzV;
  return v;
}

// the member-to-member strides of a member-major handle, for the member-transposing kernels (MemberOff, mw_march.h)
static MemberOff member_off(const mw_dycore_s *d) {
  const View v = view(d, 0);
  MemberOff mo;
  mo.slab = v.slab; mo.tend = v.tend; mo.mx = v.m[0]; mo.my = v.m[1]; mo.mz = v.m[2]; mo.fx = v.f[0]; mo.fy = v.f[1]; mo.fz = v.f[2];
  mo.cells = v.cells; mo.per = 4 * (long long)v.p.nz + 8 * (long long)(v.p.nz + 1);
  mo.n = d->p.nens; mo.sh = d->p.nens == 4 ? 2 : 1;
  mo.zq = (2 * MW_ZR_MAPS + 3) * d->zr_msz;
  return mo;
}

// halo fill of variables [v0, v0+nv) of one slab: neighbour exchange (or self wrap) in x/y, then BCs -- replaces
// halo_exchange (:574-827).  `grp` selects the pack-buffer set (0: state or all variables, 1: tracers).
static int halo_fill(mw_dycore_s *d, double *Sbase, int v0 = 0, int nv = -1, hipStream_t st = nullptr, int grp = 0, bool skip_z = false) {
  if (!st) st = d->stream;
  if (nv < 0) nv = d->p.V;
  if (nv == 0) return 0;
  ProfScope ps(d, 3, st);
  // skip_z = production path; with a member-major handle the slab holds one member after the other: every kernel below runs once
  // per member on its nens = 1 view, and the strips of member e sit at e * (strip size / nens) in the exchange buffers
  const int nv_views = (skip_z && d->member_major) ? d->p.nens : 1;
  const DyP &pf = d->p;
  bool ex_x = d->xchg && (pf.nproc_x > 1), ex_y = d->xchg && (pf.nproc_y > 1) && !pf.sim2d;
  long long nWE = d->nWE1 * nv, nSN = d->nSN1 * nv;              // all members
  const long long mWE = nWE / nv_views, mSN = nSN / nv_views;    // one view
  double **bf = d->bufs[grp];
  auto member = [&](int e, DyP &q, double *&S) {
    View v; if (nv_views > 1) v = view(d, e); else { v.p = d->p; v.e = 0; v.slab = 0; }
    q = v.p; q.v0 = v0; q.V = nv;
    S = Sbase + e * v.slab + (long long)v0 * q.sV;
  };
  if (ex_x || ex_y) {
    // a rank grid with more than one rank in a direction: ship 3-cell strips to the face neighbours.
    // Directions with a single rank still wrap locally below.
    for (int e = 0; e < nv_views; e++) {
      DyP q; double *S; member(e, q, S);
      const unsigned nbx = ex_x ? (unsigned)((mWE + 255) / 256) : 0u, nby = ex_y ? (unsigned)((mSN + 255) / 256) : 0u;
      MW_KLAUNCH(k_pack_xy, dim3(nbx + nby), dim3(256), 0, st, q, S, bf[0] + e * mWE, bf[1] + e * mWE, bf[2] + e * mSN, bf[3] + e * mSN, nbx); MW_LAUNCH_CHECK();
    }
    int rc = d->xchg(d->xchg_ctx, ex_x ? bf[0] : nullptr, ex_x ? bf[1] : nullptr, ex_y ? bf[2] : nullptr, ex_y ? bf[3] : nullptr,
                     ex_x ? bf[4] : nullptr, ex_x ? bf[5] : nullptr, ex_y ? bf[6] : nullptr, ex_y ? bf[7] : nullptr, ex_x ? nWE : 0,
                     ex_y ? nSN : 0, st);
    if (rc) MW_FAIL("halo exchange callback failed");
    for (int e = 0; e < nv_views; e++) {
      DyP q; double *S; member(e, q, S);
      const unsigned nbx = ex_x ? (unsigned)((mWE + 255) / 256) : 0u, nby = ex_y ? (unsigned)((mSN + 255) / 256) : 0u;
      MW_KLAUNCH(k_unpack_xy, dim3(nbx + nby), dim3(256), 0, st, q, S, bf[4] + e * mWE, bf[5] + e * mWE, bf[6] + e * mSN, bf[7] + e * mSN, nbx); MW_LAUNCH_CHECK();
    }
  }
  // local wrap / BC:  x when this direction has one rank (periodic self-wrap), or a wall / open boundary on a domain-edge rank.
  // With several ranks in a non-periodic direction the edge ranks have just exchanged with their periodic-wrap neighbour like
  // the reference does (:641-723 uses the periodic neighbour matrix) and the boundary rule then OVERWRITES that halo
  // (:782-825: `px == 0` west side, `px == nproc_x-1` east side; the kernel skips the sides that are rank-interior).
  for (int e = 0; e < nv_views; e++) {
    DyP q; double *S; member(e, q, S);
    const DyP &p = q;
    const bool edge_x = (p.px == 0 || p.px == p.nproc_x - 1), edge_y = (p.py == 0 || p.py == p.nproc_y - 1);
    const bool bcx_after = ex_x && p.bc_x != MW_BC_PERIODIC && edge_x, bcy_after = ex_y && p.bc_y != MW_BC_PERIODIC && edge_y;
    const long long nx_ = (long long)p.V * p.nz * p.ny * 2 * p.HX * p.nens;
    const long long ny_ = (long long)p.V * p.nz * 2 * p.HY * p.nx * p.nens;
    const long long nz_ = (long long)p.V * 2 * p.HZ * p.ny * p.nx * p.nens;
    const unsigned nbx = ((ex_x && !bcx_after) || (skip_z && p.wrap_x)) ? 0u : (unsigned)((nx_ + 255) / 256);      // skip_z = production path
    const unsigned nby = ((ex_y && !bcy_after) || p.sim2d || (skip_z && p.wrap_y)) ? 0u : (unsigned)((ny_ + 255) / 256);
    const unsigned nbz = skip_z ? 0u : (unsigned)((nz_ + 255) / 256);      // (the marching kernels apply the z rule while loading)
    if (nbx + nby + nbz) { MW_KLAUNCH(k_halo_xyz, dim3(nbx + nby + nbz), dim3(256), 0, st, p, S, nbx, nby); MW_LAUNCH_CHECK(); }
  }
  return 0;
}

static int launch_flux(mw_dycore_s *d, const double *S) {
  ProfScope ps(d, 0);
  const DyP &p = d->p;
  long long per_plane = (long long)(p.sim2d ? p.ny : p.ny + 1) * (p.nx + 1) * p.nens;
  dim3 grid = plane_grid(per_plane, p.nz + 1);
  if (d->ord == 3) {
    if (d->strict == 1) MW_KLAUNCH((k_flux<true, 3>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
    else                MW_KLAUNCH((k_flux<false, 3>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
  } else if (d->ord == 7) {
    if (d->strict == 1) MW_KLAUNCH((k_flux<true, 7>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
    else                MW_KLAUNCH((k_flux<false, 7>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
  } else if (d->ord == 9) {
    if (d->strict == 1) MW_KLAUNCH((k_flux<true, 9>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
    else                MW_KLAUNCH((k_flux<false, 9>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
  } else {
    if (d->strict == 1) MW_KLAUNCH((k_flux<true, 5>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
    else                MW_KLAUNCH((k_flux<false, 5>), grid, dim3(256), 0, d->stream, p, S, d->FX, d->FY, d->FZ);
  }
  MW_LAUNCH_CHECK();
  return 0;
}

static int launch_fct(mw_dycore_s *d, const double *S, double dt, hipStream_t st = nullptr) {
  const DyP &p = d->p;
  if (!st) st = d->stream;
  if (p.nt == 0 || p.pos_mask == 0) return 0;
  ProfScope ps(d, 1, st);
  dim3 grid = plane_grid((long long)p.ny * p.nx * p.nens, p.nz, p.nt);
  if (d->strict == 1) MW_KLAUNCH(k_fct<false>, grid, dim3(256), 0, st, p, S, d->FX, d->FY, d->FZ, dt);
  else                MW_KLAUNCH(k_fct<true>, grid, dim3(256), 0, st, p, S, d->FX, d->FY, d->FZ, dt);
  MW_LAUNCH_CHECK();
  return 0;
}

template <int STAGE, int MODE>
static int launch_update(mw_dycore_s *d, const double *Sstar, const double *Sn, double *Sout, double dt_stage, double dt_dyn,
                         const CouplerPtrs &c, double *st, double *tt) {
  ProfScope ps(d, 2);
  const DyP &p = d->p;
  dim3 grid = plane_grid((long long)p.ny * p.nx * p.nens, p.nz);
  MW_KLAUNCH((k_update<STAGE, MODE>), grid, dim3(256), 0, d->stream, p, Sstar, Sn, Sout, d->FX, d->FY, d->FZ, dt_stage,
                     dt_dyn, c, st, tt);
  MW_LAUNCH_CHECK();
  return 0;
}

// Equal chunks along the marching direction: enough of them for `target` waves (x/z kernels: ~5 rounds of 2 waves/SIMD over
// the 1024 SIMDs), none shorter than 8 cells (every chunk re-primes its pipeline).
// The count of chunks decides how many workgroups a CU holds at once, and that quantises the run time -- measured on 100 x 100 x 50:
// 0.37 ms per step with 10 chunks of 5 levels (250 workgroups: one per CU, every wave alone on its SIMD), 0.47 ms with 6 chunks of
// 9 (300 workgroups: 44 CUs hold two).  Model: duration = t(B) * (chunk + o) * (1 + chunk / 1000) with B workgroups on 256 CUs that
// hold `bpc` of them each; a CU's last, partly filled round costs less than a full one (a wave that has its SIMD to itself runs
// ~1.7 x faster); o = cells of work a chunk adds (ghost levels, pipeline priming); the last factor is what long chunks lose in
// cache locality.  Fitted to chunk sweeps on 100 x 100 x 50, 200 x 200 x 50, 256 x 256 x 64, 300 x 300 x 80 and 400 x 400 x 100
// (tools/tail_probe.py); `model` = false keeps the older rule (enough chunks for `target` waves, none under 8 cells).
static int balanced_chunk(const mw_dycore_s *d, int nz, long long base_waves, int forced, long long target, int bpc, double o, bool model) {
  if (forced > 0) return std::min(nz, forced);                 // (options chunk_y / chunk_yt / chunk_z / chunk_f)
  long long nch = std::max(1ll, (target + base_waves - 1) / base_waves);
  nch = std::min<long long>(nch, std::max(1, nz / 8));
  const long long cap = 256ll * bpc;
  if (!model || !d->o.chunk_model) return (int)((nz + nch - 1) / nch);
  auto t_of = [&](long long B) {
    const long long full = B / cap, rem = B - full * cap;
    if (rem == 0) return (double)full;
    const long long m = (rem + 255) / 256;                     // workgroups per CU in the last round
    const double part = (bpc == 2) ? (m == 1 ? 0.6 : 1.0) : (m == 1 ? 0.45 : m == 2 ? 0.75 : 1.0);
    return (double)full + part;
  };
  auto cost_of = [&](int chunk) {
    const int neff = (nz + chunk - 1) / chunk;                 // the chunk length decides; neff chunks result
    return t_of(((base_waves + 3) / 4) * neff) * (chunk + o) * (1.0 + 0.001 * chunk);
  };
  const int old_chunk = (int)((nz + nch - 1) / nch);
  double best = 1e300; int best_chunk = nz;
  for (int n = 1; n <= std::max(1, nz / 4); n++) {
    const int chunk = (nz + n - 1) / n;
    const double cost = cost_of(chunk);
    if (cost < best * (1.0 - 1e-9)) { best = cost; best_chunk = chunk; }
  }
  // the model is crude: it overrides the older rule only where it promises more than 3 % (small and odd-sized launches: 9-26 %
  // measured; the tuned large grids keep their chunks)
  return (best < 0.97 * cost_of(old_chunk)) ? best_chunk : old_chunk;
}

static int device_cus() {
  static int n = -1;
  if (n < 0) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; else n = 0; }
  return n;
}
// Which compile-time configuration of the marching kernels (Cf<K>) fits this view of the handle: 1 / 2 = the shipped supercell /
// simple_city set-ups with their run-time switches folded, 0 = everything at run time.  Option "spec" = 0 forces 0 (A/B timing, tests).
static int marching_config(const mw_dycore_s *d, const DyP &p) {
  if (!d->o.spec) return 0;                                    // (option "spec" = 0: A/B timing, tests)
  const unsigned all = (1u << p.nt) - 1u;
  if (p.nens != 1 || p.sim2d || p.bc_x != MW_BC_PERIODIC || p.bc_y != MW_BC_PERIODIC || p.bc_z != MW_BC_WALL || p.fcor != 0.0 ||
      !p.bn_default || !p.an_default || p.pos_mask != all || p.mass_mask != all || p.idWV != 0) return 0;
  if (p.nt == 3 && !p.use_immersed && p.enable_gravity) return 1;
  if (p.nt == 1 && p.use_immersed && !p.enable_gravity) return 2;
  return 0;
}

#define MW_Y_EDGE 4                                            // rows of an edge strip of the pipelined schedule (>= 4: the converting inner launch requests coupler rows up to row_end + 3 < ny)
// conv != nullptr: the slab S is still empty -- the kernel converts the coupler's fields on the way and fills it (k_y_state<true>)
// edges: only the two MW_Y_EDGE-row strips at the block's south / north end (pipelined multi-rank schedule, on stream st); a block too
// short to split (see launch_y_all) takes all its rows here
static int launch_y_state(mw_dycore_s *d, const double *S, int par, const CouplerPtrs *conv = nullptr, bool edges = false, hipStream_t st = nullptr) {
  if (d->p.sim2d) return 0;
  if (!st) st = d->stream;
  ProfScope ps(d, 5, st);
  if (conv && d->member_major) {
    // D1 inside the launch, member-major handle: ONE launch over the fused lanes (k_y_state<.., MM>): unit-stride reads of the
    // coupler's arrays, outputs into the members' arrays.  The folded configuration is decided on a member's view (nens = 1 there).
    const View v0 = view(d, 0);
    const DyP &p = d->p;
    const long long threads = (long long)p.nz * p.nx * p.nens;
    const long long mthreads = (long long)p.nz * p.nx;                                   // one member's: the chunk rule of the per-member launches
    int chunk = d->chunk_y ? d->chunk_y : (d->chunk_y = balanced_chunk(d, p.ny, (mthreads + 63) / 64, d->o.chunk_y, 5000, 2, 5.0, (mthreads + 255) / 256 < 96));
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)((p.ny + chunk - 1) / chunk));
    const MemberOff mo = member_off(d);
    const YMember mm = {v0.p.sJ, v0.p.sK, v0.p.sV, v0.slab, v0.p.fyJ, v0.p.fyK, v0.m[1], v0.p.nC, v0.tend, p.nx, mo.per, mo.n, mo.sh};
    double *Sw = const_cast<double *>(S);
    const int K = marching_config(d, v0.p);
    if (d->mm_direct && d->o.mm_conv) {      // the members of the same cells in one workgroup (k_y_state<.., MM = 2>)
      grid.x = (unsigned)((mthreads + 64 * (4 / mo.n) - 1) / (64 * (4 / mo.n)));
#define MW_YSM2(K_) { if (d->ord == 3) MW_YSM2O(K_, 3); else MW_YSM2O(K_, 5); }
#define MW_YSM2O(K_, O_) MW_KLAUNCH((k_y_state<true, K_, O_, 2>), grid, dim3(256), 0, d->stream, v0.p, S, d->M[par][1], d->UP[par][1], d->tendY, chunk, *conv, Sw, mm)
      if (K == 1) MW_YSM2(1) else if (K == 2) MW_YSM2(2) else MW_YSM2(0)
#undef MW_YSM2
#undef MW_YSM2O
      MW_LAUNCH_CHECK();
      return 0;
    }
#define MW_YSM(K_, O_) MW_KLAUNCH((k_y_state<true, K_, O_, 1>), grid, dim3(256), 0, d->stream, p, S, d->M[par][1], d->UP[par][1], d->tendY, chunk, *conv, Sw, mm)
    if (d->ord == 3) { if (K == 1) MW_YSM(1, 3); else if (K == 2) MW_YSM(2, 3); else MW_YSM(0, 3); }
    else             { if (K == 1) MW_YSM(1, 5); else if (K == 2) MW_YSM(2, 5); else MW_YSM(0, 5); }
#undef MW_YSM
    MW_LAUNCH_CHECK();
    return 0;
  }
  for (int e = 0; e < n_views(d); e++) {
    const View v = view(d, e);
    const DyP &p = v.p;
    long long threads = (long long)p.nz * p.nx * p.nens;
    // measured on 400x400x100 (625 wave columns): 8 x 50 rows for k_y_state, 14 x 29 for k_y_tracers (-5 % / -2 % vs. 32-row chunks)
    int chunk = d->chunk_y ? d->chunk_y : (d->chunk_y = balanced_chunk(d, p.ny, (threads + 63) / 64, d->o.chunk_y, 5000, 2, 5.0, (threads + 255) / 256 < 96));
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)((p.ny + chunk - 1) / chunk));
    if (edges && p.ny >= 4 * MW_Y_EDGE) { chunk = -MW_Y_EDGE; grid.y = 2u; }      // (k_y_state: chunk < 0 = the two edge strips)
    double *MY = d->M[par][1] + e * v.m[1]; unsigned char *UY = d->UP[par][1] + e * v.m[1];
#define MW_YS(CONV_, K_, O_, cp, sw) MW_KLAUNCH((k_y_state<CONV_, K_, O_>), grid, dim3(256), 0, st, p, v.S(S), MY, UY, d->tendY + e * v.tend, chunk, cp, sw, YMember())
#define MW_YS_K(K_) { if (d->ord == 3) { if (conv) MW_YS(true, K_, 3, *conv, Sw); else MW_YS(false, K_, 3, CouplerPtrs(), nullptr); } \
                      else             { if (conv) MW_YS(true, K_, 5, *conv, Sw); else MW_YS(false, K_, 5, CouplerPtrs(), nullptr); } }
    double *Sw = const_cast<double *>(v.S(S));
    switch (marching_config(d, p)) { case 1: MW_YS_K(1) break; case 2: MW_YS_K(2) break; default: MW_YS_K(0) break; }
#undef MW_YS_K
#undef MW_YS
    MW_LAUNCH_CHECK();
  }
  return 0;
}

// y faces of all variables in one launch (k_y_all): one-stream schedule, up to three tracers.  (With the switches at run time, K = 0,
// the eight register windows do not fit -- 90-96 VGPRs go to scratch -- and the one launch is still 2.5 % of the step faster than
// k_y_state + k_y_tracers: 5.44-5.49 against 5.57-5.62 ms on the supercell grid with MW_NO_SPEC=1.)
static bool y_all_ok(const mw_dycore_s *d) {
  return !d->overlap && d->fused && !d->p.sim2d && d->p.nt <= 3 && d->o.y_all;
}
// part: 0 = all rows; 1 = the rows whose chunks read no halo row (all of them with the row wrap), 2 = the two edge strips of
// MW_Y_EDGE rows (short chunks: their launch runs between the exchange and k_xz_state, with a quarter of the wavefronts)
static int launch_y_all(mw_dycore_s *d, const double *S, const CouplerPtrs *conv, int part = 0, hipStream_t st = nullptr) {
  if (!st) st = d->stream;
  ProfScope ps(d, 5, st);
  // the cells the pipelined schedule converted up front (time_step): the converting launch leaves them alone (see k_y_all)
  const int pre_lo = (conv && part == 1) ? d->pre_lo : 0, pre_hi = (conv && part == 1) ? d->pre_hi : 0;
  int fy_skip = 0;                                              // (set below where the inner launch shares its first / last face with the edge strips' launch)
  if (conv && d->member_major) {                                // mm_direct: all members in one launch, the members of the same cells in one workgroup
    const View v = view(d, 0);
    const DyP &p = v.p;
    if (!d->mm_direct || marching_config(d, p) == 0) MW_FAIL("internal: the converting k_y_all of a member-major handle exists for 2 or 4 members of a folded configuration only");
    const MemberOff mo = member_off(d);
    const long long mthreads = (long long)p.nz * p.nx;
    int chunk = d->chunk_y ? d->chunk_y : (d->chunk_y = balanced_chunk(d, p.ny, (mthreads + 63) / 64, d->o.chunk_y, 5000, 2, 5.0, (mthreads + 255) / 256 < 96));
    dim3 grid((unsigned)((mthreads + 64 * (4 / mo.n) - 1) / (64 * (4 / mo.n))), (unsigned)((p.ny + chunk - 1) / chunk));
    int row0 = 0, row_end = p.ny;
    if (part == 1 && !p.wrap_y) {                               // (pipelined schedule: the inner rows; the edge strips come from the slab later)
      const int n = (int)grid.y;
      row0 = MW_Y_EDGE; row_end = p.ny - MW_Y_EDGE; chunk = (row_end - row0 + n - 1) / n; grid.y = (unsigned)((row_end - row0 + chunk - 1) / chunk);
      fy_skip = 3;
    }
#define MW_YAM(K_, O_, T_) MW_KLAUNCH((k_y_all<true, K_, O_, T_, true>), grid, dim3(256), 0, st, p, S, d->FY, d->tendY, chunk, *conv, const_cast<double *>(S), mo, row0, chunk, row_end, pre_lo, pre_hi, fy_skip)
#define MW_YAM_O(K_, T_) { if (d->ord == 3) MW_YAM(K_, 3, T_); else MW_YAM(K_, 5, T_); }
    if (marching_config(d, p) == 1) MW_YAM_O(1, 3) else MW_YAM_O(2, 1)
#undef MW_YAM_O
#undef MW_YAM
    MW_LAUNCH_CHECK();
    return 0;
  }
  for (int e = 0; e < n_views(d); e++) {
    const View v = view(d, e);
    const DyP &p = v.p;
    long long threads = (long long)p.nz * p.nx * p.nens;
    int chunk = d->chunk_y ? d->chunk_y : (d->chunk_y = balanced_chunk(d, p.ny, (threads + 63) / 64, d->o.chunk_y, 5000, 2, 5.0, (threads + 255) / 256 < 96));
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)((p.ny + chunk - 1) / chunk));
    int row0 = 0, rstride = chunk, row_end = p.ny;
    if (part) {
      const int n = (int)grid.y;
      const bool edges = !p.wrap_y;                             // the first / last rows read halo rows of the slab
      const bool split = p.ny >= 4 * MW_Y_EDGE;                 // (an inner chunk reads up to 3 rows beyond its own: MW_Y_EDGE >= 3)
      if (part == 1) {
        if (edges) { if (!split) continue; row0 = MW_Y_EDGE; row_end = p.ny - MW_Y_EDGE; chunk = (row_end - row0 + n - 1) / n; rstride = chunk;
                     grid.y = (unsigned)((row_end - row0 + chunk - 1) / chunk); fy_skip = 3; }
      } else {
        if (!edges) continue;
        if (split) { chunk = MW_Y_EDGE; rstride = p.ny - MW_Y_EDGE; grid.y = 2u; }
      }
    }
#define MW_YA(C_, K_, O_, T_) MW_KLAUNCH((k_y_all<C_, K_, O_, T_>), grid, dim3(256), 0, st, p, v.S(S), d->FY + e * v.f[1], d->tendY + e * v.tend, chunk, \
                                         conv ? *conv : CouplerPtrs(), const_cast<double *>(v.S(S)), MemberOff(), row0, rstride, row_end, pre_lo, pre_hi, fy_skip)
#define MW_YA_O(K_, T_) { if (conv) { if (d->ord == 3) MW_YA(true, K_, 3, T_); else MW_YA(true, K_, 5, T_); } \
                          else      { if (d->ord == 3) MW_YA(false, K_, 3, T_); else MW_YA(false, K_, 5, T_); } }
    const int K = marching_config(d, p);
    if (K == 1) MW_YA_O(1, 3)
    else if (K == 2) MW_YA_O(2, 1)
    else if (p.nt == 1) MW_YA_O(0, 1)
    else if (p.nt == 2) MW_YA_O(0, 2)
    else MW_YA_O(0, 3)
#undef MW_YA_O
#undef MW_YA
    MW_LAUNCH_CHECK();
  }
  return 0;
}

static int launch_y_tracers(mw_dycore_s *d, const double *S, int par, hipStream_t st, bool edges = false) {
  if (d->p.sim2d) return 0;
  ProfScope ps(d, 6, st);
  for (int e = 0; e < n_views(d); e++) {
    const View v = view(d, e);
    const DyP &p = v.p;
    long long threads = (long long)p.nz * p.nx * p.nens;
    int chunk = d->chunk_yt ? d->chunk_yt : (d->chunk_yt = balanced_chunk(d, p.ny, (threads + 63) / 64, d->o.chunk_yt, 8400, 3, 5.0, (threads + 255) / 256 < 96));
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)((p.ny + chunk - 1) / chunk));
    if (edges && p.ny >= 4 * MW_Y_EDGE) { chunk = -MW_Y_EDGE; grid.y = 2u; }      // (k_y_tracers: chunk < 0 = the two edge strips)
    double *FY = d->FY + e * v.f[1];
    for (int t0 = 0; t0 < p.nt; t0 += 4) {
      int cnt = std::min(4, p.nt - t0);
      const double *M = d->M[par][1] + e * v.m[1]; const unsigned char *U = d->UP[par][1] + e * v.m[1];
#define MW_YT(T_) { if (d->ord == 3) MW_KLAUNCH((k_y_tracers<T_, 3>), grid, dim3(256), 0, st, p, v.S(S), FY, M, U, chunk, t0); \
                    else             MW_KLAUNCH((k_y_tracers<T_, 5>), grid, dim3(256), 0, st, p, v.S(S), FY, M, U, chunk, t0); }
      switch (cnt) { case 1: MW_YT(1) break; case 2: MW_YT(2) break; case 3: MW_YT(3) break; default: MW_YT(4) break; }
#undef MW_YT
      MW_LAUNCH_CHECK();
    }
  }
  return 0;
}

static int xz_grid(mw_dycore_s *d, const DyP &p, dim3 &grid, int &chunk, int &tiles_x) {
  int U = xz_cells_per_wave(p.nens, d->ord);
  if (U < 4) MW_FAIL("nens too large for the 64-lane x tiling (need nens <= 30)");
  tiles_x = (p.nx * p.nens + U - 1) / U;
  long long waves = (long long)p.ny * tiles_x;
  if (!d->chunk_z) {
    // k_xz_state: equal chunks, enough of them for ~5 rounds of 2 waves/SIMD over the 1024 SIMDs (measured on 400x400x100:
    // 4 x 25 levels beats 32,32,32,4 by 4 %)
    d->chunk_z = balanced_chunk(d, p.nz, waves, d->o.chunk_z, 10000, 2, 2.5, true);
    // k_xz_state<.., HPL = 1> keeps (chunk + 2) rows of 64 bytes in dynamic LDS: stay well inside the 64 KB a workgroup may have
    d->chunk_z = std::min(d->chunk_z, 900);
  }
  chunk = d->chunk_z;
  grid = dim3((unsigned)((waves + 3) / 4), (unsigned)((p.nz + chunk - 1) / chunk));
  return 0;
}

template <int STAGE, int MODE>
static int launch_xz_state(mw_dycore_s *d, const double *S, const double *Sn, double *Sout, double dt_stage, double dt_dyn, int par,
                           const CouplerPtrs &c) {
  ProfScope ps(d, 0);
  if constexpr (STAGE == 3 && MODE == 1) {
    if (d->mm_direct) {                                         // all members in one launch: workgroup = the nens members of 4 / nens tiles
      const View v = view(d, 0);
      const DyP &p = v.p;
      dim3 grid; int chunk, tiles_x;
      if (xz_grid(d, p, grid, chunk, tiles_x)) return 1;
      const MemberOff mo = member_off(d);
      const int wpb = 4 / mo.n;
      grid.x = (unsigned)(((long long)p.ny * tiles_x + wpb - 1) / wpb);
      // one background table per wave here: (chunk + 2) x 256 B of dynamic LDS on top of the kernel's ~20.5 KB of static LDS must fit
      // the 64 KB a workgroup may have -- shorter chunks for this launch when nz is large and the chunk rule asks for one long chunk
      { const int cap = (65536 - 21504) / 256 - 2;               // 170 levels
        if (chunk > cap) { chunk = cap; grid.y = (unsigned)((p.nz + chunk - 1) / chunk); } }
      const size_t lds = (size_t)(chunk + 2) * 64 * 4;
#define MW_XZ_MT(K_) { if (d->ord == 3) MW_XZ_MTO(K_, 3); else MW_XZ_MTO(K_, 5); }
#define MW_XZ_MTO(K_, O_) MW_KLAUNCH((k_xz_state<3, true, 1, 1, K_, O_, true>), grid, dim3(256), lds, d->stream, p, S, Sn, Sout, d->M[par][0], d->M[par][2], \
                                        d->UP[par][0], d->UP[par][2], d->tendY, dt_stage, dt_dyn, chunk, tiles_x, c.u, c.v, c.w, mo)
      if (marching_config(d, p) == 1) MW_XZ_MT(1) else MW_XZ_MT(0)
#undef MW_XZ_MT
#undef MW_XZ_MTO
      MW_LAUNCH_CHECK();
      return 0;
    }
  }
  for (int e = 0; e < n_views(d); e++) {
    const View v = view(d, e);
    const DyP &p = v.p;
    dim3 grid; int chunk, tiles_x;
    if (xz_grid(d, p, grid, chunk, tiles_x)) return 1;
    double *MX = d->M[par][0] + e * v.m[0], *MZ = d->M[par][2] + e * v.m[2], *tY = d->tendY + e * v.tend;
    unsigned char *UX = d->UP[par][0] + e * v.m[0], *UZ = d->UP[par][2] + e * v.m[2];
    // nens == 1 (also: one member of a member-major handle): the per-level background values come through LDS
    // (k_xz_state<.., HPL = 1>; dynamic LDS = the chunk's rows)
#define MW_XZ(N1_, HPL_, K_, O_, lds) MW_KLAUNCH((k_xz_state<STAGE, N1_, MODE, HPL_, K_, O_>), grid, dim3(256), (lds), d->stream, p, v.S(S), v.S(Sn), v.S(Sout), \
                                                 MX, MZ, UX, UZ, tY, dt_stage, dt_dyn, chunk, tiles_x, c.u, c.v, c.w, MemberOff())
#define MW_XZ_K(K_) { if (d->ord == 3) MW_XZ(true, 1, K_, 3, hpl_bytes); else MW_XZ(true, 1, K_, 5, hpl_bytes); }
    const size_t hpl_bytes = (size_t)(chunk + 2) * 64;
    if (p.nens == 1) {
      switch (marching_config(d, p)) { case 1: MW_XZ_K(1) break; case 2: MW_XZ_K(2) break; default: MW_XZ_K(0) break; }
    } else MW_XZ(false, 0, 0, 5, 0);                          // (the fused-layout form for nens > 1: WENO-5 only, see time_step)
#undef MW_XZ_K
#undef MW_XZ
    MW_LAUNCH_CHECK();
  }
  return 0;
}


template <int T, bool N1>
static void launch_xz_tracers_t(mw_dycore_s *d, const double *S, dim3 grid, int chunk, int tiles_x, int t0, int par, double dt, int rows4,
                                hipStream_t st) {
  MW_KLAUNCH((k_xz_tracers<T, N1>), grid, dim3(256), 0, st, d->p, S, d->FX, d->FY, d->FZ, d->M[par][0], d->M[par][2], d->UP[par][0],
                     d->UP[par][2], dt, chunk, tiles_x, t0, rows4);
}

// tracer x/z fluxes + FCT (dt = the stage's dt, like k_fct)
static int launch_xz_tracers(mw_dycore_s *d, const double *S, int par, double dt, hipStream_t st) {
  const DyP &p = d->p;
  ProfScope ps(d, 7, st);
  dim3 grid; int chunk, tiles_x;
  if (xz_grid(d, p, grid, chunk, tiles_x)) return 1;
  const int rows4 = p.ny >= 4 ? 1 : 0;                         // block = 4 rows of one x tile (shares the FY rows)
  if (rows4) grid.x = (unsigned)(((p.ny + 3) / 4) * tiles_x);
  for (int t0 = 0; t0 < p.nt; t0 += 4) {
    int cnt = std::min(4, p.nt - t0);
    if (p.nens == 1) {
      switch (cnt) { case 1: launch_xz_tracers_t<1, true>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     case 2: launch_xz_tracers_t<2, true>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     case 3: launch_xz_tracers_t<3, true>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     default: launch_xz_tracers_t<4, true>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break; }
    } else {
      switch (cnt) { case 1: launch_xz_tracers_t<1, false>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     case 2: launch_xz_tracers_t<2, false>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     case 3: launch_xz_tracers_t<3, false>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break;
                     default: launch_xz_tracers_t<4, false>(d, S, grid, chunk, tiles_x, t0, par, dt, rows4, st); break; }
    }
    MW_LAUNCH_CHECK();
  }
  return 0;
}

template <int STAGE, int MODE>
static int launch_tracer_update(mw_dycore_s *d, const double *Sstar, const double *Sn, double *Sout, double dt_dyn, const CouplerPtrs &c,
                                hipStream_t st) {
  ProfScope ps(d, 2, st);
  const DyP &p = d->p;
  dim3 grid = plane_grid((long long)p.ny * p.nx * p.nens, p.nz);
  MW_KLAUNCH((k_tracer_update<STAGE, MODE>), grid, dim3(256), 0, st, p, Sstar, Sn, Sout, d->FX, d->FY, d->FZ, dt_dyn, c);
  MW_LAUNCH_CHECK();
  return 0;
}

template <int STAGE, int MODE, int T, bool N1, int K, int ORD = 5>
static void launch_tracers_fused_t(mw_dycore_s *d, const View &v, const double *S, const double *Sn, double *Sout, dim3 grid, int chunk, int tiles_x, int par,
                                   double dt, double dt_dyn, const CouplerPtrs &c, int rows4, hipStream_t st) {
  const int e = v.e;
  MW_KLAUNCH((k_tracers_fused<STAGE, MODE, T, N1, K, ORD>), grid, dim3(256), 0, st, v.p, v.S(S), v.S(Sn), v.S(Sout), d->FY + e * v.f[1],
                     d->M[par][0] + e * v.m[0], d->M[par][2] + e * v.m[2], d->UP[par][0] + e * v.m[0], d->UP[par][2] + e * v.m[2],
                     d->FX + e * v.f[0], d->FZ + e * v.f[2], d->flags + e * v.cells, d->dirty + (d->fused_launches & 1), dt, dt_dyn, c, chunk, tiles_x, rows4, MemberOff());
}
// x/z tracer fluxes + FCT + update in one kernel, then the (normally empty) y-face correction
template <int STAGE, int MODE>
static int launch_tracers_fused(mw_dycore_s *d, const double *S, const double *Sn, double *Sout, int par, double dt, double dt_dyn,
                                const CouplerPtrs &c, hipStream_t st) {
  {
    ProfScope ps(d, 7, st);
    bool direct = false;
    if constexpr (STAGE == 3 && MODE == 1) direct = d->mm_direct;
    if constexpr (STAGE == 3 && MODE == 1) if (direct) {            // all members in one launch (MemberOff): workgroup = nens members x 4 / nens rows of a tile
      const View v = view(d, 0);
      const DyP &p = v.p;
      const MemberOff mo = member_off(d);
      const int U = 64 - 2 * ((d->ord - 1) / 2 + 1), tiles_x = (p.nx + U - 1) / U, rpb = 4 / mo.n;
      const long long waves = (long long)p.ny * tiles_x;
      const int chunk = d->chunk_f ? d->chunk_f : (d->chunk_f = balanced_chunk(d, p.nz, waves, d->o.chunk_f, 10000, 2, 4.5, true));
      dim3 grid((unsigned)(((p.ny + rpb - 1) / rpb) * tiles_x), (unsigned)((p.nz + chunk - 1) / chunk));
#define MW_FUSED_MT(TT) case TT: MW_FUSED_MTK(TT, 0) break;
#define MW_FUSED_MTK(TT, K_) { if (d->ord == 3) MW_FUSED_MTO(TT, K_, 3); else MW_FUSED_MTO(TT, K_, 5); }
#define MW_FUSED_MTO(TT, K_, O_) MW_KLAUNCH((k_tracers_fused<3, 1, TT, true, K_, O_, true>), grid, dim3(256), 0, st, p, S, Sn, Sout, d->FY, d->M[par][0], d->M[par][2], \
                                 d->UP[par][0], d->UP[par][2], d->FX, d->FZ, d->flags, d->dirty + (d->fused_launches & 1), dt, dt_dyn, c, chunk, tiles_x, 0, mo)
      if (marching_config(d, p) == 1) MW_FUSED_MTK(3, 1)
      else switch (p.nt) { MW_FUSED_MT(1) MW_FUSED_MT(2) MW_FUSED_MT(3) MW_FUSED_MT(4) default: MW_FAIL("fused tracer stage needs 1..4 tracers"); }
#undef MW_FUSED_MT
#undef MW_FUSED_MTK
#undef MW_FUSED_MTO
      MW_LAUNCH_CHECK();
    }
    for (int e = 0; e < (direct ? 0 : n_views(d)); e++) {
      const View v = view(d, e);
      const DyP &p = v.p;
      const int U = p.nens == 1 ? 64 - 2 * ((d->ord - 1) / 2 + 1) : 64 - 4 * p.nens;   // hs + 1 / 2 halo cells per side (k_tracers_fused)
      const int tiles_x = (p.nx * p.nens + U - 1) / U;
      const int rows4 = (p.ny >= 4 && d->o.tf_rows4) ? 1 : 0;   // (workgroup = 4 rows of one x tile: the rows' shared y faces meet in L1; option tf_rows4 = 0: 4 x tiles of one row, A/B)
      const long long waves = (long long)p.ny * tiles_x;
      const int chunk = d->chunk_f ? d->chunk_f : (d->chunk_f = balanced_chunk(d, p.nz, waves, d->o.chunk_f, 10000, 2, 4.5, true));
      dim3 grid(rows4 ? (unsigned)(((p.ny + 3) / 4) * tiles_x) : (unsigned)((waves + 3) / 4), (unsigned)((p.nz + chunk - 1) / chunk));
#define MW_FUSED_ARGS d, v, S, Sn, Sout, grid, chunk, tiles_x, par, dt, dt_dyn, c, rows4, st
#define MW_FUSED_CASE(TT) \
      case TT: if (p.nens != 1)     launch_tracers_fused_t<STAGE, MODE, TT, false, 0>(MW_FUSED_ARGS); \
               else if (d->ord == 3) launch_tracers_fused_t<STAGE, MODE, TT, true, 0, 3>(MW_FUSED_ARGS); \
               else                  launch_tracers_fused_t<STAGE, MODE, TT, true, 0>(MW_FUSED_ARGS); break;
      const int K = marching_config(d, p);
      if (K == 1)      { if (d->ord == 3) launch_tracers_fused_t<STAGE, MODE, 3, true, 1, 3>(MW_FUSED_ARGS); else launch_tracers_fused_t<STAGE, MODE, 3, true, 1>(MW_FUSED_ARGS); }
      else if (K == 2) { if (d->ord == 3) launch_tracers_fused_t<STAGE, MODE, 1, true, 2, 3>(MW_FUSED_ARGS); else launch_tracers_fused_t<STAGE, MODE, 1, true, 2>(MW_FUSED_ARGS); }
      else switch (p.nt) { MW_FUSED_CASE(1) MW_FUSED_CASE(2) MW_FUSED_CASE(3) MW_FUSED_CASE(4) default: MW_FAIL("fused tracer stage needs 1..4 tracers"); }
#undef MW_FUSED_CASE
#undef MW_FUSED_ARGS
      MW_LAUNCH_CHECK();
    }
  }
  const DyP &p = d->p;
  if (!p.sim2d && p.pos_mask && !d->o.debug_no_patch) {   // (the switch exists for the negative control in tests/)
    ProfScope ps(d, 1, st);
    for (int e = 0; e < n_views(d); e++) {
      const View v = view(d, e);
      const DyP &q = v.p;
      // (member-major: every member's launch reads the same `dirty` word; only the last one may clear the next stage's word)
      unsigned int *next = (e == n_views(d) - 1) ? d->dirty + ((d->fused_launches + 1) & 1) : d->dirty + 2;
      MW_KLAUNCH((k_tracer_patch<STAGE, MODE>), plane_grid((long long)q.ny * ((q.nx * q.nens + MW_PATCH_CELLS - 1) / MW_PATCH_CELLS), q.nz), dim3(256), 0, st, q,
                         v.S(Sout), d->flags + e * v.cells, d->FX + e * v.f[0], d->FZ + e * v.f[2], dt_dyn, c, d->dirty + (d->fused_launches & 1), next);
      MW_LAUNCH_CHECK();
    }
  } else if (!p.sim2d && p.pos_mask) {                          // (negative-control switch) nobody else clears the next word
    (void)hipMemsetAsync(d->dirty + ((d->fused_launches + 1) & 1), 0, sizeof(unsigned int), st);
  }
  d->fused_launches++;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// One RK stage on the production path, as two pipelines on two HIP streams:
//   state stream  (the handle's stream): halo(state vars) -> k_y_state -> k_xz_state        [fp64-VALU bound]
//   tracer stream (side stream)        : halo(tracers) -> k_y_tracers -> k_tracers_fused -> k_tracer_patch                 [fp64-VALU bound too]
// The state variables of stage s+1 depend only on the state variables of stage s, so the state pipeline runs ahead
// while the tracer pipeline of stage s fills the memory system beside it.  Hand-offs: the tracer kernels need the
// mass fluxes / selectors / new density of their stage (event ev_state); the state pipeline may not run more than
// one stage ahead because the M/UP buffers are double-buffered and the four slabs rotate (event ev_tr of stage s-2).
// ---------------------------------------------------------------------------------------------------------------------
static void zero_rows_stage(mw_dycore_s *d, int stage);
static void zero_rows_conv(mw_dycore_s *d, const double *S, bool done, hipStream_t st);
static void zero_rows_forget(mw_dycore_s *d, const double *S);
static int zero_rows_build(mw_dycore_s *d, const double *S0, const CouplerPtrs &c, bool from_coupler, hipStream_t st, bool first_cycle);
static bool zero_rows_ok(const mw_dycore_s *d);
static int zero_rows_local(mw_dycore_s *d, const double *S0, const CouplerPtrs &c, bool from_coupler, hipStream_t st);
static int zero_rows_merge(mw_dycore_s *d, hipStream_t st, bool first_cycle);
// Option zero_verify (test aid): the maps' claims against the data, on stream `st`, in front of the launches that rely on them.
//   what = 0: in front of the stage's tracer kernel -- the stage's input slab against Qs / QYs, the destination (slab Sout, or the coupler's
//             arrays in the last stage of a time step) against the "holds zeros already" map the kernel was handed;
//   what = 1: in front of the converting y launch -- the slab it fills against zqk.
// Uses the parameter block as the next launch will see it (zero_rows_stage / zero_rows_conv have run).  Counters: mw_debug_zero_violations.
static int zero_rows_verify(mw_dycore_s *d, int what, const double *Sin, const double *Sout, bool dst_coupler, const CouplerPtrs &c, hipStream_t st) {
  if (!d->o.zero_verify || !d->zr_on) return 0;
  if (!d->zviol) { MW_HIP(hipMalloc(&d->zviol, 4 * sizeof(unsigned long long))); MW_HIP(hipMemsetAsync(d->zviol, 0, 4 * sizeof(unsigned long long), st)); }
  for (int e = 0; e < n_views(d); e++) {
    const View v = view(d, e);
    const DyP &p = v.p;
    if (what == 0 && !p.zq) continue;
    if (what == 1 && !p.zqk) continue;
    const unsigned vmask = (marching_config(d, p) == 1) ? 0x6u : 0xFu;
    const long long nrow = (long long)p.nz * p.ny;
    const bool dstc = what == 0 && dst_coupler && p.zqc != nullptr;
    const double *dst = (what == 0 && !dst_coupler && p.zqp) ? v.S(Sout) : nullptr;
    MW_KLAUNCH(k_zero_verify, dim3((unsigned)((nrow + 3) / 4)), dim3(256), 0, st, p, c, what == 0 ? v.S(Sin) : nullptr, dst, dstc ? 1 : 0,
               what == 1 ? v.S(Sin) : nullptr, d->zr_msz, vmask, d->zviol);
    MW_LAUNCH_CHECK();
  }
  return 0;
}
template <int STAGE, int MODE>
static int rk_stage_march(mw_dycore_s *d, double *Sin, const double *Sn, double *Sout, double dt_stage, double dt_dyn,
                          const CouplerPtrs &c) {
  const long long gs = d->gstage++;
  const int par = (int)(gs & 1), slot = (int)(gs & 7);
  hipStream_t ss = d->stream, ts = d->overlap ? d->tstream : d->stream;
  const int T = d->p.nt;
  zero_rows_stage(d, STAGE);
  ProfScope stage_scope(d, 8, ss);                            // one-stream schedule: first launch to last launch of the stage
  if (d->overlap && gs >= 2) MW_HIP(hipStreamWaitEvent(ss, d->ev_tr[(gs - 2) & 7], 0));
  // ---- state pipeline
  if (halo_fill(d, Sin, 0, 5, ss, 0, true)) return 1;
  const bool conv = (STAGE == 1) && d->conv_pending;            // first stage of the step: D1 + D2 inside k_y_state
  d->conv_pending = false;
  // (the converting launch of a member-major handle exists in the members-in-one-workgroup form of the folded configurations only)
  const bool mm_conv_ok = d->mm_direct && d->o.mm_conv && marching_config(d, view(d, 0).p) != 0;
  const bool yall = y_all_ok(d) && !(conv && ((d->member_major && !mm_conv_ok) || !d->o.y_all_conv));   // y faces of state variables and tracers in one launch
  if (STAGE == 1) { if (conv) zero_rows_conv(d, Sin, false, ss); else zero_rows_forget(d, Sin); }   // (what is known about the rows of the slab that is about to be written)
  if (STAGE == 3 && MODE == 0) zero_rows_forget(d, Sout);
  {
  if (conv && zero_rows_verify(d, 1, Sin, nullptr, false, c, ss)) return 1;
  if (yall) { if (halo_fill(d, Sin, 5, T, ts, 1, true) || launch_y_all(d, Sin, conv ? &c : nullptr)) return 1; }
  else if (launch_y_state(d, Sin, par, conv ? &c : nullptr)) return 1;             // y faces: m_upw, selector, y tendencies
  if (STAGE == 1 && conv) zero_rows_conv(d, Sin, true, ss);
  if (launch_xz_state<STAGE, MODE>(d, Sin, Sn, Sout, dt_stage, dt_dyn, par, c)) return 1;   // x,z faces + finished state variables
  }
  if (STAGE == 1 && d->member_major && !d->overlap) {           // the members' maps, from the slab the y launch has just completed
    if (zero_rows_build(d, Sin, c, false, ss, d->first_cycle)) return 1;
    zero_rows_stage(d, 1);
  }
  // ---- tracer pipeline.  Its halo fill (and, on several ranks, its strip exchange over RCCL) only needs the tracer values of the
  // previous stage, which this stream produced itself: it is issued BEFORE the wait for this stage's state kernels and so
  // runs beside them; the state stream's exchange for stage s+1 in turn runs beside this stage's tracer kernels.
  // (Measured, round 3 -- profiles/r03_ab_two_stream_yt_beside_xz.txt: letting k_y_tracers start right behind k_y_state, BESIDE
  //  k_xz_state (an HBM-bound launch beside a VALU-bound one), stretches both and leaves the step where it was: 5.62-5.74 ms against
  //  5.51-5.70 on one stream.  The step as a whole moves 26 GB at 4.8 TB/s: there is no idle HBM time for a second kernel to use.)
  if (!yall && halo_fill(d, Sin, 5, T, ts, 1, true)) return 1;
  if (d->overlap) { MW_HIP(hipEventRecord(d->ev_state[slot], ss)); MW_HIP(hipStreamWaitEvent(ts, d->ev_state[slot], 0)); }
  if (!yall && launch_y_tracers(d, Sin, par, ts)) return 1;                   // tracer fluxes (public arrays)
  if (d->fused) {
    if (zero_rows_verify(d, 0, Sin, Sout, MODE == 1, c, ts)) return 1;
    if (launch_tracers_fused<STAGE, MODE>(d, Sin, Sn, Sout, par, dt_stage, dt_dyn, c, ts)) return 1;   // x/z fluxes + D10 + D11/D12 (+ D13)
  } else {
    if (launch_xz_tracers(d, Sin, par, dt_stage, ts)) return 1;               // x/z fluxes + D10 (FCT)
    if (launch_tracer_update<STAGE, MODE>(d, Sin, Sn, Sout, dt_dyn, c, ts)) return 1;
  }
  if (d->overlap) MW_HIP(hipEventRecord(d->ev_tr[slot], ts));
  return 0;
}
// ---------------------------------------------------------------------------------------------------------------------
// One RK stage of a block of a decomposed domain, PIPELINED schedule (the default with a neighbour exchange when k_y_all applies):
// one compute stream, and the strip exchange of a stage on the side stream BESIDE interior work that does not need it:
//   compute : k_y_all(chunks without halo rows) | wait | k_y_all(first + last chunk) -> k_xz_state -> k_tracers_fused (+ patch)
//   exchange:   [strips of this stage's input ]         after k_xz_state: state strips of the NEXT stage's input (beside
//                                                       k_tracers_fused); after k_tracers_fused: its tracer strips (beside the next
//                                                       stage's interior k_y_all)
// The y marching kernel reads no x halo at all and y halo rows only in its first and last chunk, so six of eight chunks start at
// once.  The first stage of a cycle exchanges all variables at its start (its input comes from the conversion pass / the previous
// cycle).  Compared with the two-stream schedule of rk_stage_march (each pipeline hides the other's exchange behind whole kernels)
// this one keeps k_y_all -- 5 % of the step -- and needs less machinery; the transfer must fit beside ~0.3-0.5 ms of kernels.
// ---------------------------------------------------------------------------------------------------------------------
template <int STAGE, int MODE>
static int rk_stage_pipe(mw_dycore_s *d, double *Sin, const double *Sn, double *Sout, double dt_stage, double dt_dyn, const CouplerPtrs &c) {
  const long long gs = d->gstage++;
  const int par = (int)(gs & 1);
  hipStream_t ss = d->stream, xs = d->tstream;
  const int T = d->p.nt;
  ProfScope stage_scope(d, 8, ss);
  const bool conv = (STAGE == 1) && d->conv_pending;            // the inner rows come from the coupler's arrays (see time_step)
  d->conv_pending = false;
  // (round 4: the two edge strips of the y launch run on the EXCHANGE stream right behind the unpack kernels -- beside the inner rows on
  //  the compute stream -- instead of behind them: a launch of 2 x 157 workgroups no longer sits alone between k_y_all and k_xz_state)
  const bool edge_side = !d->o.pipe_edge_inline;
  // (round 5: the edge strips of the NEXT stage's y launch are SPLIT by what they wait for.  Their state part -- y tendencies of the edge
  //  rows, k_y_state -- only needs the state strips, which travel beside this stage's tracer kernel: it runs right behind them, and
  //  k_xz_state of the next stage waits for nothing else.  Their tracer part -- the tracer y fluxes of the edge faces, k_y_tracers, which
  //  only the next stage's TRACER kernel reads -- runs behind the tracer strips and has the next stage's inner y rows AND its k_xz_state
  //  to hide behind.  Before, k_xz_state waited for the whole tracer chain (pack, group, unpack, edge launch: 0.3-0.4 ms of idle compute
  //  stream per stage in the rocprofv3 timeline of the self-loop transport).  pipe_split_edges = 0: one k_y_all edge launch
  //  behind the tracer strips, as in rounds 3-4.)
  const bool split_edges = edge_side && d->o.pipe_split_edges;
  const int par_next = (int)((gs + 1) & 1);
  // (round 5, first stage with the split edge strips: LOCAL zero-row maps on the compute stream in front of the y launches, the state strips
  //  and the tracer strips as two exchanges -- k_xz_state waits for the first only -- and the neighbours' maps merged in behind them; before,
  //  this stage's y launch ran without maps, 664 against 476 us, in front of one 609 us exchange chain for all eight variables)
  bool maps_early = false;
  // (every rank must take the same branch here -- it posts a different number and size of exchanges -- so the size test looks at the
  //  SMALLEST block of the decomposition, as zero_rows_ok does: blocks of 15 and 16 rows (ny_glob = 31 on two y ranks) would otherwise
  //  straddle the threshold and post mismatched send / receive groups)
  const long long ny_min_blk = d->g.ny_glob / std::max(1, d->p.nproc_y);
  if (!d->pipe_ready && STAGE == 1 && split_edges && d->o.pipe_maps_early && !d->p.wrap_y && ny_min_blk >= 4 * MW_Y_EDGE && zero_rows_ok(d)) {   // (a y-decomposed block with real edge strips)
    maps_early = true;
    // the local maps on the exchange stream: from the coupler's arrays they only need the step's inputs and run BESIDE the strip conversion
    // on the compute stream (first sub-cycle); from the slab they wait for it like everything else
    const bool beside = conv && d->first_cycle && d->entry_marked;
    MW_HIP(hipEventRecord(d->ev_pipe[0], ss));
    MW_HIP(hipStreamWaitEvent(xs, beside ? d->ev_pipe[6] : d->ev_pipe[0], 0));
    if (zero_rows_local(d, Sin, c, conv, xs)) return 1;
    MW_HIP(hipEventRecord(d->ev_pipe[5], xs));
    MW_HIP(hipStreamWaitEvent(ss, d->ev_pipe[5], 0));         // the stage's inner y rows read them
    if (beside) MW_HIP(hipStreamWaitEvent(xs, d->ev_pipe[0], 0));
    if (halo_fill(d, Sin, 0, 5, xs, 0, true)) return 1;
    if (launch_y_state(d, Sin, par, nullptr, true, xs)) return 1;
    MW_HIP(hipEventRecord(d->ev_pipe[2], xs));
    if (halo_fill(d, Sin, 5, T, xs, 1, true)) return 1;
    if (launch_y_tracers(d, Sin, par, xs, true)) return 1;
    MW_HIP(hipEventRecord(d->ev_pipe[3], xs));
    d->pipe_edge_done = true;
    if (zero_rows_merge(d, xs, d->first_cycle)) return 1;
    if (d->zr_on) MW_HIP(hipEventRecord(d->ev_pipe[4], xs));
    zero_rows_stage(d, 1);                                      // the local maps, for this stage's inner y rows
    if (conv) zero_rows_conv(d, Sin, false, ss); else zero_rows_forget(d, Sin);
  } else
  if (!d->pipe_ready) {                                       // this stage's input has not been exchanged yet
    MW_HIP(hipEventRecord(d->ev_pipe[0], ss)); MW_HIP(hipStreamWaitEvent(xs, d->ev_pipe[0], 0));
    if (halo_fill(d, Sin, 0, -1, xs, 0, true)) return 1;
    if (edge_side && launch_y_all(d, Sin, nullptr, 2, xs)) return 1;
    MW_HIP(hipEventRecord(d->ev_pipe[2], xs));
    MW_HIP(hipEventRecord(d->ev_pipe[3], xs));
    d->pipe_edge_done = edge_side;
    if (STAGE == 1) {                                           // the sub-cycle's zero-row maps, behind the strips: needed by the tracer kernel only
      if (zero_rows_build(d, Sin, c, conv, xs, d->first_cycle)) return 1;
      if (conv) zero_rows_conv(d, Sin, true, xs); else zero_rows_forget(d, Sin);
      if (d->zr_on) MW_HIP(hipEventRecord(d->ev_pipe[4], xs));
    }
  }
  if (STAGE != 1) zero_rows_stage(d, STAGE);                    // (stage 1 without the early maps: its y launches run BESIDE the map build -- the maps are handed over in front of the tracer kernel)
  if (STAGE == 3 && MODE == 0) zero_rows_forget(d, Sout);
  d->pipe_ready = false;
  if (conv && zero_rows_verify(d, 1, Sin, nullptr, false, c, ss)) return 1;
  if (launch_y_all(d, Sin, conv ? &c : nullptr, 1)) return 1;  // rows whose chunks read no halo row
  { ProfScope wait_scope(d, 10, ss);                           // (profile class 10: how long the compute stream sits in this wait)
    MW_HIP(hipStreamWaitEvent(ss, d->ev_pipe[2], 0)); }        // state strips (+ the edge rows' y tendencies) of this stage's input
  if (!d->pipe_edge_done && launch_y_all(d, Sin, nullptr, 2)) return 1;   // first and last chunk
  d->pipe_edge_done = false;
  if (launch_xz_state<STAGE, MODE>(d, Sin, Sn, Sout, dt_stage, dt_dyn, par, c)) return 1;
  const bool early = (STAGE < 3);                             // the next stage of this cycle reads Sout
  if (early) {
    MW_HIP(hipEventRecord(d->ev_pipe[0], ss)); MW_HIP(hipStreamWaitEvent(xs, d->ev_pipe[0], 0));
    if (halo_fill(d, Sout, 0, 5, xs, 0, true)) return 1;      // state strips, beside the tracer stage
    if (split_edges) {
      if (launch_y_state(d, Sout, par_next, nullptr, true, xs)) return 1;   // ... and the state part of the NEXT stage's edge strips right behind them
      MW_HIP(hipEventRecord(d->ev_pipe[2], xs));
    }
  }
  { ProfScope wait_scope(d, 11, ss);                           // (profile class 11)
    MW_HIP(hipStreamWaitEvent(ss, d->ev_pipe[3], 0)); }        // tracer strips + the edge faces' tracer fluxes of this stage's input
  if (STAGE == 1 && d->zr_on) {
    MW_HIP(hipStreamWaitEvent(ss, d->ev_pipe[4], 0)); zero_rows_stage(d, 1);
    if (maps_early && conv) zero_rows_conv(d, Sin, true, ss);     // (the slab's row map changes BEHIND the converting launch that reads it)
  }
  if (zero_rows_verify(d, 0, Sin, Sout, MODE == 1, c, ss)) return 1;
  if (launch_tracers_fused<STAGE, MODE>(d, Sin, Sn, Sout, par, dt_stage, dt_dyn, c, ss)) return 1;
  if (early) {
    MW_HIP(hipEventRecord(d->ev_pipe[1], ss)); MW_HIP(hipStreamWaitEvent(xs, d->ev_pipe[1], 0));
    if (halo_fill(d, Sout, 5, T, xs, 1, true)) return 1;      // tracer strips, beside the next stage's interior y chunks (and, split, its k_xz_state)
    if (split_edges) { if (launch_y_tracers(d, Sout, par_next, xs, true)) return 1; }                    // the tracer part of the next stage's edge strips
    else if (edge_side) { zero_rows_stage(d, STAGE + 1); if (launch_y_all(d, Sout, nullptr, 2, xs)) return 1; }   // ... or both parts in one launch (the NEXT stage's maps)
    if (!split_edges) MW_HIP(hipEventRecord(d->ev_pipe[2], xs));
    MW_HIP(hipEventRecord(d->ev_pipe[3], xs));
    d->pipe_ready = true; d->pipe_edge_done = edge_side;
  }
  return 0;
}
// Zero-row maps (mw_march.h: k_zero_rows).  Which handles: nens == 1, fused tracer stage, x and y periodic; one rank on the one-stream
// schedule, or the blocks of a decomposed domain on the pipelined schedule.  The ranks of a decomposed domain exchange maps, so all of
// them must decide alike: the size test looks at the smallest block of the decomposition, not at this rank's.
static bool zero_rows_ok(const mw_dycore_s *d) {
  const DyP &p = d->p;
  // (nens > 1: the member-major layout on one rank -- every member has its own maps and the per-member launches read them; the launches
  //  that hold all members of a tile in one workgroup run without)
  if (!(d->o.zero_skip && d->o.zero_rows && d->fused && (p.nens == 1 || (d->member_major && !d->xchg)) && p.nt >= 1 && p.nt <= 4 && !p.sim2d &&
        p.nz >= 2 && p.bc_x == MW_BC_PERIODIC && p.bc_y == MW_BC_PERIODIC)) return false;
  const long long nx_min = d->g.nx_glob / std::max(1, p.nproc_x), ny_min = d->g.ny_glob / std::max(1, p.nproc_y);
  if (ny_min < MW_ZR_HALO || nx_min < 2 * MW_ZR_HALO) return false;   // (a tracer must not cross a whole block in one sub-cycle)
  if (!d->xchg) return !d->overlap && !d->pipe;
  return d->pipe != 0;
}
// ... built at the start of a sub-cycle from its input on stream `st`: the coupler's arrays while the conversion is still pending (it
// happens inside the first y launch), slab S0 otherwise.  Blocks of a decomposed domain: + the neighbours' maps (see k_zero_merge).
static int zero_rows_build(mw_dycore_s *d, const double *S0, const CouplerPtrs &c, bool from_coupler, hipStream_t st, bool first_cycle) {
  d->zr_on = zero_rows_ok(d);
  if (!d->zr_on) return 0;
  DyP &p = d->p;
  const int ld = p.ny + 2 * MW_ZR_HALO;
  const long long msz = (long long)p.nz * ld;
  const bool ex_x = d->xchg && p.nproc_x > 1, ex_y = d->xchg && p.nproc_y > 1;
  const long long nrow = (long long)p.nz * p.ny, nedge = (long long)p.nz * MW_ZR_HALO;
  const long long dWE = (nrow + 1) / 2, dSN = (nedge + 1) / 2;   // message lengths in doubles (the transport's unit)
  if (!d->zr || d->zr_msz != msz) {
    if (d->zr) { MW_HIP(hipDeviceSynchronize()); (void)hipFree(d->zr); d->zr = nullptr; }
    if (d->zrx) { (void)hipFree(d->zrx); d->zrx = nullptr; }
    if (hipMalloc(&d->zr, (2 * MW_ZR_MAPS + 3) * (size_t)msz * sizeof(unsigned) * (size_t)p.nens) != hipSuccess) {   // per member: two sets + MC + the two q^n slabs' maps
      (void)hipGetLastError(); d->zr = nullptr; d->zr_on = false;
      if (d->xchg) MW_FAIL("zero-row maps: out of device memory");   // (a decomposed block: the other ranks are about to exchange maps -- an error, not a fall-back)
      return 0; }
    d->zr_msz = msz; d->zr_prev_ok = false; d->kz_buf[0] = d->kz_buf[1] = nullptr;
  }
  if (d->member_major) {                                        // one map set per member, from the member's slab (the caller runs this behind the first y launch)
    d->zr_cur ^= 1;
    d->zr_prev_use = d->zr_prev_ok && d->o.zero_stores;
    const long long mstride = (2 * MW_ZR_MAPS + 3) * msz;
    const int K = marching_config(d, view(d, 0).p);
    const unsigned vmask = (K == 1) ? 0x6u : 0xFu;
    ProfScope ps(d, 4, st);
    for (int e = 0; e < n_views(d); e++) {
      const View v = view(d, e);
      DyP q = v.p; q.zq_ld = ld;
      unsigned *zr = d->zr + e * mstride + (long long)d->zr_cur * MW_ZR_MAPS * msz;
      MW_KLAUNCH(k_zero_rows<true>, dim3((unsigned)((nrow + 3) / 4)), dim3(256), 0, st, q, c, v.S(S0), zr, vmask, ld, MW_ZR_HALO, 1, nullptr);
      MW_KLAUNCH(k_zero_dilate, dim3((unsigned)((p.ny + 63) / 64), (unsigned)((p.nz + 15) / 16)), dim3(256), 0, st, q, zr, msz, 0);
    }
    MW_LAUNCH_CHECK();
    return 0;
  }
  d->zr_cur ^= 1;                                               // build into the other set; the one of the sub-cycle before stays readable
  d->zr_prev_use = d->zr_prev_ok && d->o.zero_stores;
  unsigned *const zr = d->zr + (long long)d->zr_cur * MW_ZR_MAPS * msz;
  if ((ex_x || ex_y) && !d->zrx) {
    if (hipMalloc(&d->zrx, (size_t)(3 * dWE + 4 * dSN) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); MW_FAIL("zero-row maps: out of device memory"); }
    // (a failure here is an error, not a fall-back: the other ranks are about to exchange maps)
  }
  p.zq_ld = ld;
  const int K = marching_config(d, p);
  const unsigned vmask = (K == 1) ? 0x6u : 0xFu;                // = tracer_may_vanish<K>
  ProfScope ps(d, 4, st);
  const dim3 g((unsigned)((nrow + 3) / 4));
  const bool local = !ex_x && !ex_y;
  unsigned *own = local ? zr : (unsigned *)d->zrx;
  const int ldo = local ? ld : p.ny, offo = local ? MW_ZR_HALO : 0;
  if (from_coupler) MW_KLAUNCH(k_zero_rows<false>, g, dim3(256), 0, st, p, c, S0, own, vmask, ldo, offo, local ? 1 : 0, nullptr);
  else              MW_KLAUNCH(k_zero_rows<true>, g, dim3(256), 0, st, p, c, S0, own, vmask, ldo, offo, local ? 1 : 0, nullptr);
  MW_LAUNCH_CHECK();
  if (!local) {
    double *rW = d->zrx + dWE, *rE = d->zrx + 2 * dWE, *sS = d->zrx + 3 * dWE, *sN = sS + dSN, *rS = sN + dSN, *rN = rS + dSN;
    if (ex_x && d->xchg(d->xchg_ctx, d->zrx, d->zrx, nullptr, nullptr, rW, rE, nullptr, nullptr, dWE, 0, st)) MW_FAIL("zero-row maps: exchange callback failed");
    MW_KLAUNCH(k_zero_merge, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, st, p, own, ex_x ? (const unsigned *)rW : nullptr, (const unsigned *)rE, zr,
               ex_y ? (unsigned *)sS : nullptr, (unsigned *)sN);
    MW_LAUNCH_CHECK();
    if (ex_y) {
      if (d->xchg(d->xchg_ctx, nullptr, nullptr, sS, sN, nullptr, nullptr, rS, rN, 0, dSN, st)) MW_FAIL("zero-row maps: exchange callback failed");
      MW_KLAUNCH(k_zero_halo, dim3((unsigned)((nedge + 255) / 256)), dim3(256), 0, st, p, zr, (const unsigned *)rS, (const unsigned *)rN);
      MW_LAUNCH_CHECK();
    }
  }
  // the first sub-cycle's M0 doubles as "which rows of the coupler's tracer arrays are zero" until the last sub-cycle's D13 (map MC)
  if (first_cycle) MW_HIP(hipMemcpyAsync(d->zr + 2 * MW_ZR_MAPS * msz, zr, (size_t)msz * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
  MW_KLAUNCH(k_zero_dilate, dim3((unsigned)((p.ny + 63) / 64), (unsigned)((p.nz + 15) / 16)), dim3(256), 0, st, p, zr, msz, 0);
  MW_LAUNCH_CHECK();
  return 0;
}
// The same in two halves for the first stage of the pipelined schedule (round 5, profiles/r05_selfloop_timeline_maps.txt): LOCAL maps on the
// compute stream in front of the stage's y launches -- the y kernel reads no x halo, its inner rows only the block's own rows; the rows
// beyond a decomposed y edge count as "may be non-zero", and FNs as "store" -- ...
static int zero_rows_local(mw_dycore_s *d, const double *S0, const CouplerPtrs &c, bool from_coupler, hipStream_t st) {
  d->zr_on = zero_rows_ok(d);
  if (!d->zr_on) return 0;
  DyP &p = d->p;
  const int ld = p.ny + 2 * MW_ZR_HALO;
  const long long msz = (long long)p.nz * ld;
  const bool ex_x = d->xchg && p.nproc_x > 1, ex_y = d->xchg && p.nproc_y > 1;
  const long long nrow = (long long)p.nz * p.ny, nedge = (long long)p.nz * MW_ZR_HALO;
  const long long dWE = (nrow + 1) / 2, dSN = (nedge + 1) / 2;
  if (!d->zr || d->zr_msz != msz) {
    if (d->zr) { MW_HIP(hipDeviceSynchronize()); (void)hipFree(d->zr); d->zr = nullptr; }
    if (d->zrx) { (void)hipFree(d->zrx); d->zrx = nullptr; }
    if (hipMalloc(&d->zr, (2 * MW_ZR_MAPS + 3) * (size_t)msz * sizeof(unsigned) * (size_t)p.nens) != hipSuccess) {
      (void)hipGetLastError(); d->zr = nullptr; d->zr_on = false;
      if (d->xchg) MW_FAIL("zero-row maps: out of device memory");   // (as in zero_rows_build: the neighbours still post their map exchanges)
      return 0; }
    d->zr_msz = msz; d->zr_prev_ok = false; d->kz_buf[0] = d->kz_buf[1] = nullptr;
  }
  if (!d->zrx && hipMalloc(&d->zrx, (size_t)(3 * dWE + 4 * dSN) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); MW_FAIL("zero-row maps: out of device memory"); }
  d->zr_cur ^= 1;
  d->zr_prev_use = d->zr_prev_ok && d->o.zero_stores;
  unsigned *const zr = d->zr + (long long)d->zr_cur * MW_ZR_MAPS * msz;
  p.zq_ld = ld;
  const unsigned vmask = (marching_config(d, p) == 1) ? 0x6u : 0xFu;
  ProfScope ps(d, 4, st);
  const dim3 g((unsigned)((nrow + 3) / 4));
  const int wrap = ex_y ? 2 : 1;
  if (from_coupler) MW_KLAUNCH(k_zero_rows<false>, g, dim3(256), 0, st, p, c, S0, zr, vmask, ld, MW_ZR_HALO, wrap, (unsigned *)d->zrx);
  else              MW_KLAUNCH(k_zero_rows<true>, g, dim3(256), 0, st, p, c, S0, zr, vmask, ld, MW_ZR_HALO, wrap, (unsigned *)d->zrx);
  MW_KLAUNCH(k_zero_dilate, dim3((unsigned)((p.ny + 63) / 64), (unsigned)((p.nz + 15) / 16)), dim3(256), 0, st, p, zr, msz, (ex_x || ex_y) ? 1 : 0);
  MW_LAUNCH_CHECK();
  return 0;
}
// ... and the neighbours' maps merged in on the exchange stream, for the tracer kernel and everything after it.  (The y launches of the first
// stage may read either version of a word while this runs: both describe their rows correctly, the merged one only knows more.)
static int zero_rows_merge(mw_dycore_s *d, hipStream_t st, bool first_cycle) {
  if (!d->zr_on) return 0;
  DyP &p = d->p;
  const long long msz = d->zr_msz;
  const bool ex_x = d->xchg && p.nproc_x > 1, ex_y = d->xchg && p.nproc_y > 1;
  const long long nrow = (long long)p.nz * p.ny, nedge = (long long)p.nz * MW_ZR_HALO;
  const long long dWE = (nrow + 1) / 2, dSN = (nedge + 1) / 2;
  unsigned *const zr = d->zr + (long long)d->zr_cur * MW_ZR_MAPS * msz;
  ProfScope ps(d, 4, st);
  if (ex_x || ex_y) {
    double *rW = d->zrx + dWE, *rE = d->zrx + 2 * dWE, *sS = d->zrx + 3 * dWE, *sN = sS + dSN, *rS = sN + dSN, *rN = rS + dSN;
    if (ex_x && d->xchg(d->xchg_ctx, d->zrx, d->zrx, nullptr, nullptr, rW, rE, nullptr, nullptr, dWE, 0, st)) MW_FAIL("zero-row maps: exchange callback failed");
    MW_KLAUNCH(k_zero_merge, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, st, p, (const unsigned *)d->zrx, ex_x ? (const unsigned *)rW : nullptr, (const unsigned *)rE, zr,
               ex_y ? (unsigned *)sS : nullptr, (unsigned *)sN);
    MW_LAUNCH_CHECK();
    if (ex_y) {
      if (d->xchg(d->xchg_ctx, nullptr, nullptr, sS, sN, nullptr, nullptr, rS, rN, 0, dSN, st)) MW_FAIL("zero-row maps: exchange callback failed");
      MW_KLAUNCH(k_zero_halo, dim3((unsigned)((nedge + 255) / 256)), dim3(256), 0, st, p, zr, (const unsigned *)rS, (const unsigned *)rN);
      MW_LAUNCH_CHECK();
    }
  }
  if (first_cycle) MW_HIP(hipMemcpyAsync(d->zr + 2 * MW_ZR_MAPS * msz, zr, (size_t)msz * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
  if (ex_x || ex_y) { MW_KLAUNCH(k_zero_dilate, dim3((unsigned)((p.ny + 63) / 64), (unsigned)((p.nz + 15) / 16)), dim3(256), 0, st, p, zr, msz, 0); MW_LAUNCH_CHECK(); }
  return 0;
}
// ... and handed to the kernels of RK stage `stage` (1..3) through the parameter block
static void zero_rows_stage(mw_dycore_s *d, int stage) {
  DyP &p = d->p;
  p.zqk = nullptr;
  if (!d->zr_on || stage < 1) { p.zq = p.zqp = p.zqc = nullptr; p.zq_ld = 0; return; }
  const long long set = (long long)MW_ZR_MAPS * d->zr_msz;
  p.zq = d->zr + d->zr_cur * set + (long long)stage * d->zr_msz;
  p.zqp = (d->zr_prev_use && stage <= 2) ? d->zr + (d->zr_cur ^ 1) * set + (long long)stage * d->zr_msz : nullptr;   // (S1, S2: the slab of stage s is always the same one)
  p.zqc = (d->o.zero_stores && !d->member_major) ? d->zr + 2 * set : nullptr;      // (member-major: the coupler's arrays are written by launches that read no maps)
  // (member-major: these are member 0's; view() moves them on to its member)
  p.zq_ld = p.ny + 2 * MW_ZR_HALO;
}
// The converting y launch (first stage of a time step, conversion inside k_y_all) writes slab S: before it, hand over what is known about S's
// rows (written by the last conversion into S, untouched since); after it (`done`), S's rows are zero exactly where the coupler's are: map MC.
static void zero_rows_conv(mw_dycore_s *d, const double *S, bool done, hipStream_t st) {
  const long long msz = d->zr_msz;
  int sl = (d->kz_buf[0] == S) ? 0 : (d->kz_buf[1] == S) ? 1 : -1;
  if (!done) { d->p.zqk = (sl >= 0 && d->zr_on && d->o.zero_stores) ? d->zr + (2 * MW_ZR_MAPS + 1 + sl) * msz : nullptr; return; }
  d->p.zqk = nullptr;
  if (!d->zr_on) { if (sl >= 0) d->kz_buf[sl] = nullptr; return; }
  if (sl < 0) {                                                 // a free slot, else the one of a slab that is not one of the two q^n slabs any more
    const double *other = (S == d->S0) ? d->S3 : d->S0;
    sl = (d->kz_buf[0] == nullptr) ? 0 : (d->kz_buf[1] == nullptr) ? 1 : (d->kz_buf[0] != other) ? 0 : 1;
  }
  (void)hipMemcpyAsync(d->zr + (2 * MW_ZR_MAPS + 1 + sl) * msz, d->zr + 2 * MW_ZR_MAPS * msz, (size_t)msz * sizeof(unsigned), hipMemcpyDeviceToDevice, st);
  d->kz_buf[sl] = S;
}
static void zero_rows_forget(mw_dycore_s *d, const double *S) {   // slab S is about to be written by something that keeps no map
  for (int i = 0; i < 2; i++) if (!S || d->kz_buf[i] == S) d->kz_buf[i] = nullptr;
}
// One SSPRK3 sub-cycle.  Slabs: Q[0] = q^n, Q[1..3] scratch; on return the new q^n is in Q[3] (caller rotates).
static int rk_cycle_march(mw_dycore_s *d, double **Q, double dt_dyn, bool last, const CouplerPtrs &c) {
  const double dt2 = (1.0 / 4.0) * dt_dyn, dt3 = (2.0 / 3.0) * dt_dyn;
  d->zr_on = false;
  if (!d->pipe && !d->member_major && zero_rows_build(d, Q[0], c, d->conv_pending, d->stream, d->first_cycle)) return 1;   // (pipelined schedule: inside its first stage, on the exchange stream; member-major: behind the first y launch, from the slab)
  if (d->pipe) {                                              // blocks of a decomposed domain, pipelined schedule
    d->pipe_ready = false; d->pipe_edge_done = false;
    if (rk_stage_pipe<1, 0>(d, Q[0], Q[0], Q[1], dt_dyn, dt_dyn, c)) return 1;
    if (rk_stage_pipe<2, 0>(d, Q[1], Q[0], Q[2], dt2, dt_dyn, c)) return 1;
    const bool pass13p = d->member_major && !d->mm_direct;
    if (last && !pass13p) { if (rk_stage_pipe<3, 1>(d, Q[2], Q[0], Q[3], dt3, dt_dyn, c)) return 1; }
    else                  { if (rk_stage_pipe<3, 0>(d, Q[2], Q[0], Q[3], dt3, dt_dyn, c)) return 1; }
    if (last && pass13p) {
      ProfScope ps(d, 4, d->stream);
      const View v = view(d, 0);
      const MemberStrides ms = {v.p.sJ, v.p.sK, v.p.sV, v.slab};
      MW_KLAUNCH(k_member_to_coupler, plane_grid((long long)d->p.ny * d->p.nx * d->p.nens, d->p.nz), dim3(256), 0, d->stream, d->p, Q[3], c, ms);
      MW_LAUNCH_CHECK();
    }
    d->flux_src = Q[2]; d->flux_dt = dt3;
    d->zr_prev_ok = d->zr_on; d->zr_on = false; zero_rows_stage(d, 0);
    return 0;
  }
  if (rk_stage_march<1, 0>(d, Q[0], Q[0], Q[1], dt_dyn, dt_dyn, c)) return 1;                        // stage 1 (:119-132)
  if (rk_stage_march<2, 0>(d, Q[1], Q[0], Q[2], dt2, dt_dyn, c)) return 1;                           // stage 2 (:136-153)
  const bool pass13 = d->member_major && !d->mm_direct;        // D13 as a pass over the result slab
  if (last && !pass13) { if (rk_stage_march<3, 1>(d, Q[2], Q[0], Q[3], dt3, dt_dyn, c)) return 1; }   // stage 3 (:157-174) + :178
  else                 { if (rk_stage_march<3, 0>(d, Q[2], Q[0], Q[3], dt3, dt_dyn, c)) return 1; }
  if (last && pass13) {                              // D13 (:178) as one coalesced pass over the result slab
    hipStream_t ts = d->overlap ? d->tstream : d->stream;     // the tracer pipeline finishes the stage
    ProfScope ps(d, 4, ts);
    const View v = view(d, 0);
    const MemberStrides ms = {v.p.sJ, v.p.sK, v.p.sV, v.slab};
    MW_KLAUNCH(k_member_to_coupler, plane_grid((long long)d->p.ny * d->p.nx * d->p.nens, d->p.nz), dim3(256), 0, ts, d->p, Q[3], c, ms);
    MW_LAUNCH_CHECK();
    if (d->overlap) MW_HIP(hipEventRecord(d->ev_tr[(d->gstage - 1) & 7], ts));     // the step's join waits for this event
  }
  d->flux_src = Q[2]; d->flux_dt = dt3;
  d->zr_prev_ok = d->zr_on; d->zr_on = false; zero_rows_stage(d, 0);
  return 0;
}

static int make_coupler_ptrs(mw_dycore_s *d, const double *rho_d, const double *u, const double *v, const double *w,
                             const double *temp, double *const *tracers, CouplerPtrs &c) {
  if (!rho_d || !u || !v || !w || !temp || (d->g.num_tracers > 0 && !tracers)) MW_FAIL("null field pointer");
  c.rho_d = (double *)rho_d; c.u = (double *)u; c.v = (double *)v; c.w = (double *)w; c.temp = (double *)temp;
  for (int t = 0; t < MW_MAX_TRACERS; t++) c.tr[t] = (t < d->g.num_tracers) ? tracers[t] : nullptr;
  for (int t = 0; t < d->g.num_tracers; t++) if (!c.tr[t]) MW_FAIL("null tracer pointer");
  return 0;
}

// A block of a decomposed domain needs its neighbours' strips: without a transport the halo fill would wrap the block onto
// itself and the result would be silently wrong (the reference always exchanges, :641-723).
static int need_exchange(const mw_dycore_s *d) {
  const mw_grid_t &g = d->g;
  if (!d->xchg && (g.nproc_x > 1 || (g.nproc_y > 1 && g.ny_glob != 1)))
    MW_FAIL("this handle is one block of a " + std::to_string(g.nproc_x) + " x " + std::to_string(g.nproc_y) +
            " rank grid but no halo-exchange transport is installed (mw_dycore_use_rccl / mw_dycore_set_exchange)");
  return 0;
}

static int validate_grid(const mw_grid_t *g) {
  if (!g) MW_FAIL("null grid");
  if (g->nz < 3 || g->nx < 3 || g->ny < 1 || g->nens < 1) MW_FAIL("grid too small (need nz,nx >= 3, ny >= 1, nens >= 1)");
  if (g->ny_glob != 1 && g->ny < 3) MW_FAIL("3-D runs need ny >= 3 per rank");
  if (g->num_tracers < 1 || g->num_tracers > MW_MAX_TRACERS) MW_FAIL("num_tracers must be in [1, MW_MAX_TRACERS] (a water_vapor tracer is required, SURVEY 8(a) quirk 6)");
  if (g->idWV < 0 || g->idWV >= g->num_tracers) MW_FAIL("idWV out of range");
  for (int b : {g->bc_x, g->bc_y, g->bc_z}) if (b != MW_BC_PERIODIC && b != MW_BC_OPEN && b != MW_BC_WALL) MW_FAIL("bc_x / bc_y / bc_z must be 0 (periodic), 1 (open) or 2 (wall)");
  return 0;
}

// The halo kernels fill a periodic halo from the interior of the SAME block (halo_x_body: src = ih + nx) and the pack kernels read
// HX / HY interior cells per side: a block narrower than its halo would read halo cells that are not filled yet and give silently
// wrong results.  (3 cells up to WENO-5; 4 / 5 for orders 7 / 9; z only when bc_z is periodic -- wall / open copy one level.)
static int check_halo_fit(const mw_dycore_s *d) {
  const DyP &p = d->p;
  if (p.nx < p.HX) MW_FAIL("nx = " + std::to_string(p.nx) + " per rank is narrower than the x halo of WENO order " + std::to_string(d->ord) + " (" + std::to_string(p.HX) + " cells)");
  if (!p.sim2d && p.ny < p.HY) MW_FAIL("ny = " + std::to_string(p.ny) + " per rank is narrower than the y halo of WENO order " + std::to_string(d->ord) + " (" + std::to_string(p.HY) + " cells)");
  if (p.bc_z == MW_BC_PERIODIC && p.nz < p.HZ) MW_FAIL("nz = " + std::to_string(p.nz) + " is smaller than the z halo of WENO order " + std::to_string(d->ord) + " (" + std::to_string(p.HZ) + " levels) with bc_z = periodic");
  return 0;
}

extern "C" {

int mw_dycore_create(mw_dycore_t *h, const mw_grid_t *g, const unsigned char *tracer_positive,
                     const unsigned char *tracer_adds_mass, void *stream) {
  if (!h) MW_FAIL("null handle pointer");
  if (validate_grid(g)) return 1;
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  mw_dycore_s *d = new mw_dycore_s();
  d->g = *g;
  for (int t = 0; t < MW_MAX_TRACERS; t++) { d->pos[t] = (t < g->num_tracers && tracer_positive) ? tracer_positive[t] : 0;
                                             d->adds[t] = (t < g->num_tracers && tracer_adds_mass) ? tracer_adds_mass[t] : 0; }
  d->stream = (hipStream_t)stream;
  const char *s = getenv("MW_STRICT");
  d->strict = (s && s[0] == '1');
  size_t nzc = (size_t)g->nz * g->nens, nze = (size_t)(g->nz + 1) * g->nens;
  // fused tables (hyc | hytc | hye | hyte | p0c | ihytc | p0e | ihyte | packed rows), then per member: hyc | hytc | p0c | ihytc | packed rows
  d->hy_host.assign(4 * nzc + 4 * nze + 8 * nze + (size_t)g->nens * (4 * (size_t)g->nz + 8 * (size_t)(g->nz + 1)), 0.0);
  auto fail = [&](void) { mw_dycore_destroy(d); return 1; };
  if (hipMalloc(&d->hy_dev, d->hy_host.size() * sizeof(double)) != hipSuccess) { set_error("hipMalloc(hy) failed"); return fail(); }
  fill_params(d);
  if (!d->strides_ok) { set_error("mw_dycore_create: a variable of this block has more than 2^31 - 1 elements (16 GB): too large for one handle"); return fail(); }
  const DyP &p = d->p;
  size_t slab = (size_t)p.V * p.sV * sizeof(double);
  size_t fxb = (size_t)p.V * p.fxV * sizeof(double), fyb = (size_t)p.V * p.fyV * sizeof(double), fzb = (size_t)p.V * p.fzV * sizeof(double);
  if (hipMalloc(&d->S0, slab) != hipSuccess || hipMalloc(&d->S1, slab) != hipSuccess || hipMalloc(&d->S2, slab) != hipSuccess ||
      hipMalloc(&d->S3, slab) != hipSuccess ||
      hipMalloc(&d->tendY, (size_t)5 * p.nC * sizeof(double)) != hipSuccess || hipMalloc(&d->FX, fxb) != hipSuccess ||
      hipMalloc(&d->FY, fyb) != hipSuccess || hipMalloc(&d->FZ, fzb) != hipSuccess ||
      hipMalloc(&d->imm, (size_t)p.nC * sizeof(double)) != hipSuccess || hipMalloc(&d->flags, (size_t)p.nC) != hipSuccess ||
      hipMalloc(&d->dirty, 4 * sizeof(unsigned int)) != hipSuccess) {
    set_error("hipMalloc(workspace) failed"); return fail(); }
  (void)hipMemsetAsync(d->flags, 0, (size_t)p.nC, d->stream);
  (void)hipMemsetAsync(d->dirty, 0, 4 * sizeof(unsigned int), d->stream);
  d->fused = (g->num_tracers <= 4 && g->nens <= 12) ? 1 : 0;       // (option "fused_tracers" = 0: the unfused tracer stage)
  // zero everything once: halo corners are never written and the flux arrays start at 0 (:1677-1682)
  (void)hipMemsetAsync(d->S0, 0, slab, d->stream); (void)hipMemsetAsync(d->S1, 0, slab, d->stream);
  (void)hipMemsetAsync(d->S2, 0, slab, d->stream); (void)hipMemsetAsync(d->S3, 0, slab, d->stream); (void)hipMemsetAsync(d->tendY, 0, (size_t)5 * p.nC * sizeof(double), d->stream);
  (void)hipMemsetAsync(d->FX, 0, fxb, d->stream); (void)hipMemsetAsync(d->FY, 0, fyb, d->stream); (void)hipMemsetAsync(d->FZ, 0, fzb, d->stream);
  (void)hipMemsetAsync(d->imm, 0, (size_t)p.nC * sizeof(double), d->stream);
  d->nWE1 = (long long)p.nz * p.ny * p.HX * p.nens;
  d->nSN1 = (long long)p.nz * p.HY * p.nx * p.nens;
  {
    const size_t fn[3] = {(size_t)p.fxV, (size_t)p.fyV, (size_t)p.fzV};
    for (int b = 0; b < 2; b++) for (int a = 0; a < 3; a++) {
      if (hipMalloc(&d->M[b][a], fn[a] * sizeof(double)) != hipSuccess || hipMalloc(&d->UP[b][a], fn[a]) != hipSuccess) { set_error("hipMalloc(M/UP) failed"); return fail(); }
      (void)hipMemsetAsync(d->M[b][a], 0, fn[a] * sizeof(double), d->stream); (void)hipMemsetAsync(d->UP[b][a], 0, fn[a], d->stream);
    }
    { // The tracer stream gets the highest stream priority: its kernels are the older work (stage s while the state stream is
      // already in stage s+1), and with both pipelines fp64-VALU bound an even split of the chip only stretches both (measured on
      // one rank with the two-stream schedule forced: step time -0.5 % against equal priorities).
      int least = 0, greatest = 0; (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      hipError_t er = hipStreamCreateWithPriority(&d->tstream, hipStreamNonBlocking, greatest);
      if (er != hipSuccess) { set_error("hipStreamCreate failed"); return fail(); } }
    for (int i = 0; i < 8; i++) if (hipEventCreateWithFlags(&d->ev_state[i], hipEventDisableTiming) != hipSuccess ||
                                    hipEventCreateWithFlags(&d->ev_tr[i], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return fail(); }
    if (hipEventCreateWithFlags(&d->ev_misc, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return fail(); }
    for (int i = 0; i < 7; i++)
      if (hipEventCreateWithFlags(&d->ev_pipe[i], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return fail(); }
  }
  fill_params(d);
  if (hipStreamSynchronize(d->stream) != hipSuccess) { set_error("stream sync failed in create"); return fail(); }
  *h = d;
  return 0;
}

void mw_dycore_destroy(mw_dycore_t d) {
  if (!d) return;
  (void)hipStreamSynchronize(d->stream);
  if (d->tstream) (void)hipStreamSynchronize(d->tstream);
  for (double *ptr : {d->S0, d->S1, d->S2, d->S3, d->tendY, d->FX, d->FY, d->FZ, d->hy_dev, d->imm}) if (ptr) (void)hipFree(ptr);
  if (d->flags) (void)hipFree(d->flags);
  if (d->dirty) (void)hipFree(d->dirty);
  if (d->zr) (void)hipFree(d->zr);
  if (d->zrx) (void)hipFree(d->zrx);
  if (d->zviol) (void)hipFree(d->zviol);
  if (d->pinc) (void)hipFree(d->pinc);
  for (int b = 0; b < 2; b++) for (int a = 0; a < 3; a++) { if (d->M[b][a]) (void)hipFree(d->M[b][a]); if (d->UP[b][a]) (void)hipFree(d->UP[b][a]); }
  for (int i = 0; i < 8; i++) { if (d->ev_state[i]) (void)hipEventDestroy(d->ev_state[i]); if (d->ev_tr[i]) (void)hipEventDestroy(d->ev_tr[i]); }
  if (d->ev_misc) (void)hipEventDestroy(d->ev_misc);
  for (int i = 0; i < 7; i++) if (d->ev_pipe[i]) (void)hipEventDestroy(d->ev_pipe[i]);
  if (d->tstream) (void)hipStreamDestroy(d->tstream);
  if (d->xchg_free && d->xchg_ctx) d->xchg_free(d->xchg_ctx);
  for (int g = 0; g < 2; g++) for (int b = 0; b < 8; b++) if (d->bufs[g][b]) (void)hipFree(d->bufs[g][b]);
  for (int w = 0; w < 12; w++) for (auto &pr : d->ev[w]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  delete d;
}

int mw_dycore_get_grid(mw_dycore_t d, mw_grid_t *g) { if (!d || !g) MW_FAIL("null argument"); *g = d->g; return 0; }
double mw_dycore_get_etime(mw_dycore_t d) { return d ? d->etime : -1.0; }
double *mw_dycore_immersed_proportion(mw_dycore_t d) { return d ? d->imm : nullptr; }

int mw_dycore_set_bc(mw_dycore_t d, int bc_x, int bc_y, int bc_z) {
  if (!d) MW_FAIL("null handle");
  for (int b : {bc_x, bc_y, bc_z}) if (b != MW_BC_PERIODIC && b != MW_BC_OPEN && b != MW_BC_WALL) MW_FAIL("bc_x / bc_y / bc_z must be 0 (periodic), 1 (open) or 2 (wall)");
  const int old[3] = {d->g.bc_x, d->g.bc_y, d->g.bc_z};
  d->g.bc_x = bc_x; d->g.bc_y = bc_y; d->g.bc_z = bc_z;
  fill_params(d);
  if (check_halo_fit(d)) { d->g.bc_x = old[0]; d->g.bc_y = old[1]; d->g.bc_z = old[2]; fill_params(d); return 1; }
  return 0;
}
int mw_dycore_set_strict(mw_dycore_t d, int strict) { if (!d) MW_FAIL("null handle"); d->strict = strict; return 0; }

// ---- run-time options (see DyOpts) ------------------------------------------------------------------------------------
namespace {
struct OptDesc { const char *key; int DyOpts::*field; long long lo, hi; int build; };   // build: 0 = every build has it (the experiment builds of rounds 4-5 left the tree in round 6)
const OptDesc OPTS[] = {
  {"overlap", &DyOpts::overlap, -1, 1, 0}, {"pipe", &DyOpts::pipe, 0, 1, 0}, {"pipe_edge_inline", &DyOpts::pipe_edge_inline, 0, 1, 0},
  {"pipe_convert", &DyOpts::pipe_convert, 0, 1, 0}, {"pipe_split_edges", &DyOpts::pipe_split_edges, 0, 1, 0}, {"spec", &DyOpts::spec, 0, 1, 0}, {"wrap", &DyOpts::wrap, 0, 1, 0},
  {"y_all", &DyOpts::y_all, 0, 1, 0}, {"y_all_conv", &DyOpts::y_all_conv, 0, 1, 0}, {"member_major", &DyOpts::member_major, 0, 1, 0},
  {"mm_direct", &DyOpts::mm_direct, 0, 1, 0}, {"mm_conv", &DyOpts::mm_conv, 0, 1, 0}, {"fused_convert", &DyOpts::fused_convert, 0, 1, 0},
  {"fused_convert_mm", &DyOpts::fused_convert_mm, 0, 1, 0}, {"chunk_y", &DyOpts::chunk_y, 0, 1 << 20, 0}, {"chunk_yt", &DyOpts::chunk_yt, 0, 1 << 20, 0},
  {"chunk_z", &DyOpts::chunk_z, 0, 1 << 20, 0}, {"chunk_f", &DyOpts::chunk_f, 0, 1 << 20, 0}, {"chunk_model", &DyOpts::chunk_model, 0, 1, 0},
  {"tf_rows4", &DyOpts::tf_rows4, 0, 1, 0}, {"zero_skip", &DyOpts::zero_skip, 0, 1, 0}, {"zero_rows", &DyOpts::zero_rows, 0, 1, 0}, {"zero_stores", &DyOpts::zero_stores, 0, 1, 0}, {"zero_verify", &DyOpts::zero_verify, 0, 1, 0}, {"pipe_maps_early", &DyOpts::pipe_maps_early, 0, 1, 0}, {"rccl_lanes", &DyOpts::rccl_lanes, 0, 2, 0}, {"rccl_two_comms", &DyOpts::rccl_two_comms, -1, 1, 0},
  {"xchg_fuzz", &DyOpts::xchg_fuzz, 0, 0x7fffffff, 0}, {"rccl_prio", &DyOpts::rccl_prio, 0, 1, 0}, {"rccl_inline", &DyOpts::rccl_inline, 0, 1, 0},
  {"debug_no_patch", &DyOpts::debug_no_patch, 0, 1, 0},
};
constexpr int BUILD_FLAGS = 0;      // (no optional parts any more: mw_build_flags stays in the ABI and says so)
}
int mw_build_flags(void) { return BUILD_FLAGS; }
int mw_dycore_set_option(mw_dycore_t d, const char *key, long long value) {
  if (!d || !key) MW_FAIL("mw_dycore_set_option: null argument");
  if (!strcmp(key, "fused_tracers")) {                          // (0: the unfused tracer stage k_xz_tracers + k_tracer_update)
    if (value != 0 && value != 1) MW_FAIL("option fused_tracers must be 0 or 1");
    if (value && !(d->g.num_tracers <= 4 && d->g.nens <= 12)) MW_FAIL("option fused_tracers = 1 needs at most 4 tracers and 12 members");
    d->fused = (int)value;
    return 0;
  }
  for (const OptDesc &od : OPTS) {
    if (strcmp(key, od.key)) continue;
    if (value < od.lo || value > od.hi) MW_FAIL(std::string("option ") + key + ": value out of range [" + std::to_string(od.lo) + ", " + std::to_string(od.hi) + "]");
    d->o.*(od.field) = (int)value;
    if (!strncmp(key, "chunk_", 6)) d->chunk_y = d->chunk_yt = d->chunk_z = d->chunk_f = 0;      // (cached chunk sizes: decided again at the next launch)
    return 0;
  }
  MW_FAIL(std::string("unknown option: ") + key);
}
int mw_dycore_get_option(mw_dycore_t d, const char *key, long long *value) {
  if (!d || !key || !value) MW_FAIL("mw_dycore_get_option: null argument");
  if (!strcmp(key, "fused_tracers")) { *value = d->fused; return 0; }
  for (const OptDesc &od : OPTS) if (!strcmp(key, od.key)) { *value = d->o.*(od.field); return 0; }
  MW_FAIL(std::string("unknown option: ") + key);
}
int mw_dycore_set_order(mw_dycore_t d, int ord) {
  if (!d) MW_FAIL("null handle");
  if (ord != 3 && ord != 5 && ord != 7 && ord != 9) MW_FAIL("WENO order must be 3, 5, 7 or 9");
  // orders 7 and 9 reach further: x / y halo hs + 1 (the neighbour's edge value is rebuilt locally), z halo hs
  const int hs = (ord - 1) / 2, hx = std::max(HXc, hs + 1), hz = std::max(HZc, hs);
  const int old_ord = d->ord, old_hx = d->hxw, old_hz = d->hzw;
  auto rollback = [&]() { d->ord = old_ord; d->hxw = old_hx; d->hzw = old_hz; fill_params(d); return 1; };
  d->ord = ord; d->hxw = hx; d->hzw = hz;
  fill_params(d);
  if (!d->strides_ok) { set_error("mw_dycore_set_order: with this order's halo a variable of the block has more than 2^31 - 1 elements"); return rollback(); }
  if (check_halo_fit(d)) return rollback();
  if (hx != old_hx || hz != old_hz) {
    MW_HIP(hipStreamSynchronize(d->stream));
    if (d->tstream) MW_HIP(hipStreamSynchronize(d->tstream));
    // the four slabs in their new size first; the handle only changes once all of them exist
    const size_t slab = (size_t)d->p.V * d->p.sV * sizeof(double);
    double *fresh[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
      if (hipMalloc(&fresh[i], slab) != hipSuccess || hipMemsetAsync(fresh[i], 0, slab, d->stream) != hipSuccess) {   // halo corners are never written (as in create)
        for (int j = 0; j <= i; j++) if (fresh[j]) (void)hipFree(fresh[j]);
        set_error("mw_dycore_set_order: hipMalloc of the wider slabs failed (the handle keeps its previous order)");
        return rollback();
      }
    }
    d->zr_prev_ok = false; d->kz_buf[0] = d->kz_buf[1] = nullptr;
    double **S[4] = {&d->S0, &d->S1, &d->S2, &d->S3};
    for (int i = 0; i < 4; i++) { if (*S[i]) (void)hipFree(*S[i]); *S[i] = fresh[i]; }
    d->flux_src = nullptr;                                      // pointed into a slab that no longer exists
    d->chunk_y = d->chunk_yt = d->chunk_z = d->chunk_f = 0;
    d->nWE1 = (long long)d->p.nz * d->p.ny * d->p.HX * d->p.nens;
    d->nSN1 = (long long)d->p.nz * d->p.HY * d->p.nx * d->p.nens;
    bool had = false;
    for (int g = 0; g < 2; g++) for (int b = 0; b < 8; b++) if (d->bufs[g][b]) { (void)hipFree(d->bufs[g][b]); d->bufs[g][b] = nullptr; had = true; }
    if (had) {                                                  // strips are HX / HY cells deep: re-allocate (the transport and its owner stay)
      auto owner = d->xchg_free; auto fn = d->xchg; void *ctx = d->xchg_ctx;
      d->xchg_free = nullptr;                                   // (set_exchange must not free the context it is about to re-install)
      if (mw_dycore_set_exchange(d, fn, ctx)) { d->xchg_free = owner; d->xchg = fn; d->xchg_ctx = ctx; return 1; }
      d->xchg_free = owner;
    }
    MW_HIP(hipStreamSynchronize(d->stream));
  }
  return 0;
}

int mw_dycore_set_background(mw_dycore_t d, const double *hyc, const double *hytc, const double *hye, const double *hyte,
                             const double *immersed_proportion) {
  if (!d || !hyc || !hytc || !hye || !hyte) MW_FAIL("null argument");
  size_t nzc = (size_t)d->g.nz * d->g.nens, nze = (size_t)(d->g.nz + 1) * d->g.nens;
  memcpy(d->hy_host.data(), hyc, nzc * 8); memcpy(d->hy_host.data() + nzc, hytc, nzc * 8);
  memcpy(d->hy_host.data() + 2 * nzc, hye, nze * 8); memcpy(d->hy_host.data() + 2 * nzc + nze, hyte, nze * 8);
  if (upload_background(d)) return 1;
  if (immersed_proportion) MW_HIP(hipMemcpyAsync(d->imm, immersed_proportion, (size_t)d->p.nC * 8, hipMemcpyDeviceToDevice, d->stream));
  else MW_HIP(hipMemsetAsync(d->imm, 0, (size_t)d->p.nC * 8, d->stream));
  return 0;
}

int mw_dycore_get_background(mw_dycore_t d, double *hyc, double *hytc, double *hye, double *hyte) {
  if (!d) MW_FAIL("null handle");
  size_t nzc = (size_t)d->g.nz * d->g.nens, nze = (size_t)(d->g.nz + 1) * d->g.nens;
  if (hyc) memcpy(hyc, d->hy_host.data(), nzc * 8);
  if (hytc) memcpy(hytc, d->hy_host.data() + nzc, nzc * 8);
  if (hye) memcpy(hye, d->hy_host.data() + 2 * nzc, nze * 8);
  if (hyte) memcpy(hyte, d->hy_host.data() + 2 * nzc + nze, nze * 8);
  return 0;
}

int mw_dycore_get_fluxes(mw_dycore_t d, double **out6) {
  if (!d || !out6) MW_FAIL("null argument");
  if (d->flux_src) {     // production path: the state-variable fluxes of the last stage were never written; rebuild all six
    if (d->member_major) {   // the retained stage input is member-major: bring it into the fused layout the general kernels read (S1 is free)
      const long long n = (long long)d->p.V * d->p.sV;
      d->zr_prev_ok = false;
      MW_KLAUNCH(k_member_to_fused, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d->stream, d->p, d->flux_src, d->S1);
      MW_LAUNCH_CHECK();
      d->flux_src = d->S1;
    }
    { // the marching kernels apply the z boundary rule (and, in a periodic direction owned by one rank, the wrap) while
      // loading and leave those halos of the slab unfilled: fill them for k_flux
      DyP p = d->p; p.v0 = 0;
      double *S = const_cast<double *>(d->flux_src);
      const long long nx_ = (long long)p.V * p.nz * p.ny * 2 * p.HX * p.nens, ny_ = (long long)p.V * p.nz * 2 * p.HY * p.nx * p.nens;
      const long long nz_ = (long long)p.V * 2 * p.HZ * p.ny * p.nx * p.nens;
      const unsigned nbx = p.wrap_x ? (unsigned)((nx_ + 255) / 256) : 0u, nby = p.wrap_y ? (unsigned)((ny_ + 255) / 256) : 0u;
      MW_KLAUNCH(k_halo_xyz, dim3(nbx + nby + (unsigned)((nz_ + 255) / 256)), dim3(256), 0, d->stream, p, S, nbx, nby);
      MW_LAUNCH_CHECK(); }
    if (launch_flux(d, d->flux_src)) return 1;              // arrays from the retained stage input exactly as D9+D10 leave them
    if (launch_fct(d, d->flux_src, d->flux_dt)) return 1;
    d->flux_src = nullptr;
  }
  out6[0] = d->FX; out6[1] = d->FY; out6[2] = d->FZ;
  out6[3] = d->FX + 5 * d->p.fxV; out6[4] = d->FY + 5 * d->p.fyV; out6[5] = d->FZ + 5 * d->p.fzV;
  return 0;
}

int mw_dycore_set_exchange(mw_dycore_t d, mw_exchange_fn fn, void *ctx) {
  if (!d) MW_FAIL("null handle");
  if (d->xchg_free && d->xchg_ctx && d->xchg_ctx != ctx) {     // a transport the handle owned is being replaced
    (void)hipStreamSynchronize(d->stream); if (d->tstream) (void)hipStreamSynchronize(d->tstream);
    d->xchg_free(d->xchg_ctx);
  }
  d->xchg_free = nullptr;
  d->xchg = nullptr; d->xchg_ctx = nullptr;                    // committed below, once the strip buffers exist: a failed allocation
  if (fn) {                                                    // must not leave a transport (and a context its owner then frees) behind
    for (int g = 0; g < 2; g++) for (int b = 0; b < 8; b++) {
      if (d->bufs[g][b]) continue;
      long long n = ((b % 4 < 2) ? d->nWE1 : d->nSN1) * d->p.V;          // sized for all V variables
      if (n == 0) n = 1;
      MW_HIP(hipMalloc(&d->bufs[g][b], (size_t)n * sizeof(double)));
    }
  }
  d->xchg = fn; d->xchg_ctx = ctx;
  return 0;
}

int mw_dycore_profile(mw_dycore_t d, int enable) {
  if (!d) MW_FAIL("null handle");
  MW_HIP(hipStreamSynchronize(d->stream));
  d->prof = enable;
  for (int w = 0; w < 12; w++) d->ev_used[w] = 0;
  return 0;
}
int mw_dycore_profile_get(mw_dycore_t d, int which, double *total_ms, long long *launches) {
  if (!d || which < 0 || which > 11) MW_FAIL("bad argument");
  MW_HIP(hipStreamSynchronize(d->stream));
  double tot = 0;
  for (size_t i = 0; i < d->ev_used[which]; i++) { float ms = 0; MW_HIP(hipEventElapsedTime(&ms, d->ev[which][i].first, d->ev[which][i].second)); tot += ms; }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (long long)d->ev_used[which];
  return 0;
}

// ---- parked column increments (round 6) ---------------------------------------------------------------------------------------------
// ColumnNudger::nudge_to_column is two passes over five fields: the horizontal sums, and `state += dt (column - avg) / 900` (1.3 GB read and
// written again on config 2, 0.25 ms of a 5.4 ms loop iteration).  The second pass adds ONE number per (field, level) -- and the next thing the
// reference's loop does with those five arrays is dycore.time_step, whose conversion D1 reads them once and whose D13 overwrites them.  The
// deferred form parks the increments in the dycore handle instead; the next mw_dycore_time_step adds them while its converting y launch
// loads the coupler's values (the same rounded addition, so the result is bit for bit the eager one's) and the pass disappears.  Anybody who
// wants to SEE the fields in between calls mw_dycore_flush_pending first (the Coupler mirrors do that inside DataManager::get); a time step
// on a path whose conversion does not take increments along (strict / general kernels, decomposed blocks, members, 2-D) flushes by itself.
static int flush_pending(mw_dycore_s *d) {
  if (!d->pinc_on) return 0;
  const DyP &p = d->p;
  Ptr5 fp; for (int l = 0; l < 5; l++) fp.f[l] = d->pinc_fields[l];
  const long long ncell_lev = (long long)p.ny * p.nx;
  const unsigned nb = (unsigned)std::min<long long>((ncell_lev * p.nens + 256ll * 8 - 1) / (256ll * 8), 65535);
  MW_KLAUNCH(k_apply_pending, dim3(nb, (unsigned)p.nz, 5u), dim3(256), 0, d->stream, fp, p.nz, ncell_lev, p.nens, d->pinc);
  MW_LAUNCH_CHECK();
  d->pinc_on = false; d->pinc_eager++;
  return 0;
}
int mw_dycore_flush_pending(mw_dycore_t d) {
  if (!d) MW_FAIL("null handle");
  if (!d->pinc_on) return 0;                                    // (the Coupler mirrors call this in front of every field access: nothing parked = nothing done)
  fill_params(d);
  return flush_pending(d);
}
/* 1: increments are parked; out2 (may be NULL): how often parked increments rode on a conversion / were applied by a pass, since create */
int mw_dycore_pending(mw_dycore_t d, unsigned long long *out2) {
  if (!d) return 0;
  if (out2) { out2[0] = d->pinc_lazy; out2[1] = d->pinc_eager; }
  return d->pinc_on ? 1 : 0;
}
int mw_nudge_to_column_deferred(mw_dycore_t d, double *const *state5, const double *column, double dt, void *workspace, mw_allreduce_fn allreduce,
                                void *ctx) {
  if (!d || !state5 || !column || !workspace) MW_FAIL("nudge_to_column_deferred: null argument");
  for (int l = 0; l < 5; l++) if (!state5[l]) MW_FAIL("nudge_to_column_deferred: null field");
  fill_params(d);
  if (flush_pending(d)) return 1;                              // (increments of an earlier call that no time step has consumed: they count in the averages)
  const DyP &p = d->p;
  const long long n = 5ll * p.nz * p.nens;
  if (!d->pinc) MW_HIP(hipMalloc(&d->pinc, (size_t)n * sizeof(double)));
  double *avg = (double *)workspace, *rest = avg + n;
  if (mw_column_average(&d->g, state5, avg, rest, allreduce, ctx, d->stream)) return 1;      // (ordered on the handle's stream, like the time step that will use them)
  MW_KLAUNCH(k_nudge_increments, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d->stream, column, avg, dt, n, d->pinc);
  MW_LAUNCH_CHECK();
  for (int l = 0; l < 5; l++) d->pinc_fields[l] = state5[l];
  d->pinc_on = true;
  return 0;
}

// ---- time_step (:81-198) -------------------------------------------------------------------------------
int mw_dycore_time_step(mw_dycore_t d, double *rho_d, double *u, double *v, double *w, double *temp, double *const *tracers,
                        double dt_phys) {
  if (!d) MW_FAIL("null handle");
  if (!(dt_phys > 0)) MW_FAIL("dt_phys must be > 0");
  CouplerPtrs c;
  if (make_coupler_ptrs(d, rho_d, u, v, w, temp, tracers, c)) return 1;
  fill_params(d);
  const DyP &p = d->p;
  if (need_exchange(d) || check_halo_fit(d)) return 1;
  ProfScope step_scope(d, 9);                                   // (mw_dycore_profile(h, 3): the whole time step, first launch to the join on the handle's stream)
  dim3 cgrid = plane_grid((long long)p.ny * p.nx * p.nens, p.nz);
  // production path; strict = 1/2 use the general flux-materialising kernels below.  So does a z-PERIODIC domain (:752-763,
  // :1008-1019; no shipped case): the marching kernels apply the wall / open z rule while loading and have no periodic form.
  // WENO-3 (the reference's GPU-benchmark build, -DMW_ORD=3) marches too, in the forms that exist for it: fused tracer stage, and
  // nens == 1 or the member-major layout.  Orders 7 / 9 run on the general kernels.
  const bool ord3_ok = d->fused && (p.nens == 1 || d->o.member_major);
  const bool march = (d->strict == 0) && (p.bc_z != MW_BC_PERIODIC) && (d->ord == 5 || (d->ord == 3 && ord3_ok));
  d->last_march = march ? 1 : 0;
  if (march && d->o.wrap) {                       // index wrap instead of halo cells (see DyP::wrap_x)
    d->p.wrap_x = (p.bc_x == MW_BC_PERIODIC) && !(d->xchg && p.nproc_x > 1) && p.nx >= 3;
    d->p.wrap_y = !p.sim2d && (p.bc_y == MW_BC_PERIODIC) && !(d->xchg && p.nproc_y > 1) && p.ny >= 3;
  }
  // Two-stream schedule (see rk_stage_march): the default when strips are exchanged with neighbour ranks (the exchange of one
  // pipeline then runs beside the kernels of the other).  On one rank both pipelines are fp64-VALU bound, the step takes the
  // same time either way (measured without any timing events: +-0.3 %) and sharing the chip only stretches every kernel, so the
  // default there is the handle's stream for everything.  Option "overlap" = 1 / 0 forces either schedule.
  { const bool ov = d->o.overlap >= 0;                          // forced either way
    const bool want = ov ? (d->o.overlap != 0) : (d->xchg != nullptr);
    d->overlap = march && d->tstream && want;
    // ... unless k_y_all applies: then the pipelined one-stream schedule (rk_stage_pipe) is the default with an exchange
    d->pipe = 0;
    if (d->overlap && d->xchg && !ov && d->o.pipe) {
      d->overlap = 0;
      if (y_all_ok(d) && d->ev_pipe[0]) d->pipe = 1; else d->overlap = 1;
    } }
  // nens > 1 on the production path: member-major internal layout (see View)
  // (Measured, round 3, config 4's block 256 x 512 x 128 x 4: the members' coupler-touching launches -- D1 inside k_y_state, D13 inside the
  //  last stage -- issued SIDE BY SIDE on one stream per member, hoping that the quarter lines the four members read / write would
  //  meet in L2: they do not.  27.3 ms per step with the two coalesced conversion passes; 33.2 with D13 inside the member launches
  //  (k_tracers_fused 7.6 -> 12.0 ms, k_xz_state 9.4 -> 11.7), 30.3 with D1 inside (k_y_state 3.8 -> 7.7), 34.3 with both.)
  d->member_major = march && d->fused && p.nens > 1 && d->o.member_major;
  // ... with D13 written by the last stage's kernels themselves and D1 read by the first k_y_state, the members of a tile in one
  // workgroup so that their quarter-sector accesses meet in L1 / L2 (MemberOff, mw_march.h): 2 or 4 members
  d->mm_direct = d->member_major && (p.nens == 2 || p.nens == 4) && d->o.mm_direct;
  // D1 + D2 (:101, :248-255).  Production path with periodic x and y owned by this rank (either schedule: the tracer stream waits
  // for the stage's state kernels anyway): done inside the first k_y_state (no separate pass); otherwise a conversion kernel first
  // (the reference's operation order on the general path).
  d->conv_pending = march && d->p.wrap_x && d->p.wrap_y && p.nt <= 4 && d->o.fused_convert &&
                    (!d->member_major || d->o.fused_convert_mm);
  // Pipelined schedule of a decomposed block (rk_stage_pipe): only the strips that are packed for the neighbours and the rows the
  // edge-strip y launch reads are converted up front; the inner rows are converted by the first k_y_all<true> while the strips travel.
  const bool pipe_conv = d->pipe && (!d->member_major || (d->mm_direct && d->o.mm_conv && marching_config(d, view(d, 0).p) != 0)) && p.nt <= 3 &&
                         d->o.fused_convert && d->o.pipe_convert && (d->p.wrap_y || p.ny >= 4 * MW_Y_EDGE);
  // (the pipelined schedule converts inside k_y_all<true> ONLY in the forms pipe_conv names: any other handle -- e.g. three members, or a
  //  member-major handle whose configuration is not a folded one -- gets the full conversion pass below; without this a 1 x 1
  //  decomposition with a transport installed (both wraps on) reached the members-in-one-workgroup launch with the wrong kernel)
  if (d->pipe) d->conv_pending = false;
  { // the dispatcher's decisions of this time step, for the tests' path-coverage matrix (mw_dycore_path)
    const int K = march ? marching_config(d, view(d, 0).p) : 0;
    d->path = std::string(march ? "march" : (d->strict == 1 ? "general-strict" : "general-fast")) + " ord" + std::to_string(d->ord);
    if (march) {
      d->path += " K" + std::to_string(K);
      d->path += p.nens == 1 ? " nens1" : d->mm_direct ? " mm_direct" : d->member_major ? " member_major" : " fused_members";
      d->path += d->pipe ? " pipe" : d->overlap ? " two_stream" : " one_stream";
      d->path += y_all_ok(d) ? " y_all" : " y_split";
      d->path += pipe_conv ? " conv_pipe" : d->conv_pending ? " conv_in_y" : " conv_pass";
      d->path += d->fused ? " tracers_fused" : " tracers_unfused";
      d->path += p.sim2d ? " 2d" : " 3d";
    } else d->path += p.nens == 1 ? " nens1" : " fused_members";
    if (d->xchg) d->path += " transport"; }
  if (d->pinc_on) {
    // parked column increments: they ride on the conversion where it happens inside k_y_all<true, K = 1> of a one-rank handle, and belong to
    // exactly the arrays this call was handed; anything else applies them with a pass first
    const bool same = c.rho_d == d->pinc_fields[0] && c.u == d->pinc_fields[1] && c.v == d->pinc_fields[2] && c.temp == d->pinc_fields[3] &&
                      p.idWV >= 0 && p.idWV < p.nt && c.tr[p.idWV] == d->pinc_fields[4];
    const bool lazy = same && march && d->conv_pending && !d->pipe && !d->overlap && !d->member_major && p.nens == 1 && marching_config(d, d->p) == 1 &&
                      y_all_ok(d) && d->o.y_all_conv;
    if (lazy) { d->p.pinc = d->pinc; d->pinc_on = false; d->pinc_lazy++; }
    else if (flush_pending(d)) return 1;
  }
  d->pre_lo = d->pre_hi = 0;
  d->entry_marked = false;
  if (pipe_conv && d->ev_pipe[6]) { MW_HIP(hipEventRecord(d->ev_pipe[6], d->stream)); d->entry_marked = true; }   // the coupler's arrays are ready here (see rk_stage_pipe: the row scan runs beside the strip conversion)
  if (pipe_conv) {
    ProfScope ps(d, 4);
    const int ylo = d->p.wrap_y ? 0 : MW_Y_EDGE + 3, yhi = d->p.wrap_y ? p.ny : p.ny - MW_Y_EDGE - 3;
    d->pre_lo = ylo; d->pre_hi = yhi;
    if (d->member_major) {
      const View v = view(d, 0);
      const MemberStrides ms = {v.p.sJ, v.p.sK, v.p.sV, v.slab};
      MW_KLAUNCH(k_coupler_to_member, cgrid, dim3(256), 0, d->stream, p, c, d->S0, ms, ylo, yhi);
    } else
    { // the strip cells only (see the kernel)
      const long long nstrip = (long long)(ylo + p.ny - yhi) * p.nx * p.nens + (long long)(yhi - ylo) * 2 * p.HX * p.nens;
      const bool strips = p.nx > 2 * p.HX && nstrip > 0;
      MW_KLAUNCH(k_coupler_to_state_fast, strips ? plane_grid(nstrip, p.nz) : cgrid, dim3(256), 0, d->stream, p, c, d->S0, ylo, yhi, strips ? 1 : 0); }
    MW_LAUNCH_CHECK();
    d->conv_pending = true;
  }
  if (!d->conv_pending) {
    ProfScope ps(d, 4);
    if (d->member_major) {      // one coalesced pass in the coupler's order (see k_coupler_to_member)
      const View v = view(d, 0);
      const MemberStrides ms = {v.p.sJ, v.p.sK, v.p.sV, v.slab};
      MW_KLAUNCH(k_coupler_to_member, cgrid, dim3(256), 0, d->stream, p, c, d->S0, ms, p.ny, p.ny);
    } else if (march && p.nt <= 4) MW_KLAUNCH(k_coupler_to_state_fast, cgrid, dim3(256), 0, d->stream, p, c, d->S0, p.ny, p.ny, 0);
    else       MW_KLAUNCH(k_coupler_to_state, cgrid, dim3(256), 0, d->stream, p, c, d->S0);
    MW_LAUNCH_CHECK();
  }
  if (d->overlap) { MW_HIP(hipEventRecord(d->ev_misc, d->stream)); MW_HIP(hipStreamWaitEvent(d->tstream, d->ev_misc, 0)); d->gstage = 0; }
  double dt_dyn = mw_dycore_compute_time_step(&d->g);                     // :104
  int ncycles = (int)std::ceil(dt_phys / dt_dyn);                         // :107
  dt_dyn = dt_phys / ncycles;                                             // :108
  for (int icycle = 0; icycle < ncycles; icycle++) {
    bool last = (icycle == ncycles - 1);
    d->first_cycle = (icycle == 0);
    if (march) {
      double *Q[4] = {d->S0, d->S1, d->S2, d->S3};
      if (rk_cycle_march(d, Q, dt_dyn, last, c)) return 1;
      std::swap(d->S0, d->S3);                                // (P,A,B,C) -> (C,A,B,P): the new q^n is C
      continue;
    }
    // stage 1 (:119-132)
    d->zr_prev_ok = false; zero_rows_forget(d, nullptr);        // (the general path writes the slabs without maps)
    if (halo_fill(d, d->S0)) return 1;
    if (launch_flux(d, d->S0)) return 1;
    if (launch_fct(d, d->S0, dt_dyn)) return 1;
    if (launch_update<1, 0>(d, d->S0, d->S0, d->S1, dt_dyn, dt_dyn, c, nullptr, nullptr)) return 1;
    // stage 2 (:136-153)
    double dt2 = (1.0 / 4.0) * dt_dyn;
    if (halo_fill(d, d->S1)) return 1;
    if (launch_flux(d, d->S1)) return 1;
    if (launch_fct(d, d->S1, dt2)) return 1;
    if (launch_update<2, 0>(d, d->S1, d->S0, d->S1, dt2, dt_dyn, c, nullptr, nullptr)) return 1;
    // stage 3 (:157-174)
    double dt3 = (2.0 / 3.0) * dt_dyn;
    if (halo_fill(d, d->S1)) return 1;
    if (launch_flux(d, d->S1)) return 1;
    if (launch_fct(d, d->S1, dt3)) return 1;
    if (last) { if (launch_update<3, 1>(d, d->S1, d->S0, d->S0, dt3, dt_dyn, c, nullptr, nullptr)) return 1; }     // + :178
    else      { if (launch_update<3, 0>(d, d->S1, d->S0, d->S0, dt3, dt_dyn, c, nullptr, nullptr)) return 1; }
    d->flux_src = nullptr;                                    // all six flux arrays are already materialised
  }
  if (d->overlap) MW_HIP(hipStreamWaitEvent(d->stream, d->ev_tr[(d->gstage - 1) & 7], 0));   // join: the handle's stream owns the result
  d->etime += dt_phys;                                                    // :181
  return 0;
}

int mw_dycore_compute_tendencies(mw_dycore_t d, const double *rho_d, const double *u, const double *v, const double *w,
                                 const double *temp, double *const *tracers, double dt, double *state_tend, double *tracers_tend) {
  if (!d) MW_FAIL("null handle");
  CouplerPtrs c;
  if (make_coupler_ptrs(d, rho_d, u, v, w, temp, tracers, c)) return 1;
  fill_params(d);
  const DyP &p = d->p;
  if (need_exchange(d) || check_halo_fit(d)) return 1;
  if (flush_pending(d)) return 1;
  dim3 cgrid = plane_grid((long long)p.ny * p.nx * p.nens, p.nz);
  MW_KLAUNCH(k_coupler_to_state, cgrid, dim3(256), 0, d->stream, p, c, d->S0); MW_LAUNCH_CHECK();
  if (halo_fill(d, d->S0)) return 1;
  if (launch_flux(d, d->S0)) return 1;
  if (launch_fct(d, d->S0, dt)) return 1;
  d->flux_src = nullptr;
  if (state_tend && tracers_tend) { if (launch_update<1, 2>(d, d->S0, d->S0, nullptr, dt, dt, c, state_tend, tracers_tend)) return 1; }
  return 0;
}

int mw_weno5_edges(long long n, const double *stencils, double *edges, int strict, void *stream) {
  if (n < 1 || !stencils || !edges) MW_FAIL("weno5_edges: bad argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  MW_KLAUNCH(k_weno5_edges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stencils, edges, n, strict);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_strict_pow(long long n, const double *x, const double *y, double *out, unsigned char *main_path, void *stream) {
  if (n < 1 || !x || !y || !out) MW_FAIL("strict_pow: bad argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  MW_KLAUNCH(k_strict_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, main_path, n);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_calib_copy(const double *in, double *out, long long n, void *stream) {
  if (!in || !out || n < 1) MW_FAIL("mw_calib_copy: bad arguments");
  MW_KLAUNCH(k_calib_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  MW_LAUNCH_CHECK();
  return 0;
}

// ---- calibration (mw_calib.h) --------------------------------------------------------------------------------------------
// Sustained v_fma_f64 issue rate with `waves_per_simd` wavefronts per SIMD on every CU, for about `seconds` (a short run sizes the
// long one).  out5 (HOST): wave-instructions per second, kernel milliseconds, shader clock in GHz during the run (the kernel's cycle
// counter over its 100 MHz real-time counter; 0 when the two counters run at the same rate on this part), wave-instructions issued, CUs.
int mw_calib_fma64(int waves_per_simd, double seconds, double *out5, void *stream) {
  if (waves_per_simd < 1 || waves_per_simd > 8 || !(seconds > 0) || seconds > 20 || !out5) MW_FAIL("mw_calib_fma64: waves_per_simd in 1..8, seconds in (0, 20]");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  const int cus = device_cus();
  if (cus < 1) MW_FAIL("mw_calib_fma64: cannot read the device's CU count");
  double *sink = nullptr; long long *clk = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  MW_HIP(hipMalloc(&sink, 8)); MW_HIP(hipMalloc(&clk, 16));
  MW_HIP(hipEventCreate(&e0)); MW_HIP(hipEventCreate(&e1));
  const dim3 grid((unsigned)(cus * waves_per_simd));            // 256 threads = one wave per SIMD; waves_per_simd workgroups per CU
  auto run = [&](long long trips, float &ms) -> int {
    MW_HIP(hipEventRecord(e0, st));
    MW_KLAUNCH(k_calib_fma64, grid, dim3(256), 0, st, trips, 1.0, sink, clk);
    MW_LAUNCH_CHECK();
    MW_HIP(hipEventRecord(e1, st));
    MW_HIP(hipEventSynchronize(e1));
    MW_HIP(hipEventElapsedTime(&ms, e0, e1));
    return 0;
  };
  float ms = 0;
  long long trips = 20000;
  int rc = run(trips, ms) || run(trips, ms);                     // (the first launch also loads the code object)
  if (!rc) { trips = std::max(1000ll, (long long)(trips * (seconds * 1e3 / std::max(1e-3f, ms)))); rc = run(trips, ms); }
  long long h[2] = {0, 0};
  if (!rc && hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) { set_error("mw_calib_fma64: download failed"); rc = 1; }
  (void)hipFree(sink); (void)hipFree(clk); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) return 1;
  const double winstr = (double)trips * 64.0 * (double)grid.x * 4.0;
  out5[0] = winstr / (ms * 1e-3); out5[1] = ms;
  out5[2] = (h[1] > 0 && h[0] != h[1]) ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
  out5[3] = winstr; out5[4] = cus;
  return 0;
}

// The arithmetic floor of one RK stage: `cells` cell-stages (24 reconstructions + 3 Riemann solves + the passive fluxes each, the
// production arithmetic of mw_weno.h / mw_march.h) on register windows fed from `tab` -- DEVICE (nlev, 8, 64) doubles, a few KB that
// stay in L2 -- in workgroups of 256 threads, two per CU, `levels` cells per thread (k_xz_state's shape).  bg4 (HOST): hyr, hyt, p0,
// 1/hyt of the level.  sink: DEVICE, one double per thread (mw_calib_stage_arith_threads).  out3 (HOST): milliseconds, cells processed,
// workgroups.  The table decides smooth or rough data; the time is what a stage of that many cells cannot beat on this chip.
long long mw_calib_stage_arith_threads(long long cells, int levels) {
  if (cells < 1 || levels < 1) return 0;
  const long long thr = (cells + levels - 1) / levels;
  return ((thr + 255) / 256) * 256;
}
int mw_calib_stage_arith(const double *tab, int nlev, long long cells, int levels, int active_tracers, const double *bg4, double *sink, double *out3, void *stream) {
  if (!tab || nlev < 6 || cells < 1 || levels < 1 || !bg4 || !sink || !out3) MW_FAIL("mw_calib_stage_arith: bad argument (nlev >= 6)");
  if (active_tracers != 1 && active_tracers != 3) MW_FAIL("mw_calib_stage_arith: active_tracers must be 3 (24 reconstructions per cell) or 1 (cloud and rain zero: 18)");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  const long long thr = mw_calib_stage_arith_threads(cells, levels);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  MW_HIP(hipEventCreate(&e0)); MW_HIP(hipEventCreate(&e1));
  float ms = 0; int rc = 0;
  for (int rep = 0; rep < 2 && !rc; rep++) {                     // (the second launch is the measurement)
    if (hipEventRecord(e0, st) != hipSuccess) rc = 1;
    if (active_tracers == 3) MW_KLAUNCH((k_calib_stage_arith<8>), dim3((unsigned)(thr / 256)), dim3(256), 0, st, tab, nlev, levels, bg4[0], bg4[1], bg4[2], bg4[3], sink);
    else                     MW_KLAUNCH((k_calib_stage_arith<6>), dim3((unsigned)(thr / 256)), dim3(256), 0, st, tab, nlev, levels, bg4[0], bg4[1], bg4[2], bg4[3], sink);
    if (hipGetLastError() != hipSuccess || hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = 1;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) MW_FAIL("mw_calib_stage_arith: launch or timing failed");
  out3[0] = ms; out3[1] = (double)(thr * levels); out3[2] = (double)(thr / 256);
  return 0;
}
// Test aid: the zero-row maps of the last sub-cycle, on the host (include/mw_cdna4.h).
long long mw_debug_zero_maps(mw_dycore_t d, unsigned int *out_host, long long cap_words, int *dims2) {
  if (!d) return 0;
  if (!d->zr || !d->zr_prev_ok || d->member_major) return 0;    // (zr_prev_ok: the last sub-cycle ran with maps; member-major handles keep one set per member)
  (void)hipStreamSynchronize(d->stream);
  if (d->tstream) (void)hipStreamSynchronize(d->tstream);
  const long long n = (long long)MW_ZR_MAPS * d->zr_msz;
  if (dims2) { dims2[0] = d->p.nz; dims2[1] = d->p.ny + 2 * MW_ZR_HALO; }
  if (out_host && cap_words > 0)
    (void)hipMemcpy(out_host, d->zr + (long long)d->zr_cur * n, (size_t)std::min(n, cap_words) * sizeof(unsigned), hipMemcpyDeviceToHost);
  return n;
}
// Test aid: the four violation counters of option zero_verify (k_zero_verify, mw_march.h) since the handle was created; -1: the option
// never ran.  out4: [0] input row non-zero under a clear Qs word, [1] ... under a clear QYs word, [2] a destination row the tracer kernel
// was told holds zeros does not, [3] likewise a row of the slab the converting y launch fills.
long long mw_debug_zero_violations(mw_dycore_t d, unsigned long long *out4) {
  if (!d || !d->zviol) return -1;
  (void)hipStreamSynchronize(d->stream);
  if (d->tstream) (void)hipStreamSynchronize(d->tstream);
  unsigned long long h[4] = {0, 0, 0, 0};
  if (hipMemcpy(h, d->zviol, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (out4) for (int i = 0; i < 4; i++) out4[i] = h[i];
  return (long long)(h[0] + h[1] + h[2] + h[3]);
}
// Test aid: the names (as the code object spells them, i.e. mangled; newline-separated) of the dycore kernels this PROCESS has launched
// since the last reset -- every instantiation of the dispatcher's templates has its own.  Returns the bytes needed (terminator included);
// writes at most `cap` of them.  reset != 0 clears the registry afterwards.
long long mw_debug_launched_kernels(char *buf, long long cap, int reset) {
  std::string all;
  {
    std::lock_guard<std::mutex> lk(g_launch_mu);
    for (const void *fn : g_launched) {
      const char *n = hipKernelNameRefByPtr(fn, nullptr);
      if (!n) { (void)hipGetLastError(); continue; }
      all += n; all += "\n";
    }
    if (reset) g_launched.clear();
  }
  if (buf && cap > 0) { const size_t m = std::min<size_t>((size_t)cap - 1, all.size()); memcpy(buf, all.data(), m); buf[m] = 0; }
  return (long long)all.size() + 1;
}
// Test aid: occupies `stream` for about `usec` microseconds (one wavefront polling the 100 MHz counter).
int mw_debug_spin(long long usec, void *stream) { return launch_spin(usec, (hipStream_t)stream); }

int mw_perturb_temperature(const mw_grid_t *g, double *temp, void *stream) {
  if (!g || !temp) MW_FAIL("null argument");
  long long n = (long long)g->nz * g->ny * g->nx * g->nens;
  MW_KLAUNCH(k_perturb_temperature, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g->nz, g->ny, g->nx,
                     g->nens, g->i_beg, g->j_beg, g->xlen / g->nx_glob, g->ylen / g->ny_glob, g->zlen / g->nz, g->xlen, g->ylen, temp);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_perturb_temperature_random(const mw_grid_t *g, double *temp, void *stream) {
  if (!g || !temp) MW_FAIL("null argument");
  const int num_levels = g->nz / 4;
  const long long ncol = (long long)g->ny * g->nx * g->nens;
  if (num_levels < 1) return 0;
  const unsigned long long myrank = (unsigned long long)g->py * g->nproc_x + g->px;
  const unsigned long long seed = myrank * (unsigned long long)g->nz * g->nx * g->ny * g->nens;
  const long long n = (long long)num_levels * ncol;
  MW_KLAUNCH(k_perturb_temperature_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, num_levels, ncol, seed, temp);
  MW_LAUNCH_CHECK();
  return 0;
}

} // extern "C"

namespace mw {
int dycore_set_exchange_owned(mw_dycore_t d, mw_exchange_fn fn, void *ctx, void (*free_ctx)(void *)) {
  if (mw_dycore_set_exchange(d, fn, ctx)) return 1;
  d->xchg_free = free_ctx;
  return 0;
}
// the installed transport of a handle (mw_rccl.cpp recognises its own by the callback's address)
int dycore_option(mw_dycore_t d, const char *key) { long long v = 0; return (d && !mw_dycore_get_option(d, key, &v)) ? (int)v : 0; }
void *dycore_exchange_ctx(mw_dycore_t d, mw_exchange_fn *fn) { if (fn) *fn = d ? d->xchg : nullptr; return d ? d->xchg_ctx : nullptr; }
} // namespace mw

// Which schedule the LAST mw_dycore_time_step of this handle chose (decided per call from the transport, the configuration and the
// MW_* switches): 0 = one stream, 1 = two streams (state | tracer pipelines, rk_stage_march with overlap), 2 = the pipelined
// one-stream schedule of a decomposed block (rk_stage_pipe); + 4 when the y faces of all variables go through the one k_y_all launch,
// + 8 when that time step ran on the general (flux-materialising) kernels instead of the marching ones.  -1: null handle.
extern "C" const char *mw_dycore_path(mw_dycore_t d) { return d ? d->path.c_str() : ""; }
extern "C" int mw_dycore_schedule(mw_dycore_t d) {
  if (!d) return -1;
  return (d->pipe ? 2 : d->overlap ? 1 : 0) + (d->last_march && y_all_ok(d) ? 4 : 0) + (d->last_march ? 0 : 8);
}

// ---- init (:1197-1683): host column profiles + device quadrature --------------------------------------
namespace {

double h_supercell_temperature(double z, double z_0, double z_trop, double z_top, double T_0, double T_trop, double T_top) {  // :1144-1153
  if (z <= z_trop) { double lapse = -(T_trop - T_0) / (z_trop - z_0); return T_0 - lapse * (z - z_0); }
  double lapse = -(T_top - T_trop) / (z_top - z_trop);
  return T_trop - lapse * (z - z_trop);
}
double h_supercell_pressure_dry(double z, double z_0, double z_trop, double z_top, double T_0, double T_trop, double T_top,
                                double p_0, double R_d, double grav) {          // :1157-1177
  if (z <= z_trop) {
    double lapse = -(T_trop - T_0) / (z_trop - z_0);
    double T = h_supercell_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top);
    return p_0 * pow(T / T_0, grav / (R_d * lapse));
  }
  double lapse = -(T_trop - T_0) / (z_trop - z_0);
  double p_trop = p_0 * pow(T_trop / T_0, grav / (R_d * lapse));
  lapse = -(T_top - T_trop) / (z_top - z_trop);
  if (lapse != 0) {
    double T = h_supercell_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top);
    return p_trop * pow(T / T_trop, grav / (R_d * lapse));
  }
  return p_trop * exp(-grav * (z - z_trop) / (R_d * T_trop));
}
double h_supercell_relhum(double z, double, double z_trop) {                    // :1181-1187
  if (z <= z_trop) return 1.0 - 0.75 * pow(z / z_trop, 1.25);
  return 0.25;
}
double h_supercell_sat_mix_dry(double press, double T) { return 380 / (press)*exp(17.27 * (T - 273) / (T - 36)); }   // :1191-1193

void h_hydro_const_theta(double z, double grav, double C0, double cp, double p0, double gamma, double rd, double &r, double &t) {   // :1108-1117
  const double theta0 = 300., exner0 = 1.;
  t = theta0;
  double exner = exner0 - grav * z / (cp * theta0);
  double p = p0 * std::pow(exner, (cp / rd));
  double rt = std::pow((p / C0), (1.0 / gamma));
  r = rt / t;
}

const double h_gll5_pts[5] = {-0.50000000000000000000000000000000000000, -0.32732683535398857189914622812342917778,
                              0.00000000000000000000000000000000000000, 0.32732683535398857189914622812342917778,
                              0.50000000000000000000000000000000000000};
const double h_gll5_wts[5] = {0.050000000000000000000000000000000000000, 0.27222222222222222222222222222222222222,
                              0.35555555555555555555555555555555555556, 0.27222222222222222222222222222222222222,
                              0.050000000000000000000000000000000000000};
const double h_gll9_pts[9] = {-0.50000000000000000000000000000000000000, -0.44987899770573007865617262220916897903,
                              -0.33859313975536887672294271354567122536, -0.18155873191308907935537603435432960651,
                              0.00000000000000000000000000000000000000, 0.18155873191308907935537603435432960651,
                              0.33859313975536887672294271354567122536, 0.44987899770573007865617262220916897903,
                              0.50000000000000000000000000000000000000};
const double h_gll9_wts[9] = {0.013888888888888888888888888888888888889, 0.082747680780402762523169860014604152919,
                              0.13726935625008086764035280928968636297, 0.17321425548652317255756576606985914397,
                              0.18575963718820861678004535147392290249, 0.17321425548652317255756576606985914397,
                              0.13726935625008086764035280928968636297, 0.082747680780402762523169860014604152919,
                              0.013888888888888888888888888888888888889};
const double h_gl3_pts[3] = {0.112701665379258311482073460022, 0.500000000000000000000000000000, 0.887298334620741688517926539980};
const double h_gl3_wts[3] = {0.277777777777777777777777777779, 0.444444444444444444444444444444, 0.277777777777777777777777777779};

} // namespace

extern "C" int mw_dycore_init(mw_dycore_t d, int init_data, double *rho_d, double *u, double *v, double *w, double *temp,
                              double *const *tracers) {
  if (!d) MW_FAIL("null handle");
  if (init_data < 0 || init_data > 3) MW_FAIL("ERROR: Invalid init_data");      // :1310
  CouplerPtrs c;
  if (make_coupler_ptrs(d, rho_d, u, v, w, temp, tracers, c)) return 1;
  mw_grid_t &g = d->g;
  g.latitude = 0;                                                                // :1249
  g.bc_x = MW_BC_PERIODIC; g.bc_y = MW_BC_PERIODIC; g.bc_z = MW_BC_WALL;         // :1332-1334, 1340-1342, 1423-1425, 1551-1553
  g.use_immersed = (init_data == MW_DATA_CITY || init_data == MW_DATA_BUILDING); // :1312, 1426, 1554
  d->etime = 0;                                                                  // :1317
  const int nz = g.nz, nens = g.nens, ord = d->ord;               // `ord` GLL points per cell (:1725-1727)
  const double dz = g.zlen / g.nz, dx = g.xlen / g.nx_glob;
  const double h_gll3_pts[3] = {-0.50000000000000000000000000000000000000, 0.00000000000000000000000000000000000000, 0.50000000000000000000000000000000000000};   // TransformMatrices.h:83-88
  const double h_gll3_wts[3] = {0.16666666666666666666666666666666666667, 0.66666666666666666666666666666666666667, 0.16666666666666666666666666666666666667};   // :90-95
  const double h_gll7_pts[7] = MW_GLL7_PTS, h_gll7_wts[7] = MW_GLL7_WTS;
  const double *h_gllN_pts = (ord == 3) ? h_gll3_pts : (ord == 7) ? h_gll7_pts : (ord == 9) ? h_gll9_pts : h_gll5_pts;
  const double *h_gllN_wts = (ord == 3) ? h_gll3_wts : (ord == 7) ? h_gll7_wts : (ord == 9) ? h_gll9_wts : h_gll5_wts;
  size_t nzc = (size_t)nz * nens, nze = (size_t)(nz + 1) * nens;
  double *hyc = d->hy_host.data(), *hytc = hyc + nzc, *hye = hyc + 2 * nzc, *hyte = hye + nze;
  std::vector<double> gllcols;     // supercell: hyDensGLL | hyDensThetaGLL | hyDensVapGLL, each (nz,5)
  InitP q;  memset(&q, 0, sizeof(q));
  q.init_data = init_data; q.i_beg = g.i_beg; q.j_beg = g.j_beg; q.xlen = g.xlen; q.ylen = g.ylen; q.cp_d = g.cp_d; q.p0 = g.p0;
  q.nx_glob = g.nx_glob; q.ny_glob = g.ny_glob; q.ord = ord;
  std::vector<double> bheights;
  if (init_data == MW_DATA_SUPERCELL) {                                          // init_supercell, :1687-1840
    const double z_0 = 0, z_trop = 12000, T_0 = 300, T_trop = 213, T_top = 213, p_0 = 100000;
    const double R_d = g.R_d, R_v = g.R_v, grav = g.grav, gamma = g.gamma_d, C0 = g.C0, ztop = g.zlen;
    std::vector<double> quad_temp((size_t)nz * (ord - 1) * ord), hyP((size_t)nz * ord);
    gllcols.assign((size_t)3 * nz * ord, 0.0);
    double *hyDensGLL = gllcols.data(), *hyDensThetaGLL = hyDensGLL + (size_t)nz * ord, *hyDensVapGLL = hyDensThetaGLL + (size_t)nz * ord;
    for (int k = 0; k < nz; k++) for (int kk = 0; kk < ord - 1; kk++) for (int kkk = 0; kkk < ord; kkk++) {       // :1736-1756
      double cellmid = (k + 0.5) * dz;
      double ord_b = cellmid + h_gllN_pts[kk] * dz, ord_t = cellmid + h_gllN_pts[kk + 1] * dz;
      double ord_m = 0.5 * (ord_b + ord_t);
      double ord_dz = dz * (h_gllN_pts[kk + 1] - h_gllN_pts[kk]);
      double zloc = ord_m + ord_dz * h_gllN_pts[kkk];
      double T = h_supercell_temperature(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top);
      double press_dry = h_supercell_pressure_dry(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top, p_0, R_d, grav);
      double qvs = h_supercell_sat_mix_dry(press_dry, T);
      double relhum = h_supercell_relhum(zloc, z_0, z_trop);
      if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
      double qv = std::min(0.014, qvs * relhum);
      quad_temp[((size_t)k * (ord - 1) + kk) * ord + kkk] = -(1 + qv) * grav / (R_d + qv * R_v) / T;
    }
    hyP[0] = p_0;                                                                                                   // :1759-1774
    for (int k = 0; k < nz; k++) for (int kk = 0; kk < ord - 1; kk++) {
      double tot = 0;
      for (int kkk = 0; kkk < ord; kkk++) tot += quad_temp[((size_t)k * (ord - 1) + kk) * ord + kkk] * h_gllN_wts[kkk];
      tot *= dz * (h_gllN_pts[kk + 1] - h_gllN_pts[kk]);
      hyP[(size_t)k * ord + kk + 1] = hyP[(size_t)k * ord + kk] * exp(tot);
      if (kk == ord - 2 && k < nz - 1) hyP[(size_t)(k + 1) * ord] = hyP[(size_t)k * ord + ord - 1];
    }
    for (int k = 0; k < nz; k++) for (int kk = 0; kk < ord; kk++) {                                               // :1777-1805
      double zloc = (k + 0.5) * dz + h_gllN_pts[kk] * dz;
      double T = h_supercell_temperature(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top);
      double press_tmp = h_supercell_pressure_dry(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top, p_0, R_d, grav);
      double qvs = h_supercell_sat_mix_dry(press_tmp, T);
      double relhum = h_supercell_relhum(zloc, z_0, z_trop);
      if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
      double qv = std::min(0.014, qvs * relhum);
      double press = hyP[(size_t)k * ord + kk];
      double dens_dry = press / (R_d + qv * R_v) / T;
      double dens_vap = qv * dens_dry;
      double dens = dens_dry + dens_vap;
      double dens_theta = pow(press / C0, 1.0 / gamma);
      hyDensGLL[(size_t)k * ord + kk] = dens; hyDensThetaGLL[(size_t)k * ord + kk] = dens_theta; hyDensVapGLL[(size_t)k * ord + kk] = dens_vap;
      if (kk == 0) for (int e = 0; e < nens; e++) { hye[(size_t)k * nens + e] = dens; hyte[(size_t)k * nens + e] = dens_theta; }
      if (k == nz - 1 && kk == ord - 1) for (int e = 0; e < nens; e++) { hye[(size_t)(k + 1) * nens + e] = dens; hyte[(size_t)(k + 1) * nens + e] = dens_theta; }
    }
    for (int k = 0; k < nz; k++) {                                                                                  // :1808-1840
      double dens_tot = 0, dens_theta_tot = 0;
      for (int kk = 0; kk < ord; kk++) { dens_tot += hyDensGLL[(size_t)k * ord + kk] * h_gllN_wts[kk];
                                         dens_theta_tot += hyDensThetaGLL[(size_t)k * ord + kk] * h_gllN_wts[kk]; }
      for (int e = 0; e < nens; e++) { hyc[(size_t)k * nens + e] = dens_tot; hytc[(size_t)k * nens + e] = dens_theta_tot; }
    }
  } else {
    bool use_hydro = (init_data == MW_DATA_THERMAL) || g.enable_gravity;
    if (use_hydro) {                                                             // :1396-1419, 1516-1541, 1620-1645
      const int nq = (init_data == MW_DATA_THERMAL) ? 3 : 9;
      const double *qp = (init_data == MW_DATA_THERMAL) ? h_gl3_pts : h_gll9_pts;
      const double *qw = (init_data == MW_DATA_THERMAL) ? h_gl3_wts : h_gll9_wts;
      for (int k = 0; k < nz; k++) for (int e = 0; e < nens; e++) {
        hyc[(size_t)k * nens + e] = 0.; hytc[(size_t)k * nens + e] = 0.;
        for (int kk = 0; kk < nq; kk++) {
          double z = (k + 0.5) * dz + (qp[kk] - 0.5) * dz;
          double hr, ht;
          h_hydro_const_theta(z, g.grav, g.C0, g.cp_d, g.p0, g.gamma_d, g.R_d, hr, ht);
          hyc[(size_t)k * nens + e] += hr * qw[kk];
          hytc[(size_t)k * nens + e] += hr * ht * qw[kk];
        }
      }
      for (int k = 0; k < nz + 1; k++) for (int e = 0; e < nens; e++) {
        double z = k * dz, hr, ht;
        h_hydro_const_theta(z, g.grav, g.C0, g.cp_d, g.p0, g.gamma_d, g.R_d, hr, ht);
        hye[(size_t)k * nens + e] = hr; hyte[(size_t)k * nens + e] = hr * ht;
      }
    } else {                                                                     // :1542-1547, 1646-1651
      for (size_t n = 0; n < nzc; n++) { hyc[n] = 1.15; hytc[n] = 1.15 * 300; }
      for (size_t n = 0; n < nze; n++) { hye[n] = 1.15; hyte[n] = 1.15 * 300; }
    }
    if (init_data == MW_DATA_CITY) {                                             // :1429-1452
      int building_length = 30;
      q.cells_per_building = (int)std::round(building_length / dx);
      q.buildings_pad = 20;
      q.nblocks_x = (static_cast<int>(g.xlen) / building_length - 2 * q.buildings_pad) / 3;
      q.nblocks_y = (static_cast<int>(g.ylen) / building_length - 2 * q.buildings_pad) / 9;
      q.nbx = q.nblocks_x * 3; q.nby = q.nblocks_y * 9;
      if (q.cells_per_building < 1) MW_FAIL("city init: dx too coarse for 30 m buildings");
      bheights.assign((size_t)std::max(1, q.nbx * q.nby), 0.0);
      std::mt19937 gen{17};
      std::normal_distribution<> dist{60, 10};
      for (int j = 0; j < q.nby; j++) for (int i = 0; i < q.nbx; i++) bheights[(size_t)j * q.nbx + i] = dist(gen);
    }
  }
  if (upload_background(d)) return 1;
  fill_params(d);
  double *dev_cols = nullptr, *dev_bh = nullptr;
  if (!gllcols.empty()) {
    MW_HIP(hipMalloc(&dev_cols, gllcols.size() * 8));
    MW_HIP(hipMemcpy(dev_cols, gllcols.data(), gllcols.size() * 8, hipMemcpyHostToDevice));
    q.hyDensGLL = dev_cols; q.hyDensThetaGLL = dev_cols + (size_t)nz * ord; q.hyDensVapGLL = dev_cols + (size_t)2 * nz * ord;
  }
  if (!bheights.empty()) {
    MW_HIP(hipMalloc(&dev_bh, bheights.size() * 8));
    MW_HIP(hipMemcpy(dev_bh, bheights.data(), bheights.size() * 8, hipMemcpyHostToDevice));
    q.bheights = dev_bh;
  }
  MW_HIP(hipMemsetAsync(d->imm, 0, (size_t)d->p.nC * 8, d->stream));                    // :1315
  const DyP &p = d->p;
  MW_KLAUNCH(k_init_cells, plane_grid((long long)p.ny * p.nx * p.nens, p.nz), dim3(256), 0, d->stream, p, q, c, d->imm);
  MW_LAUNCH_CHECK();
  MW_HIP(hipStreamSynchronize(d->stream));
  if (dev_cols) (void)hipFree(dev_cols);
  if (dev_bh) (void)hipFree(dev_bh);
  // the six flux arrays start at zero (:1677-1682)
  MW_HIP(hipMemsetAsync(d->FX, 0, (size_t)p.V * p.fxV * 8, d->stream));
  MW_HIP(hipMemsetAsync(d->FY, 0, (size_t)p.V * p.fyV * 8, d->stream));
  MW_HIP(hipMemsetAsync(d->FZ, 0, (size_t)p.V * p.fzV * 8, d->stream));
  return 0;
}

// =====================================================================================================
// mw_march.h -- the production RK-stage kernels ("shared reconstruction, marching").   Included by mw_dycore.hip.
//
// Each cell is reconstructed ONCE per direction and variable (24 WENO calls per cell and stage for V = 8, like the
// reference's D6 kernel) and its two edge values are handed to the two adjacent faces without touching memory:
//   * along the marching direction (y in the k_y_* kernels, z in the k_xz_* / fused kernels) a thread walks a chunk of cells with a 5-deep
//     register window per variable; the previous cell's upper edge values and face fluxes are carried in registers;
//   * along x a wavefront spans 64 consecutive (x,ens) lanes; stencil neighbours, the west cell's east-edge
//     values and the east face's fluxes travel by wavefront shuffles (ds_bpermute), 3*nens lanes of overlap per side.
// k_y_state / k_y_tracers : y faces -> y part of the state tendencies (scratch, 5 doubles/cell), the upwind mass flux and
//              selector of every y face, and the tracer y-fluxes.
// k_xz_state : x and z faces of the state variables -> COMPLETE state tendencies -> SSPRK3 combine -> new state slab.
//              The state-variable fluxes (15 of the 24 flux doubles per cell) never go to HBM.
// k_tracers_fused + k_tracer_patch : x and z tracer faces + FCT + divergence + SSPRK3 combine (+ D13 on the last stage);
//              the x/z tracer fluxes never go to HBM either.  (k_xz_tracers + k_tracer_update: the unfused form.)
// reference: dynamics_euler_stratified_wenofv.h:271-388 (D6), :395-485 (D9), :519-551 (D11), :121-174 (D12).
// =====================================================================================================
#pragma once
#ifndef MW_XCD_SWIZZLE
#define MW_XCD_SWIZZLE 1
#endif

namespace mw {

__device__ __forceinline__ double shfl_from(double v, int src_lane) { return __shfl(v, src_lane, 64); }

// 1/x to full fp64 accuracy: v_rcp_f64 + two Newton steps (5 instructions instead of the ~12 of an IEEE division)
__device__ __forceinline__ double fast_rcp(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}

// vmcnt counts loads AND stores on gfx9-family parts and retires them in issue order, and the compiler's counted waits assume the
// fewest stores on any path (stores sit in divergent blocks that a wave may skip).  A load waited for AFTER a store therefore
// waits for that store's write acknowledgement.  The marching kernels issue a level's loads at the top of the iteration and its
// stores at the bottom; landed() names the loaded values in front of the first store, which puts the (by then free) wait
// there, and in front of the loop for the prologue's loads, whose pending state would otherwise leak into every iteration.
// A fence for the instruction scheduler.  With its run-time switches folded (Cf<K>, K != 0) a marching loop body loses the rare
// branches that used to cut it into basic blocks and becomes one block of ~500 instructions, which the scheduler then re-orders
// as a whole -- it sank the loads of k_tracers_fused towards their uses and the kernel became 8 % SLOWER than the unspecialised
// one (measured, round 3).  The fences restore the intended order: all loads of a level first, register-only arithmetic behind.
#define MW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// (Measured, round 3: non-temporal stores / loads (__builtin_nontemporal_*) for the write-once / read-once intermediates between two
//  launches -- y tendencies, face mass fluxes, tracer y fluxes, q^n -- change nothing: 5.55-5.70 ms either way.)

template <int N>
__device__ __forceinline__ void landed(double (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("" : "+v"(a[i]));
}
__device__ __forceinline__ void landed(double &a) { asm volatile("" : "+v"(a)); }

// Whole-wavefront shifts by one lane as DPP moves (v_mov_b32_dpp wave_shr:1 / wave_shl:1, gfx9 family): two full-rate
// VALU moves per double and no LDS round trip.  Used when nens == 1 (x neighbours are adjacent lanes).
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);      // bound_ctrl: the edge lane reads 0, so no "old value" mov is needed
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// The same shift, but the edge lane (no source lane) keeps `old` instead of reading 0.
template <int CTRL> __device__ __forceinline__ double dpp_mov_old(double old, double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// value held by the lane n lanes to the west (lower x) / east (higher x)
template <bool N1> __device__ __forceinline__ double from_west(double v, int lane, int n) {
  if (N1) return dpp_mov<0x138>(v);            // wave_shr:1
  return shfl_from(v, lane - n);
}
template <bool N1> __device__ __forceinline__ double from_east(double v, int lane, int n) {
  if (N1) return dpp_mov<0x130>(v);            // wave_shl:1
  return shfl_from(v, lane + n);
}

// Logical (x, y) block of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (observed, not promised: a speed
// matter only), so physical blocks b and b+8 share an L2; the bijective remap below makes the blocks of one XCD logical
// NEIGHBOURS -- adjacent x tiles / row groups, whose 512-byte wave loads overlap by one 128-byte line (the 58-cell tiles are
// not line-aligned) and whose halo columns and boundary flux rows coincide -- so that the second reader finds the line in L2.
struct BlockXY { unsigned x, y; };
__device__ __forceinline__ BlockXY xcd_block() {
#if MW_XCD_SWIZZLE
  const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const unsigned l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  BlockXY b; b.y = l / gridDim.x; b.x = l - b.y * gridDim.x; return b;
#else
  BlockXY b; b.x = blockIdx.x; b.y = blockIdx.y; return b;
#endif
}

// Which (column, [a, b)) SEGMENT of the marching direction a workgroup processes: blockIdx.x = column (XCD-swizzled), blockIdx.y = chunk of
// `chunk` cells.  Every chunk pays its ghost iterations (2 / 4 / 2 for k_xz_state / k_tracers_fused / k_y_all).  (Round 4 measured a balanced
// alternative -- one-dimensional launches whose workgroups are a longest-first list of whole columns and equal slices -- at k_y_all +24 %,
// k_xz_state +1.5 %, k_tracers_fused -2.5 %: no gain; it left the tree in round 6, docs/rounds/DESIGN_rounds1-5.md section 0c.)
struct Segment { unsigned col; int a, b; };
__device__ __forceinline__ void block_segment(int chunk, int len, Segment &sg) {
  const BlockXY blk = xcd_block(); sg.col = blk.x; sg.a = (int)blk.y * chunk; sg.b = min(sg.a + chunk, len);
}

// Periodic direction owned by one rank (DyP::wrap_x / wrap_y): the interior index that a halo index stands for.
__device__ __forceinline__ int wrap_xq(const DyP &p, int q, int NXI) { return p.wrap_x ? (q < 0 ? q + NXI : (q >= NXI ? q - NXI : q)) : q; }
__device__ __forceinline__ int wrap_row(const DyP &p, int j) { return p.wrap_y ? (j < 0 ? j + p.ny : (j >= p.ny ? j - p.ny : j)) : j; }

// Level kl of a marching column (kl may lie in the z halo, -3..nz+2) with the z boundary rule (:752-781) applied on the fly: halo
// levels repeat the nearest interior level, w is 0 behind a wall.  The marching kernels therefore never read the slab's z halo
// and the production path does not fill it (k is wave-uniform: the clamp is scalar work).
template <int K = 0>
__device__ __forceinline__ double load_zlevel(const DyP &p, const double *__restrict__ colv, int kl, bool is_w) {
  const int kc = min(max(kl, 0), p.nz - 1);
  double val = colv[(long long)(kc + p.HZ) * p.sK];
  if (is_w && kc != kl && Cf<K>::z_wall(p)) val = 0.0;
  return val;
}

#ifndef MW_ZERO_SKIP
#define MW_ZERO_SKIP 1
#endif
// Tracers that can be identically zero over large parts of a domain (see the zero short-cut of k_tracers_fused): everything but the
// water vapour of the supercell set-ups; simple_city's vapour IS zero (its only tracer); with the switches at run time: all of them.
// (DyP::zero_skip = the handle's option zero_skip: 0 switches the short-cut off at run time -- the bitwise A/B of tests/ and tools/)
template <int K> __device__ __forceinline__ bool tracer_may_vanish(const DyP &p, int v) { return p.zero_skip && (K == 1 ? v != 0 : true); }

struct FaceState {   // what the passive variables need from the Riemann solve of one face
  double m_upw;      // upwind mass flux (= flux of idR)
  int ind;           // 0: upwind is the low ("L") side, 1: the high ("R") side
};

// Acoustic-upwind Riemann solve for the three primary variables of one face (:399-414).  Edge values:
// rX = rho edge (perturbation + hy), uX = normal velocity edge, eTX = (rho theta)' edge (perturbation only).
// Returns the fluxes of idR, the normal momentum and idT, plus the upwind selector.
template <int K = 0>
__device__ __forceinline__ FaceState riemann_primary(const DyP &p, double rL, double rR, double uL, double uR, double eTL,
                                                     double eTR, double hyt, double p0, double ihyt, bool zero_nrm,
                                                     double &f_nrm, double &f_T) {
#pragma clang fp contract(fast)
  const double cs = 350;
  double mL = zero_nrm ? 0.0 : uL * rL;
  double mR = zero_nrm ? 0.0 : uR * rR;
  double p_L, p_R;
  pressure_fast_pair<K>(p, eTL, eTR, hyt, p0, ihyt, p_L, p_R);
  double w1 = 0.5 * (p_R - cs * mR);
  double w2 = 0.5 * (p_L + cs * mL);
  double p_upw = w1 + w2;
  FaceState fs;
  fs.m_upw = (w2 - w1) * (1.0 / 350.0);
  // (an explicit, unfused sum: under contract(fast) the compiler may turn mL + mR into fma(uL, rL, mR) or fma(uR, rR, mL), and WHICH one
  //  depended on the instantiation -- k_state_xyz and k_y_all chose differently -- which flips the selector where the two momenta
  //  cancel to rounding (v = 0 in a y-symmetric run).  The reference adds the two rounded products, :408.)
  fs.ind = (__dadd_rn(mL, mR) > 0) ? 0 : 1;
  double r_upw = fs.ind ? rR : rL;
  double u_upw = zero_nrm ? 0.0 : (fs.ind ? uR : uL);
  f_nrm = fs.m_upw * u_upw + p_upw;
  f_T = fs.m_upw * ((fs.ind ? eTR : eTL) + hyt) * fast_rcp(r_upw);
  return fs;
}

// ---------------------------------------------------------------------------------------------------------------
// The work of one stage is split by variable group so that every kernel keeps its register windows small enough
// for >= 2-3 wavefronts per SIMD (a single 8-variable kernel needs ~320 VGPR+AGPR = 1 wave/SIMD):
//   state kernels  (rho', u, v, w, (rho theta)'): Riemann solve, publish the upwind mass flux m_upw (it IS the public
//                  state_flux_*(idR) array) and the upwind selector (1 byte per face), finish the state variables;
//   tracer kernels (q_t): flux = m_upw * upwind edge value -> the public tracers_flux_* arrays.
// ---------------------------------------------------------------------------------------------------------------

// y boundary rule (:1061-1081): 0 none, 1: L := R (low wall/open), 2: R := L (high), 3: quirk 1 (R = row 0's south edge)
template <int K = 0>
__device__ __forceinline__ int bc_mode_y(const DyP &p, int j) {
  if (Cf<K>::y_periodic(p)) return 0;
  if (p.py == 0) { if (j == 0) return 1; if (j == p.ny && p.nproc_y == 1) return 3; return 0; }
  if (p.py == p.nproc_y - 1 && j == p.ny) return 2;
  return 0;
}
template <int K = 0>
__device__ __forceinline__ int bc_mode_x(const DyP &p, int i) {                      // :1040-1060
  if (Cf<K>::x_periodic(p)) return 0;
  if (p.px == 0) { if (i == 0) return 1; if (i == p.nx && p.nproc_x == 1) return 3; return 0; }
  if (p.px == p.nproc_x - 1 && i == p.nx) return 2;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// D1 + D2 for the production path (convert_coupler_to_dynamics :1955-2015 with the stage-1 divide :248-255), fast arithmetic:
// rho theta = (p/C0)^(1/gamma) is written as hy_rho_theta(k) (1 + delta)^(1/gamma) with delta = p/p_hy(k) - 1 -- the inverse of
// the Riemann solver's pressure series: 10 FMAs instead of the ~230 instructions of pow, and the stored PERTURBATION
// hyt ((1+delta)^(1/gamma) - 1) comes out without the cancellation of pow(...) - hyt.  |delta| > 0.05 or a non-default gamma: pow.
// C(1/gamma, n), n = 10..1, for gamma = 1003/716 (the long-double recurrence of fill_params as hex literals).
// ---------------------------------------------------------------------------------------------------------------
// out of line (as pressure_pow): the pow body must not be inlined into k_y_state, which sits at the register limit
__device__ __attribute__((noinline)) double rhotheta_ratio_pow(double press_over_C0, double inv_gamma, double hyt) {
  return pow(press_over_C0, inv_gamma) / hyt - 1.0;
}
__device__ __forceinline__ double inv_gamma_series_default(double dl) {
#pragma clang fp contract(fast)
  double acc = -0x1.327f77d85aeeep-8;
  acc = acc * dl + 0x1.71e467e9895fp-8;
  acc = acc * dl + -0x1.c8e61cbaa3102p-8;
  acc = acc * dl + 0x1.22bbebca1e45p-7;
  acc = acc * dl + -0x1.80febd2957c9ap-7;
  acc = acc * dl + 0x1.0d783d4c75011p-6;
  acc = acc * dl + -0x1.9a025de3c9f2fp-6;
  acc = acc * dl + 0x1.66b0e7bdc9cadp-5;
  acc = acc * dl + -0x1.a255770e765c3p-4;
  acc = acc * dl + 0x1.6d7ed9f857ccfp-1;
  return acc;
}
// Index into a COUPLER-layout array (nz,ny,nx,nens) from an index `ci` of the handle's internal cell numbering.  Normally the two
// coincide (cst = 1, ce = 0).  Member-major mode (nens > 1, production path): the handle's arrays hold one member after the other and
// every kernel runs its nens = 1 form on member ce; the coupler's arrays keep their member-fastest layout: cst = nens doubles apart.
__device__ __forceinline__ long long cpl(const DyP &p, long long ci) { return ci * p.cst + p.ce; }

// One cell of the coupler, as loaded (the marching kernel requests row j+3 at the top of iteration j and converts it at the end).
struct CouplerCell { double rho_d, u, v, w, temp, tr[4]; };
// (ldmask: wave-uniform; bit tr clear = tracer tr is known to be zero in this row -- the zero-row maps -- and is not loaded)
template <int K = 0>
__device__ __forceinline__ CouplerCell load_coupler_cell(const DyP &p, const CouplerPtrs &c, long long ci, unsigned ldmask = ~0u) {
  CouplerCell r;
  r.rho_d = c.rho_d[ci]; r.u = c.u[ci]; r.v = c.v[ci]; r.w = c.w[ci]; r.temp = c.temp[ci];
#pragma unroll
  for (int tr = 0; tr < 4; tr++) r.tr[tr] = 0.0;
  if (ldmask == ~0u) {
#pragma unroll
    for (int tr = 0; tr < 4; tr++) if (tr < Cf<K>::ntr(p)) r.tr[tr] = c.tr[tr][ci];
  } else {
#pragma unroll
    for (int tr = 0; tr < 4; tr++) if (tr < Cf<K>::ntr(p) && ((ldmask >> tr) & 1u)) r.tr[tr] = c.tr[tr][ci];
  }
  return r;
}
// -> the five state variables of the slab (rho', u, v, w, (rho theta)') and 1/rho for the tracers.  hi = k*nens + e.  The last
// operation of every result is not contractable, so that a caller that goes on computing with the values sees exactly what is stored.
// (hyc, hyt, p0: the cell's background density, rho*theta and pressure = DyP::hyc / hytc / p0c [hi] = the first three of hypk's row)
template <int K = 0>
__device__ __forceinline__ void convert_cell_fast(const DyP &p, const CouplerCell &r, double hyc, double hyt, double p0, double *s5, double &inv_den) {
  double rho, ru, rv, rw, sd, rp;
  {
#pragma clang fp contract(fast)
    rho = r.rho_d;
    double rho_v = 0;
#pragma unroll
    for (int tr = 0; tr < 4; tr++) {
      if (tr < Cf<K>::ntr(p) && Cf<K>::adds_mass(p, tr)) rho += r.tr[tr];
      if (Cf<K>::is_wv(p, tr)) rho_v = r.tr[tr];
    }
    const double press = r.rho_d * p.R_d * r.temp + rho_v * p.R_v * r.temp;
    const double dl = press * fast_rcp(p0) - 1.0;
    if (fabs(dl) <= 0.05 && Cf<K>::an_default(p)) sd = inv_gamma_series_default(dl) * dl;
    else                                  sd = rhotheta_ratio_pow(press / p.C0, 1.0 / p.gamma, hyt);
    rp = rho - hyc;                                            // state(idR)
    inv_den = fast_rcp(rp + hyc);                              // what D2 divides by (:249)
    ru = rho * r.u; rv = rho * r.v; rw = rho * r.w;
  }
  {
#pragma clang fp contract(off)
    s5[idR] = rp; s5[idU] = ru * inv_den; s5[idV] = rv * inv_den; s5[idW] = rw * inv_den; s5[idT] = hyt * sd;
  }
}
__device__ __forceinline__ void convert_cell_fast(const DyP &p, const CouplerCell &r, int hi, double *s5, double &inv_den) {
  convert_cell_fast<0>(p, r, p.hyc[hi], p.hytc[hi], p.p0c[hi], s5, inv_den);
}
// the tracers of that cell: slab value = rho_t / rho
template <int K = 0>
__device__ __forceinline__ void convert_cell_tracers(const DyP &p, const CouplerCell &r, double inv_den, double *__restrict__ s, long long sV) {
#pragma clang fp contract(off)
#pragma unroll
  for (int tr = 0; tr < 4; tr++) if (tr < Cf<K>::ntr(p)) s[(long long)(5 + tr) * sV] = r.tr[tr] * inv_den;
}
// stand-alone form (2-D runs, walls / open boundaries or a neighbour exchange in y, two-stream schedule)
// (ylo, yhi: only the cells with j < ylo, j >= yhi or within HX cells of the block's west / east edge -- the strips that the pipelined
//  multi-rank schedule packs and the rows its edge-strip y launch reads; the rest is converted inside k_y_all<true>.  ylo >= ny: all.)
// (strips != 0: the launch covers the strip cells only -- whole rows j < ylo and j >= yhi first, then the 2 HX west / east columns of the rows
//  between -- instead of all cells with most threads leaving at once: 50 -> a few us on the compute stream of a 400 x 400 x 100 block)
__global__ __launch_bounds__(256) void k_coupler_to_state_fast(DyP p, CouplerPtrs c, double *__restrict__ S, int ylo, int yhi, int strips) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const int NXI = p.nx * p.nens;
  int j, ie;
  if (strips) {
    const int hxn = p.HX * p.nens;
    const long long nA = (long long)(ylo + p.ny - yhi) * NXI, nB = (long long)(yhi - ylo) * 2 * hxn;
    if (t >= nA + nB) return;
    if (t < nA) { const int jj = (int)(t / NXI); ie = (int)(t - (long long)jj * NXI); j = jj < ylo ? jj : yhi + (jj - ylo); }
    else { const long long u = t - nA; const int r = (int)(u % (2 * hxn)); j = ylo + (int)(u / (2 * hxn)); ie = r < hxn ? r : NXI - 2 * hxn + r; }
  } else {
    if (t >= (long long)p.ny * NXI) return;
    j = (int)(t / NXI); ie = (int)(t - (long long)j * NXI);
  }
  if (j >= ylo && j < yhi && ie >= p.HX * p.nens && ie < NXI - p.HX * p.nens) return;
  const long long ci = ((long long)k * p.ny + j) * NXI + ie;
  const CouplerCell r = load_coupler_cell(p, c, cpl(p, ci));
  double s5[5], inv_den;
  convert_cell_fast(p, r, k * p.nens + ie % p.nens, s5, inv_den);
  double *s = S + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
#pragma unroll
  for (int v = 0; v < 5; v++) s[(long long)v * p.sV] = s5[v];
  convert_cell_tracers(p, r, inv_den, s, p.sV);
}

// ---------------------------------------------------------------------------------------------------------------
// Zero-row maps (round 5).  Cloud water and rain are exactly zero over most of a supercell domain (all of it at the start), and after
// the zero short-cut (MW_ZERO_SKIP) the fused tracer kernel is bound by the HBM traffic of values that are all zero.  The maps let it
// not issue those loads at all:
//   M0[k][j], bit v : tracer v is non-zero somewhere in the x row (k, j) of the sub-cycle's input q^n   (k_zero_rows: one pass over the
//                     tracers that can vanish, 16 bytes per cell of the ~500 a stage moves)
//   Qs[k][j], bit v : (s = 1, 2, 3) tracer v may be non-zero in something that iterations k-3 .. k of row j's marching wave touch in RK
//                     stage s: the OR of M0 over rows j-3s .. j+3s (periodic) and levels k-3s-5 .. k+3s+1 (clamped)  (k_zero_dilate)
// One RK stage moves a tracer by at most 3 cells per direction (cell (k, j) is updated from the faces j, j+1 / k, k+1, whose upwind
// WENO-5 stencils reach rows j-3 .. j+3 / levels k-3 .. k+3; an FCT multiplier only scales a flux, the y-face correction of
// k_tracer_patch is a difference of scaled and unscaled flux, and every stage adds q^n's own row), and a flux of zeros is zero: a row
// that is non-zero in the input of stage s (its output, its y fluxes) lies within 3(s-1) (3s) rows and levels of a non-zero row of q^n.
// Iteration k of k_tracers_fused reads input levels k-2 .. k+3, y fluxes of levels k-1 .. k+1 and q^n of levels k-2 .. k+2, and its
// carries come from the three iterations before (input levels from k-5, y fluxes from k-5): with Qs[k][j]'s bit clear all of that is
// exactly zero, and so is the tracer's new value.  The maps are a SUPERSET of the non-zero rows -- a set bit costs the loads, nothing
// else -- and results equal the run without them (tests/: bitwise up to the sign of a zero; a skipped value enters as +0.0).
// Handles: x and y periodic, the fused tracer stage; nens == 1 on one rank or as the blocks of a decomposed domain on the pipelined schedule,
// member-major handles (nens > 1) on one rank with one map set per member (host side: zero_rows_ok / zero_rows_build in mw_dycore.hip).
// ---------------------------------------------------------------------------------------------------------------
#define MW_ZR_REACH 3
#define MW_ZR_HALO (3 * MW_ZR_REACH)                              // rows a tracer can travel in one sub-cycle = halo rows of the maps
// x SEGMENTS (round 6).  A storm covers a fifth of the x rows but a fourteenth of the 58-cell tiles a marching wave works on (bench.py:
// storm.extent), so a word also says WHERE in its row something can be:
//   bits 0 .. 3   tracer v may be non-zero somewhere in the row (as before; the y kernel's store decision, the tests)
//   bits 4 .. 29  segment s of the row -- cells [s L, (s + 1) L), L = ceil(NXI / 26) -- may hold a non-zero value of a tracer that can vanish.
//                 Set at SCAN time for every segment within MW_ZR_HALO = 9 cells of a non-zero cell (what a whole sub-cycle can move it in
//                 x; periodic when one rank owns the direction), so that the row / level growth of k_zero_dilate -- a plain OR of words --
//                 carries the segments along.  A wave is lean when the segments its lanes (halo lanes and patch cells included) overlap are clear.
//   bits 30, 31   (the compact copy a block sends to its west / east neighbours only) the grown set touches the block's west / east edge:
//                 the neighbour then sets the segments of its first / last 9 cells (k_zero_merge).
// Per cell the invariants are the old ones -- "bit clear => the cell is zero" -- so the zero-store rules hold cell by cell whatever the tiling.
#define MW_ZR_SEGS 26
#define MW_ZR_SEG_SHIFT 4
#define MW_ZR_ROWBITS 0xFu
#define MW_ZR_TOUCH_W 0x40000000u
#define MW_ZR_TOUCH_E 0x80000000u
__host__ __device__ __forceinline__ int zr_seg_len(int NXI) { return (NXI + MW_ZR_SEGS - 1) / MW_ZR_SEGS; }
// segment bits (in place: shifted) of the cells [a, b], 0 <= a <= b < NXI
__host__ __device__ __forceinline__ unsigned zr_seg_span(int a, int b, int L) { return ((2u << (b / L)) - (1u << (a / L))) << MW_ZR_SEG_SHIFT; }
// ... of the cells [lo, hi] of a row of NXI cells (lo may be < 0, hi >= NXI): periodic images when `wrap`, else clamped into the row
__host__ __device__ __forceinline__ unsigned zr_seg_mask(int lo, int hi, int NXI, bool wrap) {
  const int L = zr_seg_len(NXI);
  if (hi - lo + 1 >= NXI) return zr_seg_span(0, NXI - 1, L);
  if (!wrap) return zr_seg_span(max(lo, 0), min(hi, NXI - 1), L);
  if (lo < 0) return zr_seg_span(lo + NXI, NXI - 1, L) | zr_seg_span(0, hi, L);
  if (hi >= NXI) return zr_seg_span(lo, NXI - 1, L) | zr_seg_span(0, hi - NXI, L);
  return zr_seg_span(lo, hi, L);
}
// Map layout: word of (k, j) at [k * zq_ld + j + MW_ZR_HALO], zq_ld = ny + 2 MW_ZR_HALO (j = -9 .. ny+8: the rows beyond the block's
// south / north edge -- its own rows again when it owns the periodic y direction, the neighbours' rows otherwise).
// Blocks of a decomposed domain (pipelined schedule): a row's x halo holds the west / east neighbour's cells and its tracer can come
// from there, so M0 is OR-ed with the neighbours' M0 of the same row (k_zero_merge; whole rows: coarse, but a superset), and the rows
// beyond the y edges come from the south / north neighbours' merged maps -- two small messages per sub-cycle through the handle's
// halo transport, on the exchange stream beside the stage's y and x/z kernels.
// k_zero_rows: wave = one x row (k, j).  SLAB: the tracers of slab S (later sub-cycles of a step, or after a conversion pass); else the
// coupler's arrays.  vmask: the tracers that can vanish -- the others are flagged non-zero without being read.  ldo / offo: layout of
// `out` (compact [k][ny] for k_zero_merge, or the map's own); wrap: also write the row's periodic images into the map's halo rows.
template <bool SLAB>
// (wrap == 2: the halo rows are marked "may be non-zero" instead -- the LOCAL maps of a decomposed block, used by the first stage's y launches
//  before the neighbours' maps have arrived; out2: a second, compact [k][ny] copy for k_zero_merge)
__global__ __launch_bounds__(256) void k_zero_rows(DyP p, CouplerPtrs c, const double *__restrict__ S, unsigned *__restrict__ out, unsigned vmask,
                                                   int ldo, int offo, int wrap, unsigned *__restrict__ out2) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)p.nz * p.ny) return;
  const int k = (int)(row / p.ny), j = (int)(row - (long long)k * p.ny), lane = threadIdx.x & 63;
  const int NXI = p.nx * p.nens;
  // (all tracers' values of eight 64-cell pieces are requested before the first one is tested: the kernel is a latency-bound stream otherwise)
  const double *src[4];
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int vv = min(v, p.nt - 1);
    src[v] = SLAB ? S + (long long)(5 + vv) * p.sV + (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens
                  : c.tr[vv] + ((long long)k * p.ny + j) * NXI;
  }
  const unsigned scan = vmask & ((1u << min(p.nt, 4)) - 1u);
  bool nz[4] = {false, false, false, false};
  unsigned segb = 0;                                              // this lane's part of the segment bits (+ the edge flags)
  const bool xwrap = p.wrap_x != 0;
  for (int base = 0; base < NXI; base += 512) {
    double a[4][8];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      if (!((scan >> v) & 1u)) continue;                        // (wave-uniform)
#pragma unroll
      for (int u = 0; u < 8; u++) { const int ie = base + u * 64 + lane; a[v][u] = src[v][min(ie, NXI - 1)]; }
    }
    bool cellnz[8] = {false, false, false, false, false, false, false, false};
#pragma unroll
    for (int v = 0; v < 4; v++) {
      if (!((scan >> v) & 1u)) continue;
#pragma unroll
      for (int u = 0; u < 8; u++) { const bool b = (a[v][u] != 0.0); nz[v] = nz[v] || b; cellnz[u] = cellnz[u] || b; }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (!cellnz[u]) continue;
      const int ie = min(base + u * 64 + lane, NXI - 1);
      segb |= zr_seg_mask(ie - MW_ZR_HALO, ie + MW_ZR_HALO, NXI, xwrap);
      if (!xwrap && ie - MW_ZR_HALO < 0) segb |= MW_ZR_TOUCH_W;
      if (!xwrap && ie + MW_ZR_HALO >= NXI) segb |= MW_ZR_TOUCH_E;
    }
  }
  for (int off = 32; off > 0; off >>= 1) segb |= (unsigned)__shfl_xor((int)segb, off, 64);
  unsigned word = ((1u << min(p.nt, 4)) - 1u) & ~vmask;           // the tracers that cannot vanish
#pragma unroll
  for (int v = 0; v < 4; v++) if (((scan >> v) & 1u) && __any(nz[v])) word |= 1u << v;
  word |= segb;                                                  // (with the two edge flags when x is decomposed: k_zero_merge of the neighbours reads them; nobody else looks at bits 30, 31)
  if (lane == 0) {
    unsigned *o = out + (long long)k * ldo + offo;
    o[j] = word;
    if (wrap && j < MW_ZR_HALO) o[p.ny + j] = (wrap == 2) ? ~0u : word;
    if (wrap && j >= p.ny - MW_ZR_HALO) o[j - p.ny] = (wrap == 2) ? ~0u : word;
    if (out2) out2[(long long)k * p.ny + j] = word;
  }
}
// Decomposed block: M = own rows (compact) OR the west / east neighbours' (rW / rE, nullptr: x is not decomposed); the merged rows next
// to the south / north edge are packed for those neighbours (sS / sN, nullptr: y is not decomposed -- the halo rows are then the block's
// own periodic images).
__global__ __launch_bounds__(256) void k_zero_merge(DyP p, const unsigned *__restrict__ own, const unsigned *__restrict__ rW, const unsigned *__restrict__ rE,
                                                    unsigned *__restrict__ M, unsigned *__restrict__ sS, unsigned *__restrict__ sN) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.nz * p.ny) return;
  const int k = (int)(t / p.ny), j = (int)(t - (long long)k * p.ny);
  unsigned wd = own[t];
  if (rW) {
    // the neighbours' rows: their row bits as before (coarse, but a superset -- the y kernel's store decision looks at them); of their
    // segments only what can cross the shared edge within a sub-cycle: a grown set that touches the neighbour's edge reaches my first / last 9 cells
    const unsigned w = rW[t], e = rE[t];
    const int NXI = p.nx * p.nens;
    wd |= (w | e) & MW_ZR_ROWBITS;
    if (w & MW_ZR_TOUCH_E) wd |= zr_seg_mask(0, MW_ZR_HALO - 1, NXI, false);
    if (e & MW_ZR_TOUCH_W) wd |= zr_seg_mask(NXI - MW_ZR_HALO, NXI - 1, NXI, false);
  }
  unsigned *o = M + (long long)k * p.zq_ld + MW_ZR_HALO;
  o[j] = wd;
  if (sS) {
    if (j < MW_ZR_HALO) sS[k * MW_ZR_HALO + j] = wd;
    if (j >= p.ny - MW_ZR_HALO) sN[k * MW_ZR_HALO + j - (p.ny - MW_ZR_HALO)] = wd;
  } else {
    if (j < MW_ZR_HALO) o[p.ny + j] = wd;
    if (j >= p.ny - MW_ZR_HALO) o[j - p.ny] = wd;
  }
}
// ... and the rows beyond the y edges from what the south / north neighbours packed (their northernmost / southernmost rows)
__global__ __launch_bounds__(256) void k_zero_halo(DyP p, unsigned *__restrict__ M, const unsigned *__restrict__ rS, const unsigned *__restrict__ rN) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= p.nz * MW_ZR_HALO) return;
  const int k = t / MW_ZR_HALO, h = t - k * MW_ZR_HALO;
  unsigned *o = M + (long long)k * p.zq_ld + MW_ZR_HALO;
  o[h - MW_ZR_HALO] = rS[t];
  o[p.ny + h] = rN[t];
}
// M0 -> Q1, Q2, Q3 (Q_s at M + s * msz), separably through LDS: rows first (three radii), then levels; and
//   FNs[k][j] at M + (3 + s) * msz = the OR of Qs[k-1 .. k+1][j]: "row j's tracer kernel may load its y fluxes of level k" (its iteration k'
//   reads level k'-1, clamped into the chunk: k' = k-1, k or k+1) -- k_y_all does not store the fluxes of a face whose two rows are clear.
#define MW_ZR_BEFORE 5                                            // levels below: the three iterations whose carries an iteration inherits
#define MW_ZR_AFTER 1                                             // levels above: the y fluxes of level k+1 (chunk ends clamp k-1 upwards)
//   QYs[k][j] at M + (6 + s) * msz = M0 dilated by 3s rows and 3(s-1) levels = the input of stage s, three more rows either way: what
//   iteration j of k_y_all's wave at level k touches (the window of face j, the row that enters it, the edge value carried from j-1).
// Zeros are not stored over zeros either (DyP::zqc / zqp, host side: zero_rows_build):
//   MC (zqc) = M0 of the time step's FIRST sub-cycle = which rows of the coupler's own tracer arrays are zero (nothing writes them before the
//   last sub-cycle's D13): a lean iteration of the last stage does not store there;
//   the PREVIOUS sub-cycle's Qs (zqp; s = 1, 2 -- slabs S1 and S2 always take the output of stages 1 and 2): level k of a row is stored by
//   iteration k+2, in the lean form (zeros) exactly when Qs[k+2][j] was clear -- for every x tile of the row alike, the form follows from the
//   row's word alone -- so where the previous Qs is clear the slab row holds zeros already and a lean iteration leaves it alone.  The host
//   hands the previous maps over only when the previous sub-cycle ran with maps and nothing else has written the slabs since.
#define MW_ZR_MAPS 10
// (fn_ones: FNs = "store" everywhere -- the local maps of a decomposed block cannot know what the neighbours make the tracer kernel read)
__global__ __launch_bounds__(256) void k_zero_dilate(DyP p, unsigned *__restrict__ M, long long msz, int fn_ones) {
  constexpr int TK = 16, TJ = 64, R = MW_ZR_HALO, LO = R + MW_ZR_BEFORE + 1, HI = R + MW_ZR_AFTER + 1, EK = TK + LO + HI, EJ = TJ + 2 * R;
  __shared__ unsigned a[EK][EJ];
  __shared__ unsigned b[3][EK][TJ];
  const int k0 = blockIdx.y * TK, j0 = blockIdx.x * TJ, tid = threadIdx.x;
  for (int t = tid; t < EK * EJ; t += 256) {
    const int kk = t / EJ, jj = t - kk * EJ;
    const int k = min(max(k0 + kk - LO, 0), p.nz - 1);
    const int j = j0 + jj - R;                                   // -9 .. ny+8 are rows of the map; beyond (the last tile's overhang): nobody's
    a[kk][jj] = (j < p.ny + R) ? M[(long long)k * p.zq_ld + j + MW_ZR_HALO] : 0u;
  }
  __syncthreads();
  for (int t = tid; t < EK * TJ; t += 256) {
    const int kk = t / TJ, jj = t - kk * TJ, cc = jj + R;
    unsigned o = a[kk][cc];
#pragma unroll
    for (int r = 1; r <= R; r++) {
      o |= a[kk][cc - r] | a[kk][cc + r];
      if (r % MW_ZR_REACH == 0) b[r / MW_ZR_REACH - 1][kk][jj] = o;
    }
  }
  __syncthreads();
  for (int t = tid; t < TK * TJ; t += 256) {
    const int kk = t / TJ, jj = t - kk * TJ, k = k0 + kk, j = j0 + jj;
    if (k >= p.nz || j >= p.ny) continue;
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const int r = MW_ZR_REACH * (m + 1);
      const int ry = MW_ZR_REACH * m;
      unsigned o = 0, fn = 0, qy = 0;
      for (int dd = -r - MW_ZR_BEFORE - 1; dd <= r + MW_ZR_AFTER + 1; dd++) {
        const unsigned v = b[m][kk + LO + dd][jj];
        fn |= v;
        if (dd >= -r - MW_ZR_BEFORE && dd <= r + MW_ZR_AFTER) o |= v;
        if (dd >= -ry && dd <= ry) qy |= v;
      }
      M[(long long)(m + 1) * msz + (long long)k * p.zq_ld + j + MW_ZR_HALO] = o;
      M[(long long)(m + 4) * msz + (long long)k * p.zq_ld + j + MW_ZR_HALO] = fn_ones ? ~0u : fn;
      M[(long long)(m + 7) * msz + (long long)k * p.zq_ld + j + MW_ZR_HALO] = qy;
    }
  }
}

// Test aid (option zero_verify, any build): the CLAIMS the maps make, checked against the data right in front of the launches that rely on
// them.  wave = one x row (k, j) of the block; viol[] counts rows:
//   [0] a tracer that can vanish is non-zero in row (k, j) of the stage's INPUT slab although Qs[k'][j] is clear for one of the iterations
//       k' = k-3 .. k+2 (clamped) of the row's tracer wave that touch level k  (k_tracers_fused would not have loaded it);
//   [1] ... although QYs[k][j'] is clear for one of the rows j' = j-3 .. j+3 whose y-marching iteration touches row j  (k_y_all);
//   [2] a destination row whose store of zeros the tracer kernel is about to SKIP (its storing iteration is lean and the map handed over as
//       "the destination holds zeros already" is clear -- zqp: slab S1 / S2; zqc: the coupler's arrays) is NOT all zero: a non-zero value
//       would be left standing;
//   [3] likewise a row of the slab that the converting y launch is about to skip (zqk).
// dst == nullptr / the map pointers == nullptr: that claim is not made in this launch and not checked.
__global__ __launch_bounds__(256) void k_zero_verify(DyP p, CouplerPtrs c, const double *__restrict__ Sin, const double *__restrict__ Sdst, int dst_coupler,
                                                     const double *__restrict__ Skz, long long msz, unsigned vmask, unsigned long long *__restrict__ viol) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)p.nz * p.ny) return;
  const int k = (int)(row / p.ny), j = (int)(row - (long long)k * p.ny), lane = threadIdx.x & 63;
  const int NXI = p.nx * p.nens;
  const unsigned scan = vmask & ((1u << min(p.nt, 4)) - 1u);
  unsigned in_nz = 0, dst_nz = 0, kz_nz = 0;
  unsigned in_sg = 0, dst_sg = 0, kz_sg = 0;                      // ... and the x segments (exact: no growth) that hold a non-zero cell
  const long long so = (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens;
  const long long ci = ((long long)k * p.ny + j) * NXI;
  for (int v = 0; v < 4; v++) {
    if (!((scan >> v) & 1u)) continue;
    bool a = false, b = false, e = false;
    for (int ie = lane; ie < NXI; ie += 64) {
      const unsigned sg = zr_seg_mask(ie, ie, NXI, false);
      if (Sin && Sin[(long long)(5 + v) * p.sV + so + ie] != 0.0) { a = true; in_sg |= sg; }
      if (Sdst && !dst_coupler && Sdst[(long long)(5 + v) * p.sV + so + ie] != 0.0) { b = true; dst_sg |= sg; }
      if (dst_coupler && c.tr[v][cpl(p, ci + ie)] != 0.0) { b = true; dst_sg |= sg; }
      if (Skz && Skz[(long long)(5 + v) * p.sV + so + ie] != 0.0) { e = true; kz_sg |= sg; }
    }
    if (__any(a)) in_nz |= 1u << v;
    if (__any(b)) dst_nz |= 1u << v;
    if (__any(e)) kz_nz |= 1u << v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    in_sg |= (unsigned)__shfl_xor((int)in_sg, off, 64); dst_sg |= (unsigned)__shfl_xor((int)dst_sg, off, 64); kz_sg |= (unsigned)__shfl_xor((int)kz_sg, off, 64);
  }
  // (a segment's claim is a claim about its cells: checked with the same rules as the row bits, the segment bits beside them)
  in_nz |= in_sg; dst_nz |= dst_sg; kz_nz |= kz_sg;
  if (lane != 0) return;
  const long long ld = p.zq_ld;
  if (p.zq && in_nz) {
    bool bad0 = false, bad1 = false;
    for (int kk = max(k - 3, 0); kk <= min(k + 2, p.nz - 1); kk++) bad0 = bad0 || ((in_nz & ~p.zq[(long long)kk * ld + j + MW_ZR_HALO]) != 0u);
    const unsigned *qy = p.zq + 6 * msz;
    // (the derived maps hold the block's own rows only -- the kernels index them behind wrap_row: rows beyond a decomposed y edge belong to
    //  the neighbour's maps and are not checked here)
    for (int dj = -MW_ZR_REACH; dj <= MW_ZR_REACH; dj++) {
      const int jj = wrap_row(p, j + dj);
      if (jj >= 0 && jj < p.ny) bad1 = bad1 || ((in_nz & ~qy[(long long)k * ld + jj + MW_ZR_HALO]) != 0u);
    }
    if (bad0) atomicAdd(&viol[0], 1ull);
    if (bad1) atomicAdd(&viol[1], 1ull);
  }
  // (the store of level k is skipped by the iteration k + 2 of the row's wave -- word min(k + 2, nz - 1) -- when that iteration is LEAN and
  //  the destination's map is clear; the converting y launch skips row j when the iteration that converts it, j - 3, is lean and zqk is clear)
  if (dst_nz && p.zq) {
    const unsigned *m = dst_coupler ? p.zqc : p.zqp;
    const int kq = min(k + 2, p.nz - 1);
    if (m && (dst_nz & ~p.zq[(long long)kq * ld + j + MW_ZR_HALO] & ~m[(long long)(dst_coupler ? k : kq) * ld + j + MW_ZR_HALO])) atomicAdd(&viol[2], 1ull);
  }
  if (kz_nz && p.zqk && p.zq) {
    const unsigned *qy = p.zq + 6 * msz;
    const int jl = wrap_row(p, j - MW_ZR_REACH);                 // the iteration that converts row j; beyond a decomposed edge: never lean
    if (jl >= 0 && jl < p.ny && (kz_nz & ~qy[(long long)k * ld + jl + MW_ZR_HALO] & ~p.zqk[(long long)k * ld + j + MW_ZR_HALO])) atomicAdd(&viol[3], 1ull);
  }
}

// Member-major handles (nens > 1, see View in mw_dycore.hip): the two conversions between the coupler's member-fastest arrays and
// the member-after-member slabs, as coalesced passes.  thread = one cell in the COUPLER's order (fused x, member fastest): the
// coupler side is a unit-stride stream, the slab side 16-lane segments of nens different members.  (Done from inside the
// per-member marching kernels -- as on the nens = 1 path -- every launch would touch 1/nens of every cache line of the coupler's
// arrays, and the partial lines of the four launches do not meet in L2: measured +65 % on k_y_state, +20 % on k_tracers_fused.)
// p = the FUSED parameter block; msV .. mslab = the member view's strides.
struct MemberStrides { long long sJ, sK, sV, slab; };
// (ylo, yhi: as in k_coupler_to_state_fast -- only the strips the pipelined multi-rank schedule needs up front; ylo >= ny: all cells)
__global__ __launch_bounds__(256) void k_coupler_to_member(DyP p, CouplerPtrs c, double *__restrict__ S, MemberStrides m, int ylo, int yhi) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  const int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  if (j >= ylo && j < yhi && ie >= p.HX * p.nens && ie < NXI - p.HX * p.nens) return;
  const int i = ie / p.nens, e = ie - i * p.nens;
  const long long ci = ((long long)k * p.ny + j) * NXI + ie;
  const CouplerCell r = load_coupler_cell(p, c, ci);
  double s5[5], inv_den;
  convert_cell_fast(p, r, k * p.nens + e, s5, inv_den);
  double *s = S + (long long)e * m.slab + (long long)(k + p.HZ) * m.sK + (long long)(j + p.HY) * m.sJ + p.HX + i;
#pragma unroll
  for (int v = 0; v < 5; v++) s[(long long)v * m.sV] = s5[v];
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int tr = 0; tr < 4; tr++) if (tr < p.nt) s[(long long)(5 + tr) * m.sV] = r.tr[tr] * inv_den;
  }
}
// D13 (:1927-1950) from the member-major result slab of the last stage (stored form: rho', u, v, w, (rho theta)', q_t), the same
// arithmetic as the D13 tail of k_tracers_fused<3, 1>: pressure by the series around the hydrostatic state.
__global__ __launch_bounds__(256) void k_member_to_coupler(DyP p, const double *__restrict__ S, CouplerPtrs c, MemberStrides m) {
#pragma clang fp contract(off)
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const int NXI = p.nx * p.nens;
  if (t >= (long long)p.ny * NXI) return;
  const int j = (int)(t / NXI), ie = (int)(t - (long long)j * NXI);
  const int i = ie / p.nens, e = ie - i * p.nens;
  const long long ci = ((long long)k * p.ny + j) * NXI + ie;
  const double *s = S + (long long)e * m.slab + (long long)(k + p.HZ) * m.sK + (long long)(j + p.HY) * m.sJ + p.HX + i;
  const int hi = k * p.nens + e;
  const double rho_new = s[0] + p.hyc[hi];
  double rho_dry = rho_new, rho_v = 0;
#pragma unroll
  for (int tr = 0; tr < 4; tr++) {
    if (tr < p.nt) {
      const double q = s[(long long)(5 + tr) * m.sV] * rho_new;          // the conserved tracer density (:1943)
      c.tr[tr][ci] = q;
      if (tr == p.idWV) rho_v = q;
      if ((p.mass_mask >> tr) & 1u) rho_dry -= q;
    }
  }
  const double press = pressure_fast(p, s[(long long)idT * m.sV], p.hytc[hi], p.p0c[hi], p.ihytc[hi]);
  c.rho_d[ci] = rho_dry;
  c.u[ci] = s[(long long)idU * m.sV]; c.v[ci] = s[(long long)idV * m.sV]; c.w[ci] = s[(long long)idW * m.sV];
  c.temp[ci] = press / (rho_dry * p.R_d + rho_v * p.R_v);
}

// ---------------------------------------------------------------------------------------------------------------
// Y pass, state variables.  thread = (k, interior fused-x lane), marches j over [ja-1, jb] for the chunk [ja, jb).
// CONV (first stage of a step, periodic y owned by one rank): the rows are not read from the slab but converted from the
// coupler's arrays while they are loaded, and the chunk writes the slab rows it owns (state variables and tracers) on the way:
// the separate conversion pass -- 16 arrays read or written -- disappears.
// Writes FY[idR] (= m_upw), UPY (selector) for faces ja..jb and tendY (5,nz,ny,nx,nens) for rows ja..jb-1.
// ---------------------------------------------------------------------------------------------------------------
// (Measured, round 3: with the switches folded the non-converting variant needs 192 VGPRs; capped at 168 for a third wave per SIMD
//  it spills 24 of them.)
// MM (member-major handles, nens > 1; CONV only): ONE launch over the FUSED lanes (lane = (x, member), the coupler's order) reads the
// coupler's arrays as unit-stride streams, and writes its outputs -- slab rows, y tendencies, face mass fluxes -- into the handle's
// member-after-member arrays (16-lane segments of nens different members: whole 128-byte lines at nens = 4).  The y direction does not
// care which lane holds which x cell, so this replaces the separate k_coupler_to_member pass and the members' first k_y_state.
// MM = 2 (2 or 4 members): the per-member form of the launch -- wave = one member's 64 x cells, coalesced stores into that member's
// arrays -- with the nens members of the same cells in ONE workgroup: their 8-byte reads of the coupler's member-fastest arrays, nens
// doubles apart, then meet in the CU's L1 / the XCD's L2 (the same idea as MemberOff below for the way out).
struct MemberOff {
  long long slab, tend, mx, my, mz, fx, fy, fz, cells, per;    // doubles (selectors / flags: bytes) from member e to member e + 1
  long long zq;                                                // ... and words, for the members' zero-row maps
  int n, sh;                                                   // members per workgroup (2 or 4) and log2 of it
};
struct YMember { long long sJ, sK, sV, slab, fyJ, fyK, mfy, nC, tend; int nx; long long per; int n, sh; };
template <bool CONV, int K, int ORD, int MM = 0>
__global__ __launch_bounds__(256, 2) void k_y_state(DyP p, const double *__restrict__ S, double *__restrict__ MY,
                                                 unsigned char *__restrict__ UPY, double *__restrict__ tendY, int chunk,
                                                 CouplerPtrs c, double *__restrict__ Sw, YMember mm) {
  static_assert(!MM || CONV, "the fused-lane form exists for the converting launch only");
  const int NXI = p.nx * p.nens;
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // flattened (k, ie): no idle tail per row
  if (MM == 2) {                                                 // (p = one member's view: nens = 1, cst = the member count)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), em = wv & (mm.n - 1);
    t = ((long long)blockIdx.x * (4 >> mm.sh) + (wv >> mm.sh)) * 64 + (threadIdx.x & 63);
    MY += em * mm.mfy; UPY += em * mm.mfy; tendY += em * mm.tend; Sw += em * mm.slab; S += em * mm.slab; p.hypk += em * mm.per; p.ce = em;
  }
  if (t >= (long long)p.nz * NXI) return;
  const int k = (int)(t / NXI);
  const int ie = (int)(t - (long long)k * NXI);
  // (chunk < 0: the two EDGE strips of -chunk rows at the block's south and north end -- grid.y = 2 -- for the pipelined multi-rank
  //  schedule, which runs them on the exchange stream as soon as the state strips have arrived; see rk_stage_pipe)
  const int ja = chunk < 0 ? (blockIdx.y ? p.ny + chunk : 0) : (int)blockIdx.y * chunk;
  const int jb = chunk < 0 ? ja - chunk : min(ja + chunk, p.ny);
  const int e = ie % p.nens;
  const double *hp = p.hypk + (long long)(k * p.nens + e) * 8;
  const double hyr = hp[0], hyt = hp[1], p0 = hp[2], ihyt = hp[3];
  const double *col = S + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;      // row j at col + (j+HY)*sJ
  // where the outputs go: the parameter block's own (fused or one-member) layout, or member e's part of a member-major handle
  const int io = MM == 1 ? ie / p.nens : ie;
  const long long o_fyJ = MM == 1 ? mm.fyJ : p.fyJ, o_fyK = MM == 1 ? mm.fyK : p.fyK, o_row = MM == 1 ? mm.nx : NXI, o_nC = MM == 1 ? mm.nC : p.nC;
  const long long o_sJ = MM == 1 ? mm.sJ : p.sJ, o_sK = MM == 1 ? mm.sK : p.sK, o_sV = MM == 1 ? mm.sV : p.sV;
  const long long o_x0 = MM == 1 ? p.HX + io : (long long)p.HX * p.nens + ie;
  double *fy = MY + (MM == 1 ? e * mm.mfy : 0) + (long long)k * o_fyK + io;                        // face j at fy + j*o_fyJ
  unsigned char *upy = UPY + (MM == 1 ? e * mm.mfy : 0) + (long long)k * o_fyK + io;
  double *ty = tendY + (MM == 1 ? e * mm.tend : 0) + ((long long)k * p.ny) * o_row + io;           // row j at ty + j*o_row (+ l*o_nC)
  double *Swm = MM == 1 ? Sw + e * mm.slab : Sw;
  constexpr int HS = (ORD - 1) / 2;                           // stencil half width; the window holds rows j-HS .. j+HS
  double w[5][ORD], nxt[5], cn[5], fprev[5];
  // CONV: row r (halo rows wrap) comes from the coupler; the rows ja..jb-1 are this chunk's to store.  The row is REQUESTED at the
  // top of an iteration and converted at its end, when the values have arrived.
  const int hi = k * p.nens + e;
#define MW_ROW_CI(r) cpl(p, (long long)(k * p.ny + wrap_row(p, (r))) * NXI + ie)
#define MW_ROW_FINISH(raw, r, out5)                                                                                   \
  { double inv_den_;                                                                                                  \
    convert_cell_fast<K>(p, raw, hyr, hyt, p0, out5, inv_den_);   /* (the row's background values are in registers already) */ \
    if ((r) >= ja && (r) < jb) {                                                                                      \
      double *s_ = Swm + (long long)(k + p.HZ) * o_sK + (long long)((r) + p.HY) * o_sJ + o_x0;                        \
      s_[0] = out5[0]; s_[o_sV] = out5[1]; s_[2 * o_sV] = out5[2]; s_[3 * o_sV] = out5[3]; s_[4 * o_sV] = out5[4];    \
      convert_cell_tracers<K>(p, raw, inv_den_, s_, o_sV);                                                               \
    } }
#pragma unroll
  for (int v = 0; v < 5; v++) { cn[v] = 0; fprev[v] = 0; }
  if (CONV) {
#pragma unroll
    for (int s = 0; s < ORD; s++) {
      const CouplerCell raw = load_coupler_cell<K>(p, c, MW_ROW_CI(ja - 1 - HS + s));
      double r5[5];
      MW_ROW_FINISH(raw, ja - 1 - HS + s, r5)
#pragma unroll
      for (int v = 0; v < 5; v++) w[v][s] = r5[v];
    }
  } else {
#pragma unroll
    for (int v = 0; v < 5; v++) {
#pragma unroll
      for (int s = 0; s < ORD; s++) w[v][s] = col[(long long)v * p.sV + (long long)(wrap_row(p, ja - 1 - HS + s) + p.HY) * p.sJ];
    }
  }
#pragma unroll
  for (int v = 0; v < 5; v++) landed(w[v]);
  for (int j = ja - 1; j <= jb; j++) {
    const int jn = min(j + HS + 1, p.ny + p.HY - 1);            // clamp: the last prefetch is never used
    CouplerCell raw;
    if (CONV) raw = load_coupler_cell<K>(p, c, MW_ROW_CI(jn));
    else {
#pragma unroll
      for (int v = 0; v < 5; v++) nxt[v] = col[(long long)v * p.sV + (long long)(wrap_row(p, jn) + p.HY) * p.sJ];
    }
    double se[5], ne[5];
#pragma unroll
    for (int v = 0; v < 5; v++) weno_window_edges<ORD>(w[v], se[v], ne[v]);
    {                                                          // (computed on the ghost iteration j = ja-1 too: only the stores are predicated)
      const bool face = (j >= ja);
      const int jc = max(j, 0);
      // face j: L = north edge of cell j-1 (cn), R = south edge of cell j (se)
      const int bcmode = bc_mode_y<K>(p, jc);
      const bool zero = (bcmode == 1 || bcmode == 2) && (p.bc_y == MW_BC_WALL);
      if (__builtin_expect(bcmode == 3, 0)) {
#pragma unroll
        for (int v = 0; v < 5; v++) { const double *qv = col + (long long)v * p.sV + (long long)p.HY * p.sJ; double l_, r_, q_[ORD];
#pragma unroll
          for (int s = 0; s < ORD; s++) q_[s] = qv[(long long)(s - HS) * p.sJ];
          weno_window_edges<ORD>(q_, l_, r_); se[v] = l_; }
      }
      double Lr = cn[idR], Lu = cn[idV], Lt = cn[idT], Rr = se[idR], Ru = se[idV], Rt = se[idT];
      const bool ybc = (bcmode == 1 || bcmode == 2);           // wave-uniform (j is): a branch, interior faces carry no selects
      if (__builtin_expect(ybc, 0)) {
        if (bcmode == 1) { Lr = Rr; Lu = Ru; Lt = Rt; } else { Rr = Lr; Ru = Lu; Rt = Lt; }
        if (zero) { Lu = 0.0; Ru = 0.0; }                      // = the reference's zeroed normal momentum on both sides
      }
      double f[5], fn, fT;
      FaceState fs = riemann_primary<K>(p, Lr + hyr, Rr + hyr, Lu, Ru, Lt, Rt, hyt, p0, ihyt, false, fn, fT);
      int up = fs.ind;
      if (__builtin_expect(ybc, 0)) up = (bcmode == 1) ? 1 : 0;   // both sides hold the R (cell j) / L values
      f[idR] = fs.m_upw; f[idV] = fn; f[idT] = fT;
      {   // scalar copies first: a select between elements of two arrays is lowered to a pointer select -> scratch memory
        const double sU = se[idU], cU = cn[idU], sW = se[idW], cW = cn[idW];
        f[idU] = fs.m_upw * (up ? sU : cU);
        f[idW] = fs.m_upw * (up ? sW : cW);
      }
      if (CONV) {
        landed(raw.rho_d); landed(raw.u); landed(raw.v); landed(raw.w); landed(raw.temp);
#pragma unroll
        for (int tr = 0; tr < 4; tr++) if (!Cf<K>::spec || tr < Cf<K>::ntr(p)) landed(raw.tr[tr]);
      }
      else landed(nxt);                                        // the iteration's loads, in front of its stores (see landed())
      if (face) { fy[(long long)j * o_fyJ] = fs.m_upw; upy[(long long)j * o_fyJ] = (unsigned char)up; }
      if (j > ja) {
#pragma unroll
        for (int l = 0; l < 5; l++) ty[(long long)l * o_nC + (long long)(j - 1) * o_row] = -(f[l] - fprev[l]) * p.rdy;
      }
#pragma unroll
      for (int l = 0; l < 5; l++) fprev[l] = f[l];
    }
    if (CONV) MW_ROW_FINISH(raw, jn, nxt)
#pragma unroll
    for (int v = 0; v < 5; v++) {
      cn[v] = ne[v];
#pragma unroll
      for (int s = 0; s + 1 < ORD; s++) w[v][s] = w[v][s + 1];
      w[v][ORD - 1] = nxt[v];
    }
  }
#undef MW_ROW_CI
#undef MW_ROW_FINISH
}

// (Measured dead end, round 2: the 5-row window of every variable in LDS instead of registers -- a "ring", slot = row mod 5, no
//  per-iteration register shift (48 v_mov_b64 of 543 VALU instructions) and 50 VGPRs less.  With two waves per SIMD it was 10 %
//  SLOWER (1.39 against 1.25 ms per step: the 25 LDS reads of an iteration sit in front of the arithmetic that needs them), and
//  capped at 168 VGPRs for a third wave the compiler still spilled 16-52 registers and it was 5 % slower.)

// Y pass, tracers: flux(face j) = m_upw * (up ? south edge of cell j : north edge of cell j-1)
// (Measured, round 2: a wave-uniform "all five stencil rows are exactly 0 -> edges 0" short cut for cloud and rain -- bitwise
//  neutral -- removes 2/3 of this kernel's arithmetic on the supercell state and changes its time by 3 %: the kernel moves 57 B per
//  cell at 4.6 TB/s and is bandwidth-bound once the arithmetic shrinks.  k_tracers_fused gained 10 %, the step 4 %; a state with
//  cloud and rain everywhere lost 3 % to the test.  Not kept.)
// (No Cf<K> form: measured in round 3, the folded variant needs 112 instead of 154 VGPRs for three tracers, runs 4 instead of 3 waves
//  per SIMD and is 4-6 % SLOWER -- the kernel is bound by HBM, and more resident waves only interleave more rows' streams.)
template <int T, int ORD>
__global__ __launch_bounds__(256) void k_y_tracers(DyP p, const double *__restrict__ S, double *__restrict__ FY,
                                                   const double *__restrict__ MY, const unsigned char *__restrict__ UPY, int chunk,
                                                   int t0) {
  const int NXI = p.nx * p.nens;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.nz * NXI) return;
  const int k = (int)(t / NXI);
  const int ie = (int)(t - (long long)k * NXI);
  const int ja = chunk < 0 ? (blockIdx.y ? p.ny + chunk : 0) : (int)blockIdx.y * chunk;      // (chunk < 0: the two edge strips, see k_y_state)
  const int jb = chunk < 0 ? ja - chunk : min(ja + chunk, p.ny);
  const double *col = S + (long long)(5 + t0) * p.sV + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;
  double *fy = FY + (long long)k * p.fyK + ie;
  const unsigned char *upy = UPY + (long long)k * p.fyK + ie;
  // Loads run two rows ahead of their use (row j+4 and the face data of row j+1 are requested in iteration j): with ~75 VALU
  // instructions per tracer and row, one iteration is shorter than the memory latency.  (The copy nxt = nxt2 at the end of the
  // iteration still waits for the row just requested.  Measured, round 2: two register sets taking turns in a loop unrolled by
  // two, so that a set is only waited for two iterations after its request, need 236 instead of 154 VGPRs for three tracers --
  // two waves per SIMD instead of three -- and the kernel was 10 % slower.)
  constexpr int HS = (ORD - 1) / 2;
  double w[T][ORD], nxt[T], nxt2[T], cn[T];
#pragma unroll
  for (int v = 0; v < T; v++) {
    cn[v] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) w[v][s] = col[(long long)v * p.sV + (long long)(wrap_row(p, ja - 1 - HS + s) + p.HY) * p.sJ];
    nxt[v] = col[(long long)v * p.sV + (long long)(wrap_row(p, min(ja - 1 + HS + 1, p.ny + p.HY - 1)) + p.HY) * p.sJ];
  }
  double m_n = MY[(long long)k * p.fyK + ie + (long long)ja * p.fyJ];
  int up_n = upy[(long long)ja * p.fyJ];
#pragma unroll
  for (int v = 0; v < T; v++) landed(w[v]);
  landed(nxt); landed(m_n); asm volatile("" : "+v"(up_n));
  for (int j = ja - 1; j <= jb; j++) {
    const int jn = min(j + HS + 2, p.ny + p.HY - 1);
#pragma unroll
    for (int v = 0; v < T; v++) nxt2[v] = col[(long long)v * p.sV + (long long)(wrap_row(p, jn) + p.HY) * p.sJ];
    const int jl = min(max(j + 1, ja), jb);                     // unconditional loads (clamped row): nothing waits inside a branch
    const double m = m_n;
    const int up = up_n;
    m_n = MY[(long long)k * p.fyK + ie + (long long)jl * p.fyJ];
    up_n = upy[(long long)jl * p.fyJ];
    double se[T], ne[T];
#pragma unroll
    for (int v = 0; v < T; v++) weno_window_edges<ORD>(w[v], se[v], ne[v]);
    landed(nxt2); landed(m_n); asm volatile("" : "+v"(up_n));  // the iteration's loads, in front of its stores (see landed())
    if (j >= ja) {
      if (__builtin_expect(bc_mode_y<0>(p, j) == 3, 0)) {
#pragma unroll
        for (int v = 0; v < T; v++) { const double *qv = col + (long long)v * p.sV + (long long)p.HY * p.sJ; double l_, r_, q_[ORD];
#pragma unroll
          for (int s = 0; s < ORD; s++) q_[s] = qv[(long long)(s - HS) * p.sJ];
          weno_window_edges<ORD>(q_, l_, r_); se[v] = l_; }
      }
#pragma unroll
      for (int v = 0; v < T; v++) {   // scalar copies first (a select between two arrays' elements would go through scratch)
        const double sv = se[v], cv = cn[v];
        fy[(long long)(5 + t0 + v) * p.fyV + (long long)j * p.fyJ] = m * (up ? sv : cv);
      }
    }
#pragma unroll
    for (int v = 0; v < T; v++) {
      cn[v] = ne[v];
#pragma unroll
      for (int s = 0; s + 1 < ORD; s++) w[v][s] = w[v][s + 1];
      w[v][ORD - 1] = nxt[v]; nxt[v] = nxt2[v];
    }
  }
}

// Y pass, state variables AND tracers in one launch (one-stream schedule, folded configurations): the face's upwind mass flux and
// selector go from the Riemann solve straight into the tracer fluxes -- they are neither written nor read back (17 of 146 bytes
// per cell of the two launches), and the launch boundary between them is gone.  Same arithmetic as k_y_state / k_y_tracers.
// CONV (first stage of a step): rows come from the coupler's arrays and are converted on the way, as in k_y_state<true>; the
// tracers' slab values (rho_t / rho) then enter their windows from registers instead of being read back (24 more bytes per cell).
// The face-flux carry of the five state variables lives in LDS there (one private slot per thread): the eight windows plus the
// row in flight leave no registers for it.
__device__ __forceinline__ double tracer_slab_value(double rho_t, double inv_den) {
#pragma clang fp contract(off)
  return rho_t * inv_den;                                      // (= convert_cell_tracers)
}
// MT (member-major handles with 2 or 4 members, CONV): one launch for all members, the members of the same cells in one workgroup
// (see MemberOff) -- the converting launch of such a handle.
template <bool CONV, int K, int ORD, int T, bool MT = false>
__global__ __launch_bounds__(256, 2) void k_y_all(DyP p, const double *__restrict__ S, double *__restrict__ FY, double *__restrict__ tendY, int chunk,
                                               CouplerPtrs c, double *__restrict__ Sw, MemberOff mo, int row0, int rstride, int row_end,
                                               int pre_lo, int pre_hi, int fy_skip) {
  static_assert(!MT || CONV, "the member-co-located form exists for the converting launch only");
  // (fy_skip: the pipelined multi-rank schedule runs the inner rows and the two edge strips as TWO launches, possibly of two different
  //  instantiations, side by side on two streams; both compute the faces row0 and row_end they share.  The edge launch owns their tracer
  //  fluxes: bit 0 / bit 1 = this launch does not store face row0 / row_end -- one writer per face, whatever the compiler contracted where.)
  constexpr int NV = 5 + T;
  // (pre_lo < pre_hi: the pipelined multi-rank schedule has converted the strips it packs for the neighbours up front -- the HX cells
  //  next to the block's west / east edge in every row, and the rows outside [pre_lo, pre_hi) -- with k_coupler_to_state_fast on this
  //  stream, and the pack kernels read them on the exchange stream WHILE this launch runs: those slab cells are not stored again here.
  //  Storing the same values twice was only benign as long as two instantiations of the conversion produce the same bits.)
  const int NXI = p.nx * p.nens;
  __shared__ double lds_fprev[CONV ? 5 : 1][CONV ? 256 : 1];
  // parked column increments (DyP::pinc, mw_nudge_to_column_deferred): the thread's five numbers -- its level's increments of density_dry, uvel,
  // vvel, temp and water vapour -- in a private LDS slot, read back where a row is converted (no registers to spare for them across the loop)
  __shared__ double lds_inc[(CONV && !MT) ? 5 : 1][(CONV && !MT) ? 256 : 1];
  const bool pinc_on = CONV && !MT && p.pinc != nullptr;          // (wave-uniform)
  int mt_sub = 0;
  if (MT) {                                                      // (p = one member's view: nens = 1, cst = the member count)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), em = wv & (mo.n - 1);
    mt_sub = wv >> mo.sh;
    FY += em * mo.fy; tendY += em * mo.tend; Sw += em * mo.slab; S += em * mo.slab; p.hypk += em * mo.per; p.ce = em;
  }
  unsigned colx; int ja, jb;
  { colx = blockIdx.x; ja = row0 + (int)blockIdx.y * rstride; jb = min(ja + chunk, row_end); }   // (a launch covers the rows [row0, row_end) in chunks `rstride` rows apart: all of them, the inner ones, or the two edge strips)
  long long t = (long long)colx * 256 + threadIdx.x;             // flattened (k, ie): no idle tail per row
  if (MT) t = ((long long)colx * (4 >> mo.sh) + mt_sub) * 64 + (threadIdx.x & 63);
  if (t >= (long long)p.nz * NXI) return;
  const int k = (int)(t / NXI);
  const int ie = (int)(t - (long long)k * NXI);
  const int e = ie % p.nens;
  const double *hp = p.hypk + (long long)(k * p.nens + e) * 8;
  const double hyr = hp[0], hyt = hp[1], p0 = hp[2], ihyt = hp[3];
  if (pinc_on) {
#pragma unroll
    for (int l = 0; l < 5; l++) lds_inc[(CONV && !MT) ? l : 0][threadIdx.x] = p.pinc[((long long)l * p.nz + k) * p.nens + e];   // (the index: the array has one row in the forms that never get here)
  }
  const double *col = S + (long long)(k + p.HZ) * p.sK + (long long)p.HX * p.nens + ie;      // row j at col + (j+HY)*sJ
  double *fy = FY + (long long)k * p.fyK + ie;                                                // tracer v, face j at fy + (5+v)*fyV + j*fyJ
  double *ty = tendY + ((long long)k * p.ny) * NXI + ie;                                      // row j at ty + j*NXI (+ l*nC)
  constexpr int HS = (ORD - 1) / 2;
  double w[NV][ORD], nxt[NV], cn[NV], fprev_r[CONV ? 1 : 5];
  // (the pre-converted cells, see above: the lane's column inside a west / east strip, rows outside [plo_, phi_))
  // -> this lane's slab rows are [sja_, sjb_): the chunk's rows minus the pre-converted ones (an empty range for a strip column)
  const bool pre_on_ = pre_lo < pre_hi;
  const int sja_ = pre_on_ ? max(ja, pre_lo) : ja;
  const int sjb_ = (pre_on_ && (ie < p.HX * p.nens || ie >= NXI - p.HX * p.nens)) ? sja_ : (pre_on_ ? min(jb, pre_hi) : jb);
#define MW_ROW_CI(r) cpl(p, (long long)(k * p.ny + wrap_row(p, (r))) * NXI + ie)
  constexpr unsigned VANM = (K == 1) ? (((1u << T) - 1u) & ~1u) : ((1u << T) - 1u);        // (= tracer_may_vanish<K>)
  /* the coupler's value + its parked column increment: the addition ColumnNudger's second pass would have stored (rounded once, no contraction) */
#define MW_ROW_NUDGE(raw)                                                                                             \
  if (pinc_on) {                                                                                                      \
    constexpr int LI_ = (CONV && !MT) ? 1 : 0;     /* (the array has one row in the forms that never get here) */           \
    raw.rho_d = __dadd_rn(raw.rho_d, lds_inc[0][threadIdx.x]); raw.u = __dadd_rn(raw.u, lds_inc[1 * LI_][threadIdx.x]);      \
    raw.v = __dadd_rn(raw.v, lds_inc[2 * LI_][threadIdx.x]); raw.temp = __dadd_rn(raw.temp, lds_inc[3 * LI_][threadIdx.x]);  \
    _Pragma("unroll") for (int tr_ = 0; tr_ < T; tr_++) if (Cf<K>::is_wv(p, tr_)) raw.tr[tr_] = __dadd_rn(raw.tr[tr_], lds_inc[4 * LI_][threadIdx.x]); \
  }
  /* (skipv: wave-uniform -- the tracers that can vanish are zero in this row AND the slab row holds zeros already, see ym_ss) */
#define MW_ROW_FINISH(raw, r, out, skipv)                                                                             \
  { double inv_den_;                                                                                                  \
    convert_cell_fast<K>(p, raw, hyr, hyt, p0, out, inv_den_);                                                        \
    _Pragma("unroll") for (int v_ = 0; v_ < T; v_++) out[5 + v_] = tracer_slab_value(raw.tr[v_], inv_den_);          \
    if ((r) >= sja_ && (r) < sjb_) {                                                                                  \
      double *s_ = Sw + (long long)(k + p.HZ) * p.sK + (long long)((r) + p.HY) * p.sJ + (long long)p.HX * p.nens + ie; \
      _Pragma("unroll") for (int v_ = 0; v_ < 5; v_++) s_[(long long)v_ * p.sV] = out[v_];                            \
      _Pragma("unroll") for (int v_ = 5; v_ < NV; v_++) if (!((VANM >> (v_ - 5)) & 1u)) s_[(long long)v_ * p.sV] = out[v_]; \
      if (!(skipv)) { _Pragma("unroll") for (int v_ = 5; v_ < NV; v_++) if ((VANM >> (v_ - 5)) & 1u) s_[(long long)v_ * p.sV] = out[v_]; } \
    } }
#pragma unroll
  for (int v = 0; v < NV; v++) cn[v] = 0;
  if (CONV) {
#pragma unroll
    for (int l = 0; l < 5; l++) lds_fprev[l][threadIdx.x] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) {
      CouplerCell raw = load_coupler_cell<K>(p, c, MW_ROW_CI(ja - 1 - HS + s));
      double r8[NV];
      MW_ROW_NUDGE(raw)
      MW_ROW_FINISH(raw, ja - 1 - HS + s, r8, false)
#pragma unroll
      for (int v = 0; v < NV; v++) w[v][s] = r8[v];
    }
  } else {
#pragma unroll
    for (int l = 0; l < 5; l++) fprev_r[l] = 0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
      for (int s = 0; s < ORD; s++) w[v][s] = col[(long long)v * p.sV + (long long)(wrap_row(p, ja - 1 - HS + s) + p.HY) * p.sJ];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) landed(w[v]);
#if MW_ZERO_SKIP
  unsigned zm[T];                                                // (the zero short-cut of k_tracers_fused, for the tracers' y windows)
#pragma unroll
  for (int v = 0; v < T; v++) {
    zm[v] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) zm[v] |= (__any(w[5 + v][s] != 0.0) ? 1u : 0u) << s;
    if (!tracer_may_vanish<K>(p, v)) zm[v] = ~0u;
  }
  // zero-row maps (see k_zero_rows / k_zero_dilate), as scalar masks over the chunk's iterations (bit i: j = ja-1+i):
  //   ym_st: a tracer kernel's row beside face j may load the face's fluxes of the tracers that can vanish -- else they are not stored;
  //   ym_ld: those tracers may be non-zero in what the iteration touches -- else the entering row is not loaded (it is zero).
  // A wave's lanes belong to one level, or to two when it straddles a row end: both levels' words then.  (Iterations beyond the 64th,
  // waves over more than two levels: all bits set.)
  //   ym_ss (converting launch): the slab row that the iteration writes may hold something non-zero (DyP::zqk = the map of the rows the LAST
  //   conversion into this slab left zero; nullptr: unknown) -- else the zeros of a lean iteration are not stored over zeros.
  unsigned long long ym_st = ~0ull, ym_ld = ~0ull, ym_ss = ~0ull;
  if (!MT && p.zq != nullptr && p.zero_skip) {
    const int kA = __builtin_amdgcn_readfirstlane(k), ieA = __builtin_amdgcn_readfirstlane(ie);
    if (ieA + 63 < 2 * NXI) {
      const unsigned *fnA = p.zq + 3 * (long long)p.nz * p.zq_ld + (long long)kA * p.zq_ld + MW_ZR_HALO;
      const long long nextk = (ieA + 63 >= NXI && kA + 1 < p.nz) ? p.zq_ld : 0;     // the wave's last lanes lie in the next level
      // the x segments this wave's lanes lie in (the y kernel reads no x neighbour): of level kA, and of the next level when it straddles the row end
      const unsigned segA = zr_seg_mask(ieA, min(ieA + 63, NXI - 1), NXI, false);
      const unsigned segB = (ieA + 63 >= NXI) ? zr_seg_mask(0, ieA + 63 - NXI, NXI, false) : 0u;
      const int ji = ja - 1 + (int)(threadIdx.x & 63);
      bool st = (ji > jb);
      if (ji >= 1 && ji - 1 < p.ny) st = st || (((fnA[ji - 1] | fnA[ji - 1 + nextk]) & VANM) != 0u);
      if (ji >= 0 && ji < p.ny)     st = st || (((fnA[ji] | fnA[ji + nextk]) & VANM) != 0u);
      const unsigned long long idle = ~__ballot(true);           // (the grid's last wave may be partial: iterations whose lane is not there keep their bit)
      ym_st = __ballot(st) | idle;
      const unsigned *qyA = fnA + 3 * (long long)p.nz * p.zq_ld;
      const int jw = wrap_row(p, ji);
      bool ld = true;
      if (ji <= jb && jw >= 0 && jw < p.ny) ld = ((qyA[jw] & segA) | (qyA[jw + nextk] & segB)) != 0u;
      ym_ld = __ballot(ld) | idle;
      if (CONV && p.zqk != nullptr) {
        const unsigned *kzA = p.zqk + (long long)kA * p.zq_ld + MW_ZR_HALO;
        const int jr = wrap_row(p, min(ji + HS + 1, p.ny + p.HY - 1));       // the row iteration ji converts and stores
        bool ss = true;
        if (ji <= jb && jr >= 0 && jr < p.ny) ss = ((kzA[jr] & segA) | (kzA[jr + nextk] & segB)) != 0u;
        ym_ss = __ballot(ss) | idle;
      }
    }
  }
#endif
  for (int j = ja - 1; j <= jb; j++) {
    const int jn = min(j + HS + 1, p.ny + p.HY - 1);            // clamp: the last prefetch is never used
    CouplerCell raw;
#if MW_ZERO_SKIP
    const bool lean = (j - (ja - 1)) < 64 && !((ym_ld >> (j - (ja - 1))) & 1ull);
#else
    constexpr bool lean = false;
#endif
    if (CONV) raw = load_coupler_cell<K>(p, c, MW_ROW_CI(jn), lean ? ~VANM : ~0u);
    else {
#pragma unroll
      for (int v = 0; v < NV; v++) nxt[v] = 0.0;
#pragma unroll
      for (int v = 0; v < NV; v++) if (v < 5 || !((VANM >> (v >= 5 ? v - 5 : 0)) & 1u)) nxt[v] = col[(long long)v * p.sV + (long long)(wrap_row(p, jn) + p.HY) * p.sJ];
      if (!lean) {
#pragma unroll
        for (int v = 5; v < NV; v++) if ((VANM >> (v - 5)) & 1u) nxt[v] = col[(long long)v * p.sV + (long long)(wrap_row(p, jn) + p.HY) * p.sJ];
      }
    }
    double se[NV], ne[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#if MW_ZERO_SKIP
      if (v >= 5 && zm[v >= 5 ? v - 5 : 0] == 0u) { se[v] = 0.0; ne[v] = 0.0; continue; }
#endif
      weno_window_edges<ORD>(w[v], se[v], ne[v]);
    }
    {
      const bool face = (j >= ja);
      const int jc = max(j, 0);
      const int bcmode = bc_mode_y<K>(p, jc);
      const bool zero = (bcmode == 1 || bcmode == 2) && (p.bc_y == MW_BC_WALL);
      if (__builtin_expect(bcmode == 3, 0)) {
#pragma unroll
        for (int v = 0; v < NV; v++) { const double *qv = col + (long long)v * p.sV + (long long)p.HY * p.sJ; double l_, r_, q_[ORD];
#pragma unroll
          for (int s = 0; s < ORD; s++) q_[s] = qv[(long long)(s - HS) * p.sJ];
          weno_window_edges<ORD>(q_, l_, r_); se[v] = l_; }
      }
      double Lr = cn[idR], Lu = cn[idV], Lt = cn[idT], Rr = se[idR], Ru = se[idV], Rt = se[idT];
      const bool ybc = (bcmode == 1 || bcmode == 2);
      if (__builtin_expect(ybc, 0)) {
        if (bcmode == 1) { Lr = Rr; Lu = Ru; Lt = Rt; } else { Rr = Lr; Ru = Lu; Rt = Lt; }
        if (zero) { Lu = 0.0; Ru = 0.0; }
      }
      double f[5], fn, fT;
      FaceState fs = riemann_primary<K>(p, Lr + hyr, Rr + hyr, Lu, Ru, Lt, Rt, hyt, p0, ihyt, false, fn, fT);
      int up = fs.ind;
      if (__builtin_expect(ybc, 0)) up = (bcmode == 1) ? 1 : 0;
      f[idR] = fs.m_upw; f[idV] = fn; f[idT] = fT;
      {
        const double sU = se[idU], cU = cn[idU], sW = se[idW], cW = cn[idW];
        f[idU] = fs.m_upw * (up ? sU : cU);
        f[idW] = fs.m_upw * (up ? sW : cW);
      }
      if (CONV) {
        landed(raw.rho_d); landed(raw.u); landed(raw.v); landed(raw.w); landed(raw.temp);
#pragma unroll
        for (int tr = 0; tr < T; tr++) landed(raw.tr[tr]);
      }
      else landed(nxt);                                        // the iteration's loads, in front of its stores (see landed())
      if (face && !((fy_skip & 1) && j == row0) && !((fy_skip & 2) && j == row_end)) {
#pragma unroll
        for (int v = 0; v < T; v++) {                          // scalar copies first (a select between two arrays' elements would go through scratch)
          const double sv = se[5 + v], cv = cn[5 + v];
#if MW_ZERO_SKIP
          if (((VANM >> v) & 1u) && (j - (ja - 1)) < 64 && !((ym_st >> (j - (ja - 1))) & 1ull)) continue;   // nobody will load it
#endif
          fy[(long long)(5 + v) * p.fyV + (long long)j * p.fyJ] = fs.m_upw * (up ? sv : cv);
        }
      }
      double fp[5];
#pragma unroll
      for (int l = 0; l < 5; l++) fp[l] = CONV ? lds_fprev[l][threadIdx.x] : fprev_r[l];
      if (j > ja) {
#pragma unroll
        for (int l = 0; l < 5; l++) ty[(long long)l * p.nC + (long long)(j - 1) * NXI] = -(f[l] - fp[l]) * p.rdy;
      }
#pragma unroll
      for (int l = 0; l < 5; l++) { if (CONV) lds_fprev[l][threadIdx.x] = f[l]; else fprev_r[l] = f[l]; }
    }
#if MW_ZERO_SKIP
    const bool ss_skip = CONV && lean && (j - (ja - 1)) < 64 && !((ym_ss >> (j - (ja - 1))) & 1ull);
#else
    constexpr bool ss_skip = false;
#endif
    if (CONV) { MW_ROW_NUDGE(raw) MW_ROW_FINISH(raw, jn, nxt, ss_skip) }
#if MW_ZERO_SKIP
#pragma unroll
    for (int v = 0; v < T; v++) if (tracer_may_vanish<K>(p, v)) zm[v] = (zm[v] >> 1) | ((__any(nxt[5 + v] != 0.0) ? 1u : 0u) << (ORD - 1));
#endif
#pragma unroll
    for (int v = 0; v < NV; v++) {
      cn[v] = ne[v];
#pragma unroll
      for (int s = 0; s + 1 < ORD; s++) w[v][s] = w[v][s + 1];
      w[v][ORD - 1] = nxt[v];
    }
  }
#undef MW_ROW_CI
#undef MW_ROW_FINISH
#undef MW_ROW_NUDGE
}

// ---------------------------------------------------------------------------------------------------------------
// XZ pass.  wave = 64 fused-x lanes of one row j (n = nens lanes per x cell).  Marches k over [ka-1, kb] for the chunk
// [ka, kb).  Two ways to get a cell's four x-stencil neighbours:
//   nens == 1: whole-wave DPP shifts.  Lanes [0,2n) only feed stencils, [2n,64-2n) reconstruct, [3n,64-2n) own a lower x
//              face, [3n,64-3n) own a complete cell: 3 halo cells per side, 58 cells per wave.
//   nens  > 1: a DPP shift cannot move by n lanes and 3n halo lanes per side would idle 6n of 64 lanes (38 % for nens = 4),
//              so the neighbours are read from global memory (the same cache lines the wave has just touched: L1 hits) and
//              EVERY lane reconstructs; only the west neighbour's east-edge value and the east face's flux still travel
//              by lane (ds_bpermute): 1 halo cell per side, 64 - 2n cells per wave.
// (Measured dead ends for nens > 1, round 2 -- config 4's block runs at 87 % of the nens = 1 rate per cell:
//   * "member waves": a wave's lanes = 64 x cells of ONE member (nens doubles apart) so that the x stencil can use the DPP shifts:
//     bitwise the nens = 1 result, but every wave-wide access then touches nens times as many cache lines -- k_xz_state 13.7 ->
//     15.3 ms, k_tracers_fused 9.5 -> 13.4 ms per step on 256 x 512 x 128 x 4;
//   * the neighbour-load path for nens = 1 (62 instead of 58 cells per wave): 30 % slower than the DPP shifts.
//   What would close the gap is a member-major layout of the handle's internal arrays (every kernel then runs its nens = 1 path per
//   member and only the coupler-side accesses are strided) -- not done.)
// ---------------------------------------------------------------------------------------------------------------
struct XzGeom {
  int n, lane, NXI, j, q, qq, qa, e, i, qc, ka, kb, kstart;   // qq: index incl. halo (BC logic), qa: the index that is addressed (wrapped)
  int cell_lo, cell_hi, face_hi;                              // lanes [cell_lo, cell_hi) own a cell, [cell_lo, face_hi) a lower x face
  int om2, om1, op1, op2;                                     // nens > 1: offsets (doubles) of the x neighbours from the lane's own cell
  bool owns_face, owns_cell, valid;
};
// (nens == 1: hs + 1 halo lanes per side -- hs stencil cells and one more so that the west neighbour's east-edge value is rebuilt in
//  the wave: 58 cells per wave for WENO-5, 60 for WENO-3)
__host__ __device__ __forceinline__ int xz_cells_per_wave(int nens, int ord = 5) { return nens == 1 ? 64 - 2 * ((ord - 1) / 2 + 1) : 64 - 2 * nens; }
// (colx = the workgroup's column of wavefronts, [ka, kb) = the levels it marches)
template <bool N1, int ORD = 5>
__device__ __forceinline__ XzGeom xz_geom(const DyP &p, unsigned colx, int ka, int kb, int tiles_x, int rows4 = 0, int wslot = -1, int wpb = 4) {
  static_assert(N1 || ORD == 5, "the neighbour-load form (nens > 1 in the fused layout) exists for WENO-5 only");
  constexpr int HS = (ORD - 1) / 2;
  XzGeom g;
  g.n = N1 ? 1 : p.nens;
  g.lane = threadIdx.x & 63;
  g.NXI = p.nx * g.n;
  const int hw = N1 ? HS + 1 : 1;                             // halo cells per side
  const int U = 64 - 2 * hw * g.n;                            // cells (fused) a wave completes
  g.cell_lo = hw * g.n; g.cell_hi = 64 - hw * g.n; g.face_hi = N1 ? 64 - HS * g.n : 64;
  // (wslot / wpb: a block of the member-transposing form holds the nens members of 4 / nens tiles -- MemberOff below)
  const long long wid = (long long)colx * wpb + (wslot < 0 ? (int)(threadIdx.x >> 6) : wslot);   // wave id -> (row j, x tile)
  int tx;
  if (rows4) { const int jg = (int)(colx / tiles_x); tx = (int)(colx - (unsigned)jg * tiles_x); g.j = jg * 4 + (threadIdx.x >> 6); }
  else       { g.j = (int)(wid / tiles_x); tx = (int)(wid - (long long)g.j * tiles_x); }
  g.valid = g.j < p.ny;                                       // whole wave
  g.q = tx * U - hw * g.n + g.lane;                           // interior fused-x index of this lane (may be in the halo)
  const int q_hi = g.NXI + (N1 ? 3 : 1) * g.n - 1;            // last index a lane may address (nens > 1: +2n must stay in the halo)
  g.owns_face = (g.lane >= g.cell_lo) && (g.lane < g.face_hi) && (g.q < g.NXI + g.n);
  g.owns_cell = (g.lane >= g.cell_lo) && (g.lane < g.cell_hi) && (g.q < g.NXI);
  g.qq = min(g.q, q_hi);                                      // clamped for addressing
  g.e = N1 ? 0 : ((g.qq % g.n) + g.n) % g.n;
  g.i = (g.qq - g.e) / g.n;                                   // x cell index (can be -3..nx+2)
  g.qc = g.owns_cell ? g.q : 0;                               // safe index for per-cell arrays
  g.qa = wrap_xq(p, g.qq, g.NXI);
  g.om2 = wrap_xq(p, g.qq - 2 * g.n, g.NXI) - g.qa; g.om1 = wrap_xq(p, g.qq - g.n, g.NXI) - g.qa;       // qq in [-n, NXI+n): all four stay
  g.op1 = wrap_xq(p, g.qq + g.n, g.NXI) - g.qa;     g.op2 = wrap_xq(p, g.qq + 2 * g.n, g.NXI) - g.qa;   // inside the 3-cell halo
  g.ka = ka;
  g.kb = kb;
  g.kstart = (g.ka == 0) ? 0 : g.ka - 1;                      // no ghost cell below the wall
  return g;
}
// The four x-stencil neighbours of the lane's cell value c0 (level pointer lvl = the lane's own cell at that level).
template <bool N1>
__device__ __forceinline__ void x_neighbours(double c0, const double *__restrict__ lvl, int om2, int om1, int op1, int op2, int lane, int n,
                                             double &m2, double &m1, double &p1, double &p2) {
  if (N1) {
    m1 = from_west<true>(c0, lane, n); p1 = from_east<true>(c0, lane, n);
    m2 = from_west<true>(m1, lane, n); p2 = from_east<true>(p1, lane, n);
  } else { m2 = lvl[om2]; m1 = lvl[om1]; p1 = lvl[op1]; p2 = lvl[op2]; }
}

// Member-co-located form (MT) of the two kernels that write the coupler's arrays in the last stage of a time step (D13), for a
// member-major handle with 2 or 4 members.  The coupler keeps the ensemble index fastest, a marching wave holds ONE member's x row:
// its D13 values go out as 8-byte stores nens doubles apart.  Issued from one launch per member those quarter sectors never met in
// L2 (measured, round 3: +20 % step time).  Here ONE launch holds all members and a workgroup = the nens members of the same tile
// (wave w = member w % nens of tile slot w / nens): the members' stores to a line leave the same CU within a level or two of each
// other and are merged in the XCD's L2 before the line is written back.  That replaces the k_member_to_coupler pass (8 doubles
// read + 8 written per cell and time step): config 4's block 26.5 -> 25.5 ms with the same treatment of D1 (k_y_state<.., MM = 2>).
// Everything per member -- slabs, face arrays, background tables -- is one stride away.
// (Also measured: the level's values transposed through LDS, one s_barrier per level, 64 consecutive doubles stored per wave --
//  fully coalesced, and SLOWER than the pass it replaces: k_tracers_fused<3, 1> 3.7 instead of 3.3 ms, k_xz_state<3, 1> +0.5 ms;
//  the lock step costs more than the quarter-sector stores.)

// MODE 1 (last stage of the last cycle): u, v, w also go to the coupler's arrays (D13, :1929-1932: the slab holds (rho u)/rho
// already), so that the tracer stage, which finishes D13, neither re-reads nor re-writes them.
template <int STAGE, bool N1, int MODE, int HPL, int K, int ORD, bool MT = false>
__global__ __launch_bounds__(256, 2) void k_xz_state(DyP p, const double *__restrict__ S, const double *__restrict__ Sn,
                                                  double *__restrict__ Sout, double *__restrict__ MX, double *__restrict__ MZ,
                                                  unsigned char *__restrict__ UPX, unsigned char *__restrict__ UPZ,
                                                  const double *__restrict__ tendY, double dt_stage, double dt_dyn, int chunk,
                                                  int tiles_x, double *__restrict__ cu, double *__restrict__ cv, double *__restrict__ cw,
                                                  MemberOff mo) {
  static_assert(!MT || (N1 && HPL && MODE == 1), "the member-co-located form is the D13 variant of the nens == 1 kernel");
  constexpr int HS = (ORD - 1) / 2;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int mt_e = 0, mt_sub = 0;
  if (MT) {
    mt_e = wv & (mo.n - 1); mt_sub = wv >> mo.sh;
    S += mt_e * mo.slab; Sn += mt_e * mo.slab; Sout += mt_e * mo.slab; MX += mt_e * mo.mx; UPX += mt_e * mo.mx; MZ += mt_e * mo.mz; UPZ += mt_e * mo.mz;
    tendY += mt_e * mo.tend; p.hypk += mt_e * mo.per; p.ce = mt_e;
  }
  __shared__ double lds_c[8];
  __shared__ double lds_xpart[5][256], lds_fzprev[5][256];
  extern __shared__ double lds_hp_all[];
  Segment sg;
  block_segment(chunk, p.nz, sg);
  const XzGeom g = xz_geom<N1, ORD>(p, sg.col, sg.a, sg.b, tiles_x, 0, MT ? mt_sub : -1, MT ? 4 >> mo.sh : 4);
  const int seg_len = sg.b - sg.a;
  // HPL (nens == 1): the eight background values of every level of this chunk (DyP::hypk rows kstart..kb) are copied to LDS once
  // and read from there (a broadcast read, issued with the iteration's other loads).  Read as scalar loads they were placed right
  // in front of their first use -- the kernel has no spare SGPRs to hold them any earlier -- and their latency was exposed three
  // times per level (-1 % kernel time).  All 256 threads copy, also those of waves beyond the last row, which leave right after.
  double *lds_hp = lds_hp_all + (MT ? wv * (seg_len + 2) * 8 : 0);     // (MT: one table per wave -- the members' backgrounds differ)
  // ... and so do the few uniform doubles of the finalisation (grid spacings, the stage's time-step factors, gravity): as kernel
  // arguments they occupy 12 SGPRs for the whole loop in a kernel that spills SGPRs to VGPR lanes (every spilled one comes back as a
  // v_readlane, a VALU instruction); as broadcast LDS reads they are VGPR operands where they are used.
  if (HPL) {
    const int nrow = g.kb - g.kstart + 1;
    if (MT) { for (int i = g.lane; i < nrow * 8; i += 64) lds_hp[i] = p.hypk[(long long)g.kstart * 8 + i]; }
    else for (int i = threadIdx.x; i < nrow * 8; i += 256) lds_hp[i] = p.hypk[(long long)g.kstart * 8 + i];     // (nens == 1: row k at k*8)
    if (threadIdx.x < 8) {
      const double cdt = (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;
      lds_c[threadIdx.x] = threadIdx.x == 0 ? p.rdx : threadIdx.x == 1 ? p.rdz : threadIdx.x == 2 ? cdt : threadIdx.x == 3 ? -p.grav : 0.0;
    }
    __syncthreads();
  }
  if (!g.valid) return;
  const int n = g.n, lane = g.lane, NXI = g.NXI, j = g.j, q = g.q, e = g.e;
  const double *col = S + (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qa;           // level k at col + (k+HZ)*sK
  const long long cell0 = (long long)j * NXI + g.qc;                                           // + k*ny*NXI
  const long long slab0 = (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qc;           // + (k+HZ)*sK
  const int planeC = p.ny * NXI;                                                                  // (cells of a level: < 2^31, see Stride32 -- every product below is 32 x 32 -> 64)
  // The two per-cell carries that are written once and read once per level (the x+y part of the tendency and the lower z-face
  // flux) live in LDS, one private slot per thread: 20 VGPRs less in a kernel that sits at the 256-register limit (measured:
  // -5 % run time; moving more carries there, or doing the same in k_y_state / k_tracers_fused, was slower).
  double w[5][ORD], nxt[5], ct[5];
#pragma unroll
  for (int v = 0; v < 5; v++) {
    ct[v] = 0; lds_fzprev[v][threadIdx.x] = 0; lds_xpart[v][threadIdx.x] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) w[v][s] = load_zlevel<K>(p, col + (long long)v * p.sV, g.kstart - HS + s, v == idW);
  }
#pragma unroll
  for (int v = 0; v < 5; v++) landed(w[v]);
  for (int k = g.kstart; k <= g.kb; k++) {
    const bool top = (k == p.nz);                              // only the boundary face nz, no cell to reconstruct
    const bool xwork = (k >= g.ka) && (k < g.kb);              // cells of this chunk (ghost levels only do z)
    const bool zface = (k >= g.ka);                            // face k belongs to this chunk (k == kb: closing face)
    const bool fin = (k > g.ka);                               // cell k-1 gets finalised in this iteration
    // ---------------- issue this iteration's global loads
    {
      const int kn = min(k + HS + 1, p.nz + p.HZ - 1);
#pragma unroll
      for (int v = 0; v < 5; v++) nxt[v] = load_zlevel<K>(p, col + (long long)v * p.sV, kn, v == idW);
    }
    // (unconditional, at levels clamped into the chunk: a ghost iteration loads values nobody uses, and no default values have to be
    //  materialised for a skipped load -- ten v_mov_b64 per level)
    double snv[5], tyv[5], immv = 0;
    const int kfc = max(k - 1, g.ka), kxc = min(max(k, g.ka), g.kb - 1);       // the level that is finalised / reconstructed in x
#pragma unroll
    for (int l = 0; l < 5; l++) { snv[l] = 0; tyv[l] = 0; }
    if (STAGE != 1) {
#pragma unroll
      for (int l = 0; l < 5; l++) snv[l] = Sn[(long long)l * p.sV + slab0 + (long long)(kfc + p.HZ) * p.sK];
    }
    if (!Cf<K>::sim2d(p)) {
#pragma unroll
      for (int l = 0; l < 5; l++) tyv[l] = tendY[(long long)l * p.nC + cell0 + (long long)kxc * planeC];
    }
    if (Cf<K>::immersed(p)) immv = p.imm[cpl(p, cell0 + (long long)kfc * planeC)];
    double hpl[8];
    if (HPL) {
#pragma unroll
      for (int f = 0; f < 8; f++) hpl[f] = lds_hp[(k - g.kstart) * 8 + f];
    }
    // ------------------------------------------------ X direction (cell k = window centre)
    double fxs[5];
    int upx = 0;
    if (xwork) {
      const double *hp = p.hypk + (long long)(k * n + e) * 8;
      const double hyr = HPL ? hpl[0] : hp[0], hyt = HPL ? hpl[1] : hp[1], p0 = HPL ? hpl[2] : hp[2], ihyt = HPL ? hpl[3] : hp[3];
      double we[5], ee[5];
#pragma unroll
      for (int v = 0; v < 5; v++) {
        double c0 = w[v][HS], m2, m1, p1, p2;
        if (ORD == 3) { m1 = from_west<true>(c0, lane, n); p1 = from_east<true>(c0, lane, n); weno3_edges_fast(m1, c0, p1, we[v], ee[v]); }
        else {
          x_neighbours<N1>(c0, col + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK, g.om2, g.om1, g.op1, g.op2, lane, n, m2, m1, p1, p2);
          weno5_edges_fast(m2, m1, c0, p1, p2, we[v], ee[v]);
        }
      }
      const int bcmode = bc_mode_x<K>(p, g.i);
      const bool zero = (bcmode == 1 || bcmode == 2) && (p.bc_x == MW_BC_WALL);
      if (__builtin_expect(bcmode == 3, 0)) {                  // quirk 1: slot 1 at face nx keeps cell 0's west edge (:985)
        const double *c0p = col + (long long)(k + p.HZ) * p.sK - (long long)p.nx * n;
#pragma unroll
        for (int v = 0; v < 5; v++) { const double *qv = c0p + (long long)v * p.sV; double l_, r_, q_[ORD];
#pragma unroll
          for (int s = 0; s < ORD; s++) q_[s] = qv[(s - HS) * n];
          weno_window_edges<ORD>(q_, l_, r_); we[v] = l_; }
      }
      double Lv[5];
#pragma unroll
      for (int v = 0; v < 5; v++) Lv[v] = from_west<N1>(ee[v], lane, n);       // west neighbour's east-edge values
      double Lr = Lv[idR], Lu = Lv[idU], Lt = Lv[idT], Rr = we[idR], Ru = we[idU], Rt = we[idT];
      if (bcmode == 1) { Lr = Rr; Lu = Ru; Lt = Rt; }
      if (bcmode == 2) { Rr = Lr; Ru = Lu; Rt = Lt; }
      double fn, fT;
      FaceState fs = riemann_primary<K>(p, Lr + hyr, Rr + hyr, Lu, Ru, Lt, Rt, hyt, p0, ihyt, zero, fn, fT);
      int up = fs.ind;
      if (bcmode == 1) up = 1;
      if (bcmode == 2) up = 0;
      fxs[idR] = fs.m_upw; fxs[idU] = fn; fxs[idT] = fT;
      fxs[idV] = fs.m_upw * (up ? we[idV] : Lv[idV]);
      fxs[idW] = fs.m_upw * (up ? we[idW] : Lv[idW]);
      upx = up;                                                // (stored below, behind landed())
    }
    // ------------------------------------------------ Z direction: reconstruct cell k, solve face k
    // (also on the boundary face nz, where there is no cell: the window then holds clamped levels, the edge values are finite and
    //  nobody reads them -- a branch here would need ten default values materialised in every iteration)
    double be[5], te[5];
#pragma unroll
    for (int v = 0; v < 5; v++) weno_window_edges<ORD>(w[v], be[v], te[v]);
    // (also on the ghost iteration below the chunk, whose face belongs to the chunk underneath: its flux is never stored and never
    //  enters a tendency -- the first cell that is finalised is ka, with the faces ka and ka + 1)
    double fzs[5];
    int upz = 0;
    {
      const double *hp = p.hypk + (long long)(k * n + e) * 8;
      const double hyr = HPL ? hpl[4] : hp[4], hyt = HPL ? hpl[5] : hp[5], p0 = HPL ? hpl[6] : hp[6], ihyt = HPL ? hpl[7] : hp[7];
      double Lr = ct[idR], Lu = ct[idW], Lt = ct[idT], Rr = be[idR], Ru = be[idW], Rt = be[idT];
      // :1020-1038 wall/open edge-value rule at the two boundary faces.  k is wave-uniform: a branch, so that the interior faces
      // carry no selects (a zero normal velocity on both sides is what the reference's "zero the momentum" amounts to).
      const bool zbc = (k == 0) || top;
      if (__builtin_expect(zbc, 0)) {
        if (k == 0) { Lr = Rr; Lu = Ru; Lt = Rt; } else { Rr = Lr; Ru = Lu; Rt = Lt; }
        if (Cf<K>::z_wall(p)) { Lu = 0.0; Ru = 0.0; }
      }
      double fn, fT;
      FaceState fs = riemann_primary<K>(p, Lr + hyr, Rr + hyr, Lu, Ru, Lt, Rt, hyt, p0, ihyt, false, fn, fT);
      int up = fs.ind;
      if (__builtin_expect(zbc, 0)) up = (k == 0) ? 1 : 0;
      fzs[idR] = fs.m_upw; fzs[idW] = fn; fzs[idT] = fT;
      fzs[idU] = fs.m_upw * (up ? be[idU] : ct[idU]);
      fzs[idV] = fs.m_upw * (up ? be[idV] : ct[idV]);
      upz = up;
    }
    // ------------------------------------------------ all loads of this iteration have landed (see landed()); its stores follow
    landed(nxt); landed(snv); landed(tyv); landed(immv);
    if (xwork && g.owns_face && (g.owns_cell || q >= NXI)) {
      const long long fo = (long long)k * p.fxK + (long long)j * p.fxJ + q;
      MX[fo] = fxs[idR];  UPX[fo] = (unsigned char)upx;
    }
    if (zface && g.owns_cell) {
      const long long fo = (long long)k * p.fzK + (long long)j * p.fzJ + q;
      MZ[fo] = fzs[idR];  UPZ[fo] = (unsigned char)upz;
    }
    // ------------------------------------------------ finalise cell k-1 (it now has its upper z face)
    if (fin) {
      const int kc = k - 1;
      const double hyc = HPL ? lds_hp[(kc - g.kstart) * 8] : p.hypk[(long long)(kc * n + e) * 8];
      constexpr int wi = HS - 1;                               // window slot that holds level k-1 (window is centred on k)
      const double rho_s = w[idR][wi] + hyc;
      const double rho_n = (STAGE == 1) ? rho_s : snv[idR] + hyc;
      double imm_coef = 0;
      if (Cf<K>::immersed(p)) { double tau = 1.e3 * dt_stage; imm_coef = -fmin(1.0, dt_stage / tau); }
      const double ru_s = w[idU][wi] * rho_s, rv_s = w[idV][wi] * rho_s;
      double inv_rho_new = 1.0;
      double *so = Sout + slab0 + (long long)(kc + p.HZ) * p.sK;
      double xpart[5], fzprev[5];                              // the LDS reads in one batch: one exposed LDS latency, not five
#pragma unroll
      for (int l = 0; l < 5; l++) { xpart[l] = lds_xpart[l][threadIdx.x]; fzprev[l] = lds_fzprev[l][threadIdx.x]; }
#pragma unroll
      for (int l = 0; l < 5; l++) {
        double raw_s = w[l][wi];
        double q_s = (l == idR || l == idT) ? raw_s : raw_s * rho_s;
        double q_n;
        if (STAGE == 1) q_n = q_s;
        else q_n = (l == idR || l == idT) ? snv[l] : snv[l] * rho_n;
        double tend = xpart[l] - (fzs[l] - fzprev[l]) * (HPL ? lds_c[1] : p.rdz);
        if (l == idW && Cf<K>::gravity(p)) tend += (HPL ? lds_c[3] : -p.grav) * rho_s;
        if (l == idU && Cf<K>::coriolis(p)) tend += p.fcor * rv_s;
        if (l == idV && Cf<K>::coriolis(p)) tend -= p.fcor * ru_s;
        if (l == idV && Cf<K>::sim2d(p)) tend = 0;
        if (Cf<K>::immersed(p)) { double imm_tend = imm_coef * q_s / dt_stage; tend = immv * imm_tend + (1 - immv) * tend; }
        double qnew;
        const double cdt = HPL ? lds_c[2] : (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;   // (one product, as before)
        if (STAGE == 1)      qnew = q_n + cdt * tend;
        else if (STAGE == 2) qnew = (3.0 / 4.0) * q_n + (1.0 / 4.0) * q_s + cdt * tend;
        else                 qnew = (1.0 / 3.0) * q_n + (2.0 / 3.0) * q_s + cdt * tend;
        if (l == idR) inv_rho_new = fast_rcp(qnew + hyc);
        double stored = (l == idR || l == idT) ? qnew : qnew * inv_rho_new;
        // (MODE 1: what the result slab's (rho theta)' slot is read for afterwards is D13's pressure, :1935 -- by the tracer stage of
        //  this stage and its correction pass; the next time_step starts from the coupler's fields.  The series is evaluated HERE,
        //  where the level's background values are at hand, and the slot holds p: the tracer stage's D13 variant is the kernel at
        //  the register limit, this one is not.)
        if (MODE == 1 && l == idT) {
          const double *hq = HPL ? lds_hp + (kc - g.kstart) * 8 : p.hypk + (long long)(kc * n + e) * 8;
          stored = pressure_fast<K>(p, qnew, hq[1], hq[2], hq[3]);
        }
        // (MODE 1, the last stage of a time step: u, v, w go to the coupler's arrays only.  Nobody reads the result slab's velocities
        //  any more -- the tracer stage of this stage needs rho' and (rho theta)', and the next time_step starts from the coupler's
        //  fields (D1) -- so three of the five slab stores are dropped: 24 B per cell and step.)
        if (g.owns_cell && !(MODE == 1 && (l == idU || l == idV || l == idW))) so[(long long)l * p.sV] = stored;
        if (MODE == 1 && g.owns_cell && (l == idU || l == idV || l == idW))
          (l == idU ? cu : l == idV ? cv : cw)[cpl(p, cell0 + (long long)kc * planeC)] = stored;
      }
    }
    // ------------------------------------------------ carries for the next level
    if (xwork) {                                               // x (+ y) part of cell k: east face = lower face of lane + n
#pragma unroll
      for (int l = 0; l < 5; l++) {
        double fe = from_east<N1>(fxs[l], lane, n);
        lds_xpart[l][threadIdx.x] = -(fe - fxs[l]) * (HPL ? lds_c[0] : p.rdx) + tyv[l];
      }
    }
#pragma unroll
    for (int l = 0; l < 5; l++) lds_fzprev[l][threadIdx.x] = fzs[l];
    if (!top) {
#pragma unroll
      for (int v = 0; v < 5; v++) {
        ct[v] = te[v];
#pragma unroll
        for (int s = 0; s + 1 < ORD; s++) w[v][s] = w[v][s + 1];
        w[v][ORD - 1] = nxt[v];
      }
    }
  }
}

// XZ pass, tracers, with the FCT positivity step (D10, :498-516) folded in.
//   flux(face) = m_upw * upwind edge value.  The x and z fluxes of a cell are kept in registers for one more level; when the
//   cell's top face is known, all six of its face fluxes are at hand (the y pair is read from FY, written by k_y_tracers
//   before), so the cell's FCT multiplier  mult = min(1, mass_available / mass_out)  is computed here and every x/z face is
//   stored ONCE, already scaled by its donor cell's multiplier -- there is no separate FCT pass over the flux arrays.
//   Outgoing y faces are rescaled in place only where mult < 1 (rare); the race with the neighbouring rows that read them is
//   the reference's own benign one (:495-497: a face is only rescaled by the cell it leaves, and the sign never changes).
//   Faces on a wave's or chunk's edge have exactly one writer: the side that owns the donor cell (the other side computes
//   the identical flux value and drops it); domain-edge faces whose donor lies outside are stored unscaled, like the
//   reference's interior-only loop leaves them.
//   Block = 4 waves = 4 consecutive rows j of one x tile (rows4), so that FY rows are shared through the CU's L1.
template <int T, bool N1>
__global__ __launch_bounds__(256) void k_xz_tracers(DyP p, const double *__restrict__ S, double *__restrict__ FX,
                                                    double *FY, double *__restrict__ FZ, const double *__restrict__ MX,
                                                    const double *__restrict__ MZ, const unsigned char *__restrict__ UPX,
                                                    const unsigned char *__restrict__ UPZ, double dt, int chunk, int tiles_x, int t0,
                                                    int rows4) {
  const BlockXY blk_ = xcd_block();
  const XzGeom g = xz_geom<N1>(p, blk_.x, (int)blk_.y * chunk, min((int)blk_.y * chunk + chunk, p.nz), tiles_x, rows4);
  if (!g.valid) return;
  const int n = g.n, lane = g.lane, NXI = g.NXI, j = g.j, q = g.q;
  const double *col = S + (long long)(5 + t0) * p.sV + (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qa;
  const double *rcol = S + (long long)idR * p.sV + (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qa;
  const long long fxo = (long long)j * p.fxJ + (g.owns_face ? q : 0);
  const long long fzo = (long long)j * p.fzJ + g.qc;
  const long long fyo = (long long)j * p.fyJ + g.qc;
  const bool west_owns = (lane - n >= g.cell_lo) && (lane - n < g.cell_hi) && (q - n < NXI);
  const bool first_face = g.owns_face && (q < n);              // i == 0: the donor of an eastward flux is outside the rank
  const bool last_face = g.owns_face && (q >= NXI);            // i == nx
  const bool do_y = !p.sim2d;
  const double vol = p.dx * p.dy * p.dz;
  double w[T][5], nxt[T], ct[T];
  double fxp[T], fzp[T], fys[T], fyn[T], multp[T];
  bool pend = false;
#pragma unroll
  for (int v = 0; v < T; v++) {
    ct[v] = 0; fxp[v] = fzp[v] = fys[v] = fyn[v] = 0; multp[v] = 1;
#pragma unroll
    for (int s = 0; s < 5; s++) w[v][s] = load_zlevel(p, col + (long long)v * p.sV, g.kstart - 2 + s, false);
  }
  for (int k = g.kstart; k <= g.kb; k++) {
    const bool top = (k == p.nz);
    const bool xwork = (k >= g.ka) && (k < g.kb);
    const bool zface = (k >= g.ka);
    const int kn = min(k + 3, p.nz + p.HZ - 1);
#pragma unroll
    for (int v = 0; v < T; v++) nxt[v] = load_zlevel(p, col + (long long)v * p.sV, kn, false);
    double mx = 0, mz = 0, rhop = 0; int upx = 0, upz = 0;
    if (xwork) { mx = MX[(long long)k * p.fxK + fxo]; upx = UPX[(long long)k * p.fxK + fxo]; }
    if (pend) {                                                // what the cell below (k-1) still needs for its FCT multiplier
      rhop = rcol[(long long)(k - 1 + p.HZ) * p.sK] + p.hyc[(k - 1) * p.nens + g.e];
      if (do_y) {
#pragma unroll
        for (int v = 0; v < T; v++) {
          const double *fy = FY + (long long)(5 + t0 + v) * p.fyV + (long long)(k - 1) * p.fyK + fyo;
          fys[v] = fy[0]; fyn[v] = fy[p.fyJ];
        }
      }
    }
    if (zface) { mz = MZ[(long long)k * p.fzK + fzo]; upz = UPZ[(long long)k * p.fzK + fzo]; }
    double fxn[T], fzn[T];
#pragma unroll
    for (int v = 0; v < T; v++) fxn[v] = fzn[v] = 0;
    if (xwork) {
      const bool quirk = bc_mode_x(p, g.i) == 3;
#pragma unroll
      for (int v = 0; v < T; v++) {
        double c0 = w[v][2], m2, m1, p1, p2;
        x_neighbours<N1>(c0, col + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK, g.om2, g.om1, g.op1, g.op2, lane, n, m2, m1, p1, p2);
        double we, ee;
        weno5_edges_fast(m2, m1, c0, p1, p2, we, ee);
        if (__builtin_expect(quirk, 0)) {
          const double *qv = col + (long long)v * p.sV + (long long)(k + p.HZ) * p.sK - (long long)p.nx * n; double r_;
          weno5_edges_fast(qv[-2 * n], qv[-n], qv[0], qv[n], qv[2 * n], we, r_);
        }
        double Lq = from_west<N1>(ee, lane, n);
        fxn[v] = g.owns_face ? mx * (upx ? we : Lq) : 0.0;
      }
    }
    double te[T];
#pragma unroll
    for (int v = 0; v < T; v++) {
      double be = 0; te[v] = 0;
      if (!top) weno5_edges_fast(w[v][0], w[v][1], w[v][2], w[v][3], w[v][4], be, te[v]);
      if (zface && g.owns_cell) fzn[v] = mz * (upz ? be : ct[v]);
    }
    if (pend) {                                                // finish cell kp = k-1 (all six face fluxes known now)
      const int kp = k - 1;
#pragma unroll
      for (int v = 0; v < T; v++) {
        const double fe = from_east<N1>(fxp[v], lane, n);
        double mult = 1.0;
        {
#pragma clang fp contract(off)
          const double mass_available = fmax(w[v][1] * rhop, 0.0) * p.dx * p.dy * p.dz;
          const double out_x = (fmax(fe, 0.0) - fmin(fxp[v], 0.0)) * p.rdx;
          const double out_y = (fmax(fyn[v], 0.0) - fmin(fys[v], 0.0)) * p.rdy;
          const double out_z = (fmax(fzn[v], 0.0) - fmin(fzp[v], 0.0)) * p.rdz;
          const double mass_out = (out_x + out_y + out_z) * dt * p.dx * p.dy * p.dz;
          if (g.owns_cell && ((p.pos_mask >> (t0 + v)) & 1u) && mass_out > mass_available) mult = mass_available / mass_out;
        }
        const double mult_w = from_west<N1>(mult, lane, n);
        // x face (west face of this lane), level kp
        {
          const double F = fxp[v];
          double *dst = FX + (long long)(5 + t0 + v) * p.fxV + (long long)kp * p.fxK + fxo;
          if (F > 0) { if (west_owns) *dst = F * mult_w; else if (first_face) *dst = F; }
          else       { if (g.owns_cell) *dst = F * mult; else if (last_face) *dst = F; }
        }
        if (g.owns_cell) {
          // z face kp (bottom of the cell)
          const double G = fzp[v];
          double *dz_ = FZ + (long long)(5 + t0 + v) * p.fzV + (long long)kp * p.fzK + fzo;
          if (G > 0) { if (kp > g.ka) *dz_ = G * multp[v]; else if (kp == 0) *dz_ = G; }
          else       *dz_ = G * mult;
          if (k == g.kb) {                                     // top face of the chunk
            const double H = fzn[v];
            if (H > 0) dz_[p.fzK] = H * mult; else if (top) dz_[p.fzK] = H;
          }
          if (__builtin_expect(mult < 1.0, 0)) {               // outgoing y faces, in place
            double *fy = FY + (long long)(5 + t0 + v) * p.fyV + (long long)kp * p.fyK + fyo;
            if (fys[v] < 0) fy[0] = fys[v] * mult;
            if (fyn[v] > 0) fy[p.fyJ] = fyn[v] * mult;
          }
        }
        multp[v] = mult;
      }
    }
    pend = xwork;
#pragma unroll
    for (int v = 0; v < T; v++) { fxp[v] = fxn[v]; fzp[v] = fzn[v]; }
    if (!top) {
#pragma unroll
      for (int v = 0; v < T; v++) {
        ct[v] = te[v];
        w[v][0] = w[v][1]; w[v][1] = w[v][2]; w[v][2] = w[v][3]; w[v][3] = w[v][4]; w[v][4] = nxt[v];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tracer update: divergence of the (FCT-corrected) tracer fluxes + SSPRK3 combine + clip + storage divide.
//   MODE 0: write the tracer part of the new slab;   MODE 1 (last stage of the last cycle): also D13 (:1927-1950).
// The new density is read from the slab k_pass_xz has just written.
// ---------------------------------------------------------------------------------------------------------------
template <int STAGE, int MODE>
__global__ __launch_bounds__(256) void k_tracer_update(DyP p, const double *Sstar, const double *Sn, double *Sout,
                                                       const double *__restrict__ FX, const double *__restrict__ FY,
                                                       const double *__restrict__ FZ, double dt_dyn, CouplerPtrs c) {
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
  const double rho_s = Sstar[so + idR * p.sV] + hyc;
  const double rho_n = (STAGE == 1) ? rho_s : Sn[so + idR * p.sV] + hyc;
  const double rho_new = Sout[so + idR * p.sV] + hyc;
  const double inv_rho_new = fast_rcp(rho_new);
  double rho_dry = rho_new, rho_v = 0;
  for (int l = 5; l < p.V; l++) {
    double q_s = Sstar[so + l * p.sV] * rho_s;
    double q_n = (STAGE == 1) ? q_s : Sn[so + l * p.sV] * rho_n;
    double tend = -(fx[l * p.fxV + p.nens] - fx[l * p.fxV]) * p.rdx
                  -(fy[l * p.fyV + p.fyJ ] - fy[l * p.fyV]) * p.rdy
                  -(fz[l * p.fzV + p.fzK ] - fz[l * p.fzV]) * p.rdz;
    double qnew;
    if (STAGE == 1)      qnew = q_n + dt_dyn * tend;
    else if (STAGE == 2) qnew = (3.0 / 4.0) * q_n + (1.0 / 4.0) * q_s + (1.0 / 4.0) * dt_dyn * tend;
    else                 qnew = (1.0 / 3.0) * q_n + (2.0 / 3.0) * q_s + (2.0 / 3.0) * dt_dyn * tend;
    if ((p.pos_mask >> (l - 5)) & 1u) qnew = fmax(0.0, qnew);
    if (MODE == 0) Sout[so + l * p.sV] = qnew * inv_rho_new;
    else {
      c.tr[l - 5][ci] = qnew;
      if (l - 5 == p.idWV) rho_v = qnew;
      if ((p.mass_mask >> (l - 5)) & 1u) rho_dry -= qnew;
    }
  }
  if (MODE == 1) {
    // the slab holds u = (rho u)/rho etc. -- exactly what convert_dynamics_to_coupler computes (:1929-1932)
    const double press = Sout[so + idT * p.sV];                   // (k_xz_state<3, ., 1> left D13's pressure in the (rho theta)' slot)
    c.rho_d[ci] = rho_dry;                                     // (u, v, w: written to the coupler by k_xz_state<3, ., 1>, like on the fused path)
    c.temp[ci] = press / (rho_dry * p.R_d + rho_v * p.R_v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused tracer stage (<= 4 tracers, nens <= 12): x/z fluxes + FCT + flux divergence + SSPRK3 combine (+ D13) in ONE marching
// kernel -- the x/z tracer fluxes never go to HBM and the separate update pass disappears.
//   pipeline per wave (row j, 64 fused-x lanes, 3 halo cells per side with nens == 1 (+ one loaded cell), 2 with nens > 1), marching k:
//     S1(k)   : x-face fluxes of level k, z-face flux k                       (registers)
//     S2(k-1) : all six face fluxes of cell k-1 are known -> FCT multiplier; x faces and z face k-1 scaled by their donors'
//               multipliers (west/east lanes by DPP, level k-2 carried); partial tendency P = -dFx/dx - dFy/dy
//     S3(k-2) : cell k-2's upper z face (= face k-1) is final -> tendency, SSPRK3 combine, clip, store.
//   y direction: a face's donor may be a cell of another row (another wave), whose multiplier is not known here.  S3 uses
//   the UNSCALED value for incoming y fluxes ("provisional"); a donor that does scale an outgoing y flux (mult < 1: rare)
//   records the flux change in a side array (the idle public x/z flux slots) and sets a bit in its cell's flag byte; the
//   receiver-centred k_tracer_patch then subtracts the recorded change from the two neighbours' new values (exact up to
//   rounding: the provisional value over-estimates the inflow, so a clipped provisional value stays clipped).
//   FY itself is never modified here, so what a receiver used is always the unscaled flux (no race).
//   (Measured, round 2: wave-uniform branches that let the four ghost iterations of a chunk skip the stages nobody reads -- the x
//   reconstructions at k = ka-2 and kb+1, S2 / S3 before their first cell -- save 1 % at run time but cost 30-40 VGPRs (spills in
//   the MODE 1 variant) and the kernel as a whole became 15 % slower.  The loop body stays branch-free.)
// ---------------------------------------------------------------------------------------------------------------
template <int STAGE, int MODE, int T, bool N1, int K, int ORD, bool MT = false>
__global__ __launch_bounds__(256, 2) void k_tracers_fused(DyP p, const double *__restrict__ S, const double *__restrict__ Sn, double *Sout,
                                                       const double *__restrict__ FY, const double *__restrict__ MX,
                                                       const double *__restrict__ MZ, const unsigned char *__restrict__ UPX,
                                                       const unsigned char *__restrict__ UPZ, double *__restrict__ DS,
                                                       double *__restrict__ DN, unsigned char *__restrict__ flags, unsigned int *__restrict__ dirty,
                                                       double dt, double dt_dyn, CouplerPtrs c, int chunk, int tiles_x, int rows4, MemberOff mo) {
  static_assert(N1 || ORD == 5, "the neighbour-load form exists for WENO-5 only");
  static_assert(!MT || (N1 && MODE == 1), "the member-co-located form (see MemberOff) is the D13 variant of the nens == 1 kernel");
  constexpr int HS = (ORD - 1) / 2;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int mt_e = 0, mt_sub = 0;
  if (MT) {                                                    // wave = member mt_e of tile slot mt_sub: everything per member is one stride away
    mt_e = wv & (mo.n - 1); mt_sub = wv >> mo.sh;
    S += mt_e * mo.slab; Sn += mt_e * mo.slab; Sout += mt_e * mo.slab; FY += mt_e * mo.fy; MX += mt_e * mo.mx; UPX += mt_e * mo.mx;
    MZ += mt_e * mo.mz; UPZ += mt_e * mo.mz; DS += mt_e * mo.fx; DN += mt_e * mo.fz; flags += mt_e * mo.cells;
    p.hyc += mt_e * mo.per; p.ce = mt_e;
  }
  const int n = N1 ? 1 : p.nens;
  constexpr int t0 = 0;                                       // one group: all (<= 4) tracers of the cell
  constexpr bool MW_FENCED = Cf<K>::spec;                     // (see MW_SCHED_FENCE)
  const int lane = threadIdx.x & 63;
  const int NXI = p.nx * n;
  // halo cells per side.  nens == 1 (DPP shifts): the multiplier of the cell next to the updated range needs the x flux of
  // its outer face, whose upwind reconstruction reaches a 4th cell -- lanes 0 and 63 load that cell (a clamped all-lane load
  // issued first in the iteration) and the DPP shift hands it on as its edge fill, so 3 lanes per side are enough
  // (58 cells per wave: 7 instead of 8 waves per 400-cell row).  nens > 1: neighbour loads, 2 cells per side.
  const int hw = N1 ? HS + 1 : 2;                             // (WENO-3: 2 lanes per side, 60 cells per wave)
  const int U = 64 - 2 * hw * n;
  constexpr bool LC = true;
  __shared__ double lds_c[8];
  Segment sg;
  block_segment(chunk, p.nz, sg);
  int j, tx;
  if (MT)         { const int rpb = 4 >> mo.sh, jg = (int)(sg.col / tiles_x); tx = (int)(sg.col - (unsigned)jg * tiles_x); j = jg * rpb + mt_sub; }   // rows of one tile
  else if (rows4) { const int jg = (int)(sg.col / tiles_x); tx = (int)(sg.col - (unsigned)jg * tiles_x); j = jg * 4 + (threadIdx.x >> 6); }
  else            { const long long wid = (long long)sg.col * 4 + (threadIdx.x >> 6); j = (int)(wid / tiles_x); tx = (int)(wid - (long long)j * tiles_x); }
  // (MODE 1: D13's pressure comes out of the result slab -- k_xz_state<3, ., 1> evaluated the series and left p in the (rho theta)'
  //  slot -- so this kernel needs neither the series nor its three background values per level.)
  // the uniform doubles of the loop (reciprocal grid spacings, time-step factors) as broadcast LDS reads instead of resident SGPRs
  // (as in k_xz_state: the kernel spills SGPRs to VGPR lanes)
  if (LC && threadIdx.x < 8) {
    const double cdt = (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;
    lds_c[threadIdx.x] = threadIdx.x == 0 ? p.rdx : threadIdx.x == 1 ? p.rdy : threadIdx.x == 2 ? p.rdz : threadIdx.x == 3 ? dt : threadIdx.x == 4 ? cdt : 0.0;
  }
  if (LC) __syncthreads();
  if (j >= p.ny) return;
  const int q = tx * U - hw * n + lane;                       // fused-x index of this lane's cell (halo lanes included)
  const int qq = min(max(q, -3 * n), NXI + 3 * n - 1);        // clamped into the 3-cell halo for addressing
  const int e = N1 ? 0 : ((qq % n) + n) % n;
  const int i = (qq - e) / n;
  const bool interior = (q >= 0) && (q < NXI);
  const bool has_mult = (lane >= (hw - 1) * n) && (lane < 64 - (hw - 1) * n) && interior;   // both x faces of the cell are known
  const bool upd = (lane >= hw * n) && (lane < 64 - hw * n) && interior;                    // the cell this lane completes
  // nens > 1: neighbour offsets, clamped so that every load stays inside the row's 3-cell halo
  const int qa = wrap_xq(p, qq, NXI);                         // the index that is addressed
  const int om2 = wrap_xq(p, max(q - 2 * n, -3 * n), NXI) - qa, om1 = wrap_xq(p, max(q - n, -3 * n), NXI) - qa;
  const int op1 = wrap_xq(p, min(q + n, NXI + 3 * n - 1), NXI) - qa, op2 = wrap_xq(p, min(q + 2 * n, NXI + 3 * n - 1), NXI) - qa;
  const int qm = interior ? q : 0;
  const int qf = (q >= 0 && q < NXI + n) ? q : 0;
  // (lanes 1..62 read lane 0's cell too -- the value is unused there -- so that the load touches two cache lines instead of the
  //  wave's own four: every line costs HBM traffic here, neither L1 nor L2 keeps a level for the three iterations since its first use)
  const int qw0 = __builtin_amdgcn_readfirstlane(wrap_xq(p, max(q - 1, -3), NXI));
  const int opatch = (lane == 63) ? wrap_xq(p, min(q + 1, NXI + 2), NXI) - qa : qw0 - qa;
  const double *col = S + (long long)(5 + t0) * p.sV + (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + qa;
  const long long so_row = (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + qm;     // + (k+HZ)*sK + l*sV
  const long long fxo = (long long)j * p.fxJ + qf;
  const long long fzo = (long long)j * p.fzJ + qm;
  const long long fyo = (long long)j * p.fyJ + qm;
  const bool do_y = !Cf<K>::sim2d(p);
  const int ka = sg.a, kb = sg.b;
  const int k_lo = max(ka - 1, 0);                            // cells k_lo .. k_hi get all six fluxes (FCT multiplier)
  const int k_hi = min(kb, p.nz - 1);
  const int kstart = max(ka - 2, 0);
  double w[T][ORD], nxt[T], ct[T];
  double fxp[T], fzp[T], multp[T], szf[T], P[T];
  double wkm2[T];                                             // WENO-3: level k-2 has left the 3-level window and is carried instead
  double rhos2 = 0;

#pragma unroll
  for (int v = 0; v < T; v++) {
    ct[v] = 0; fxp[v] = fzp[v] = szf[v] = P[v] = 0; multp[v] = 1; wkm2[v] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) w[v][s] = load_zlevel(p, col + (long long)v * p.sV, kstart - HS + s, false);
  }
  // The patch cell (beyond lane 0 / lane 63) of a level is requested one iteration ahead: the x reconstruction is the first
  // consumer of an iteration, and vmcnt retires in order -- waiting for it would wait for everything requested before it.
  double xpn[T];
#pragma unroll
  for (int v = 0; v < T; v++) xpn[v] = N1 ? col[(long long)v * p.sV + (long long)(min(kstart, p.nz - 1) + p.HZ) * p.sK + opatch] : 0.0;
#pragma unroll
  for (int v = 0; v < T; v++) landed(w[v]);
  landed(xpn);
#if MW_ZERO_SKIP
  // Round 5: a tracer that is exactly ZERO over a wavefront's whole stencil (cloud and rain outside the storm; simple_city's vapour) has
  // edge values exactly 0 -- weno5_edges_fast of five zeros is +0 -- so the reconstruction is skipped, wave-uniformly and bit-neutrally.
  // zm[v]: bit s = "window slot s of tracer v is non-zero in some lane", kept in a scalar register and shifted with the window: one
  // v_cmp per tracer and level when the new level has landed (and one for the patch cell of the x stencil), no test on the way in.
  unsigned zm[T]; bool xzc[T], xzn[T];
#pragma unroll
  for (int v = 0; v < T; v++) {
    zm[v] = 0; xzc[v] = false;
#pragma unroll
    for (int s = 0; s < ORD; s++) zm[v] |= (__any(w[v][s] != 0.0) ? 1u : 0u) << s;
    xzn[v] = __any(xpn[v] != 0.0);
    if (!tracer_may_vanish<K>(p, v)) { zm[v] = ~0u; xzn[v] = true; }
  }
  // zero-row map of this stage (see k_zero_rows): bit i of zq_mask = "iteration kstart + i of this wave's row may touch something
  // non-zero of a tracer that can vanish" -- one word per iteration, fetched by the lanes here and kept as a scalar mask
  // (iterations beyond the 64th: bit set)
  // (the form an iteration takes follows from the map alone, for any chunk length: k_y_all relies on it -- it does not store the y fluxes
  //  that only LEAN iterations would read)
  unsigned long long zq_mask = ~0ull;
  const bool zq_on = (p.zq != nullptr) && p.zero_skip;
  const unsigned *zq_own = p.zq + (MT ? mt_e * mo.zq : 0);       // (members in one workgroup: this wave's member)
  // (the x segments this wave's lanes, halo lanes and patch cells lie in -- k_zero_rows: a segment bit is set when a tracer that can vanish may be
  //  non-zero there; the row's other tiles may be busy while this one is lean)
  const unsigned zseg = zr_seg_mask(tx * U - (hw + 1) * n, tx * U - hw * n + 63 + n, NXI, p.wrap_x != 0);
  auto zq_fetch = [&](int k_first) __attribute__((always_inline)) {
    const unsigned wq = zq_own[(long long)min(k_first + lane, p.nz - 1) * p.zq_ld + j + MW_ZR_HALO];
    zq_mask = __ballot((wq & zseg) != 0u) | ~__ballot(true);     // (all 64 lanes are here; a missing one would keep its bit)
  };
  if (zq_on) zq_fetch(kstart);
  // (bit i of zc_mask = "the row that iteration kstart + i stores to may hold something non-zero": the coupler's arrays in MODE 1 -- map MC,
  //  the level that is stored -- the result slab otherwise -- the previous sub-cycle's map of this stage, the iteration that stored it)
  unsigned long long zc_mask = ~0ull;
  const unsigned *zc_map = MT ? nullptr : (MODE == 1) ? p.zqc : p.zqp;
  const bool zc_on = zq_on && (zc_map != nullptr);
  auto zc_fetch = [&](int k_first) __attribute__((always_inline)) {
    const int kq_ = (MODE == 1) ? min(max(k_first + lane - 2, ka), kb - 1) : min(k_first + lane, p.nz - 1);
    const unsigned wc = zc_map[(long long)kq_ * p.zq_ld + j + MW_ZR_HALO];
    zc_mask = __ballot((wc & zseg) != 0u) | ~__ballot(true);
  };
  if (zc_on) zc_fetch(kstart);
  bool zc_store = true;
#endif
  // The loop body is branch-free apart from predicated stores and the rare limiter paths: every load uses a clamped
  // (always valid) address and is issued at the top, the z reconstruction (registers only) runs while they are in flight.
  // Iterations outside a quantity's range compute values that are never stored or carried into a used result.
  // Two forms of the loop body (round 5, see k_zero_rows): FULL, and LEAN for an iteration in which every tracer that can vanish IS zero
  // in everything the iteration touches -- window, entering level, patch cell, y fluxes, q^n -- and in everything the three iterations
  // before touched (their carries are zero then): those tracers issue no load and no arithmetic, their new value is 0.  The stage's
  // row map says so (zq_mask).  ACT = the tracers the form works on; one wave-uniform branch per iteration picks the form.
  constexpr unsigned FULLM = (1u << T) - 1u, VANM = (K == 1) ? (FULLM & ~1u) : FULLM;     // (VANM = tracer_may_vanish<K>)
  auto body = [&](auto act_c, const int k) __attribute__((always_inline)) {
    constexpr unsigned ACT = decltype(act_c)::value;
#define MW_ACT(v_) ((ACT >> (v_)) & 1u)
    const bool cell = (k < p.nz);
    const int kp = k - 1, ku = k - 2;
    const bool s2cell = (kp >= k_lo) && (kp <= k_hi);          // cell kp has all six fluxes (kp == nz: only the top face is scaled)
    const bool s3 = (ku >= ka) && (ku < kb);
    // (addresses of iterations outside a quantity's range are clamped INTO the range the chunk needs anyway -- x faces and cells
    //  k_lo .. k_hi, updated cells ka .. kb-1 -- so that a ghost iteration re-reads a line the neighbouring iteration uses, not a
    //  level of the chunk below or above: those lines would come from HBM for nothing)
    const int kx = min(max(k, k_lo), p.nz - 1), kz = min(max(k, k_lo), p.nz);
    const int kpc = min(max(kp, k_lo), k_hi), kuc = min(max(ku, ka), kb - 1);
    // ------------------------------------------------ loads of this iteration
    const int kn = min(k + HS + 1, p.nz + p.HZ - 1);
    double xpatch[T];                                            // level k, the cell beyond lane 0 / lane 63
#pragma unroll
    for (int v = 0; v < T; v++) xpatch[v] = MW_ACT(v) ? xpn[v] : 0.0;
#if MW_ZERO_SKIP
#pragma unroll
    for (int v = 0; v < T; v++) xzc[v] = xzn[v];
#endif
#pragma unroll
    for (int v = 0; v < T; v++) if (MW_ACT(v)) xpn[v] = N1 ? col[(long long)v * p.sV + (long long)(min(k + 1, p.nz - 1) + p.HZ) * p.sK + opatch] : 0.0;
#pragma unroll
    for (int v = 0; v < T; v++) if (MW_ACT(v)) nxt[v] = load_zlevel(p, col + (long long)v * p.sV, kn, false);
    const double mx = MX[(long long)kx * p.fxK + fxo];
    const int upx = UPX[(long long)kx * p.fxK + fxo];
    const double mz = MZ[(long long)kz * p.fzK + fzo];
    const int upz = UPZ[(long long)kz * p.fzK + fzo];
    // (loaded values are only NAMED in this section; the first arithmetic on them -- "+ hyc" -- comes behind the reconstructions:
    //  an add in here waits for its operands in the middle of the loads that are still to be issued)
    const double rhop_raw = S[(long long)idR * p.sV + (long long)(kpc + p.HZ) * p.sK + so_row], hyc_p = p.hyc[kpc * p.nens + e];
    double fys[T], fyn[T];
#pragma unroll
    for (int v = 0; v < T; v++) {
      const double *fy = FY + (long long)(5 + t0 + v) * p.fyV + (long long)kpc * p.fyK + fyo;
      fys[v] = 0.0; fyn[v] = 0.0;
      if (MW_ACT(v)) { fys[v] = do_y ? fy[0] : 0.0; fyn[v] = do_y ? fy[p.fyJ] : 0.0; }
    }
    const long long so = (long long)(kuc + p.HZ) * p.sK + so_row;
    const double hyc_u = p.hyc[kuc * p.nens + e];
    const double rho_new_raw = Sout[so + idR * p.sV];
    double qn_[T], rho_n_raw = 0, st_T = 0;
#pragma unroll
    for (int v = 0; v < T; v++) qn_[v] = 0;
    if (STAGE != 1) {
      rho_n_raw = Sn[so + idR * p.sV];
#pragma unroll
      for (int v = 0; v < T; v++) if (MW_ACT(v)) qn_[v] = Sn[so + (5 + t0 + v) * p.sV];
    }
    if (MODE == 1) st_T = Sout[so + idT * p.sV];              // (u, v, w were written to the coupler by k_xz_state<3, ., 1>)
    // nens > 1: the x-stencil neighbours of level k come from memory (issued here, with the iteration's other loads)
    double nbw2[T], nbw1[T], nbe1[T], nbe2[T];
#pragma unroll
    for (int v = 0; v < T; v++) {
      const double *lvl = col + (long long)v * p.sV + (long long)(kx + p.HZ) * p.sK;
      nbw2[v] = nbw1[v] = nbe1[v] = nbe2[v] = 0;
      if (!N1 && MW_ACT(v)) { nbw2[v] = lvl[om2]; nbw1[v] = lvl[om1]; nbe1[v] = lvl[op1]; nbe2[v] = lvl[op2]; }
    }
    // ------------------------------------------------ S1: z reconstruction (registers only), then the fluxes of level k
    if (MW_FENCED) MW_SCHED_FENCE();
    double te[T], fxn[T], fzn[T], be_[T], xe_[T];
#pragma unroll
    for (int v = 0; v < T; v++) {                                // all reconstructions first: they need no loaded operand
      if (!MW_ACT(v)) { be_[v] = 0.0; te[v] = 0.0; continue; }
#if MW_ZERO_SKIP
      if (zm[v] == 0u) { be_[v] = 0.0; te[v] = 0.0; }
      else
#endif
      weno_window_edges<ORD>(w[v], be_[v], te[v]);
      if (!cell) be_[v] = 0;
      if (MW_FENCED) MW_SCHED_FENCE();
    }
    {
      const bool quirk = bc_mode_x<K>(p, i) == 3;
#pragma unroll
      for (int v = 0; v < T; v++) {
        if (!MW_ACT(v)) { xe_[v] = 0.0; continue; }
#if MW_ZERO_SKIP
        if (N1 && !quirk && !((zm[v] >> HS) & 1u) && !xzc[v]) {   // the wave's whole x stencil of this level is zero
          be_[v] = upz ? be_[v] : ct[v];
          xe_[v] = 0.0;
          if (MW_FENCED) MW_SCHED_FENCE();
          continue;
        }
#endif
        double c0 = w[v][HS], m2 = nbw2[v], m1 = nbw1[v], p1 = nbe1[v], p2 = nbe2[v];
        if (N1) {                                              // whole-wave DPP shifts
          m1 = dpp_mov_old<0x138>(xpatch[v], c0); p1 = dpp_mov_old<0x130>(xpatch[v], c0);   // lanes 0 / 63 keep the loaded cell
          if (ORD == 5) { m2 = from_west<true>(m1, lane, 1); p2 = from_east<true>(p1, lane, 1); }
        }
        double we, ee;
        if (ORD == 3) weno3_edges_fast(m1, c0, p1, we, ee);
        else          weno5_edges_fast(m2, m1, c0, p1, p2, we, ee);
        if (__builtin_expect(quirk, 0)) {
          const double *qv = col + (long long)v * p.sV + (long long)(kx + p.HZ) * p.sK - (long long)p.nx * n; double r_, q_[ORD];
#pragma unroll
          for (int s = 0; s < ORD; s++) q_[s] = qv[(s - HS) * n];
          weno_window_edges<ORD>(q_, we, r_);
        }
        double Lq = from_west<N1>(ee, lane, n);
        be_[v] = upz ? be_[v] : ct[v];
        xe_[v] = upx ? we : Lq;
        if (MW_FENCED) MW_SCHED_FENCE();
      }
    }
#pragma unroll
    for (int v = 0; v < T; v++) { fzn[v] = 0.0; fxn[v] = 0.0; if (MW_ACT(v)) { fzn[v] = mz * be_[v]; fxn[v] = mx * xe_[v]; } }
    // ------------------------------------------------ S2: cell k-1 -> multiplier, scaled faces, partial tendency
    const double rhop = rhop_raw + hyc_p;
    double rho_new = rho_new_raw + hyc_u, rho_n = (STAGE != 1) ? rho_n_raw + hyc_u : 0.0;
    double szn[T], Pn[T];
    {
      const bool rec = upd && s2cell && (kp >= ka) && (kp < kb);           // the unique owner of cell (kp, j, q)
      unsigned fl = 0;
#pragma unroll
      for (int v = 0; v < T; v++) {
#pragma clang fp contract(off)
        szn[v] = 0.0; Pn[v] = 0.0;
        if (!MW_ACT(v)) continue;
        const double fe = from_east<N1>(fxp[v], lane, n);
        double mult = 1.0;
        {
          // the cell volume dx dy dz multiplies both sides of the reference's test (:506-511) and cancels in the multiplier
          const double mass_available = fmax(w[v][HS - 1] * rhop, 0.0);      // (window slot HS - 1 = level k-1)
          const double out_x = (fmax(fe, 0.0) - fmin(fxp[v], 0.0)) * (LC ? lds_c[0] : p.rdx);
          const double out_y = (fmax(fyn[v], 0.0) - fmin(fys[v], 0.0)) * (LC ? lds_c[1] : p.rdy);
          const double out_z = (fmax(fzn[v], 0.0) - fmin(fzp[v], 0.0)) * (LC ? lds_c[2] : p.rdz);
          const double mass_out = (out_x + out_y + out_z) * (LC ? lds_c[3] : dt);
          if (__builtin_expect(s2cell && has_mult && Cf<K>::positive(p, t0 + v) && mass_out > mass_available, 0))
            mult = mass_available / mass_out;
        }
        const double mult_w = from_west<N1>(mult, lane, n);
        const double F = fxp[v];
        const double sFw = (F > 0) ? F * mult_w : F * mult;                  // west face, scaled by its donor
        const double sFe = from_east<N1>(sFw, lane, n);
        const double G = fzp[v];
        szn[v] = (G > 0) ? G * multp[v] : G * mult;                          // z face kp, scaled by its donor
        const double ys = (fys[v] < 0) ? fys[v] * mult : fys[v];             // outgoing y faces scaled, incoming provisional
        const double yn = (fyn[v] > 0) ? fyn[v] * mult : fyn[v];
        Pn[v] = -(sFe - sFw) * (LC ? lds_c[0] : p.rdx) - (yn - ys) * (LC ? lds_c[1] : p.rdy);
        if (__builtin_expect(rec && mult < 1.0, 0)) {
          if (fys[v] < 0) { DS[(long long)(5 + t0 + v) * p.fxV + (long long)kp * p.fxK + (long long)j * p.fxJ + q] = ys - fys[v]; fl |= 1u << (2 * v); }
          if (fyn[v] > 0) { DN[(long long)(5 + t0 + v) * p.fzV + (long long)kp * p.fzK + (long long)j * p.fzJ + q] = yn - fyn[v]; fl |= 2u << (2 * v); }
        }
        multp[v] = mult;
      }
#pragma unroll
      for (int v = 0; v < T; v++) if (MW_ACT(v)) { landed(nxt[v]); landed(xpn[v]); landed(qn_[v]); }   // in front of the iteration's stores (see landed())
      landed(rho_new); landed(rho_n); landed(st_T);
      if (rec && do_y) flags[(long long)(kp * p.ny + j) * NXI + q] = (unsigned char)fl;     // 2 bits per tracer: (south, north) face scaled
      if (__builtin_expect(fl != 0u, 0)) *dirty = 1u;           // (only set inside `rec`) lets k_tracer_patch return at once when nothing was scaled
    }
    // ------------------------------------------------ S3: cell k-2 -> new value
    {
#pragma clang fp contract(off)
      const bool st = s3 && upd;
      const long long ci = (long long)(kuc * p.ny + j) * NXI + qm;                 // (row number and row length are 32-bit: one 32 x 32 -> 64 multiply)
      const double inv_rho_new = fast_rcp(rho_new);
      double rho_dry = rho_new, rho_v = 0;
#pragma unroll
      for (int v = 0; v < T; v++) {
        if (!MW_ACT(v)) {                                        // LEAN form: the tracer is zero and stays zero
          if (MODE == 0) { if (st && zc_store) Sout[so + (5 + t0 + v) * p.sV] = 0.0; }     // (not over a row that is zero already: see zc_mask)
          else           { if (st && zc_store) c.tr[v][cpl(p, ci)] = 0.0; }
          continue;
        }
        const double q_s = (ORD == 3 ? wkm2[v] : w[v][0]) * rhos2;           // level k-2
        const double q_n = (STAGE == 1) ? q_s : qn_[v] * rho_n;
        const double tend = P[v] - (szn[v] - szf[v]) * (LC ? lds_c[2] : p.rdz);
        double qnew;
        const double cdt = LC ? lds_c[4] : (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;   // (one product, as before)
        if (STAGE == 1)      qnew = q_n + cdt * tend;
        else if (STAGE == 2) qnew = (3.0 / 4.0) * q_n + (1.0 / 4.0) * q_s + cdt * tend;
        else                 qnew = (1.0 / 3.0) * q_n + (2.0 / 3.0) * q_s + cdt * tend;
        if (Cf<K>::positive(p, t0 + v)) qnew = fmax(0.0, qnew);
        if (MODE == 0) { if (st) Sout[so + (5 + t0 + v) * p.sV] = qnew * inv_rho_new; }
        else {
          if (st) c.tr[v][cpl(p, ci)] = qnew;
          if (Cf<K>::is_wv(p, v)) rho_v = qnew;
          if (Cf<K>::adds_mass(p, v)) rho_dry -= qnew;
        }
      }
      if (MODE == 1 && st) {
        // D13 (:1929-1935): p = C0 (rho theta)^gamma with rho theta = hy + (rho theta)' -- the same series around the hydrostatic
        // state as in the Riemann solver (device pow for large perturbations); rho*(rho theta / rho) differs from rho theta by rounding
        const double press = st_T;                             // (k_xz_state<3, ., 1> left p in the (rho theta)' slot)
        c.rho_d[cpl(p, ci)] = rho_dry;
        c.temp[cpl(p, ci)] = press / (rho_dry * p.R_d + rho_v * p.R_v);
      }
    }
    // ------------------------------------------------ carries
    rhos2 = rhop;
#if MW_ZERO_SKIP
#pragma unroll
    for (int v = 0; v < T; v++) {
      if (MW_ACT(v) && tracer_may_vanish<K>(p, v)) {
        zm[v] = (zm[v] >> 1) | ((__any(nxt[v] != 0.0) ? 1u : 0u) << (ORD - 1));
        xzn[v] = __any(xpn[v] != 0.0);
      }
    }
#endif
#pragma unroll
    for (int v = 0; v < T; v++) {
      if (!MW_ACT(v)) continue;                                  // (LEAN form: window and carries are zero and stay zero)
      fxp[v] = fxn[v]; fzp[v] = fzn[v]; szf[v] = szn[v]; P[v] = Pn[v];
      ct[v] = te[v];
      if (ORD == 3) wkm2[v] = w[v][0];                         // level k-1 is level (k+1)-2 of the next iteration
#pragma unroll
      for (int s = 0; s + 1 < ORD; s++) w[v][s] = w[v][s + 1];
      w[v][ORD - 1] = nxt[v];
    }
#undef MW_ACT
  };
  for (int k = kstart; k <= kb + 1; k++) {
    bool lean = false;
#if MW_ZERO_SKIP
    { const int it = k - kstart;
      if (zq_on && it > 0 && (it & 63) == 0) { zq_fetch(k); if (zc_on) zc_fetch(k); }         // (chunks of more than 64 levels)
      lean = !((zq_mask >> (it & 63)) & 1ull);
      zc_store = (zc_mask >> (it & 63)) & 1ull; }
#endif
    if (lean) body(std::integral_constant<unsigned, (FULLM & ~VANM)>{}, k);
    else      body(std::integral_constant<unsigned, FULLM>{}, k);
  }
}

// Receiver-centred correction for y faces whose donor (the row above or below) scaled them: see k_tracers_fused.
// Two phases per wavefront (round 6; the byte-per-lane scan of rounds 2-5 took 40-60 us per launch on the mature storm, where one cell in a
// few hundred is flagged -- the scan was the kernel):
//   scan    : thread = (level, row, 8 consecutive cells): the flag bytes of the two neighbouring rows come as ONE 8-byte word each (512
//             contiguous bytes per request); a wavefront whose 2 x 512 bytes are clear leaves at once;
//   correct : the wavefront's 512 cells in eight passes of 64, lane = cell (the words handed over by ds_bpermute), so that flagged cells that
//             are neighbours in x -- cloud rims are -- share their requests as they did when a lane was a cell; passes without a flag are
//             skipped wave-uniformly.
#define MW_PATCH_CELLS 8
template <int STAGE, int MODE>
__global__ __launch_bounds__(256) void k_tracer_patch(DyP p, double *Sout, const unsigned char *__restrict__ flags,
                                                      const double *__restrict__ DS, const double *__restrict__ DN, double dt_dyn,
                                                      CouplerPtrs c, const unsigned int *__restrict__ dirty, unsigned int *dirty_next) {
#pragma clang fp contract(off)
  // `dirty` is this launch's "some y face was scaled" word, written by the k_tracers_fused before it; the other word is
  // cleared here for the next stage's k_tracers_fused (launches on the tracer stream are serial).
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dirty_next = 0u;
  if (*dirty == 0u) return;
  const int NXI = p.nx * p.nens;
  const int G = (NXI + MW_PATCH_CELLS - 1) / MW_PATCH_CELLS;      // groups of 8 cells per row
  const long long nthr = (long long)p.ny * G;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  // byte m of wN = the northern neighbour's flags (it scaled its south face = cell m's north face), of wS = the southern neighbour's
  unsigned long long wN = 0ull, wS = 0ull;
  if (t < nthr) {
    const int j = (int)(t / G), ie0 = (int)(t - (long long)j * G) * MW_PATCH_CELLS;
    const long long ci0 = ((long long)k * p.ny + j) * NXI + ie0;
    if (((NXI & (MW_PATCH_CELLS - 1)) == 0) && ((reinterpret_cast<unsigned long long>(flags) & 7ull) == 0ull)) {
      if (j + 1 < p.ny) wN = *reinterpret_cast<const unsigned long long *>(flags + ci0 + NXI);
      if (j >= 1)       wS = *reinterpret_cast<const unsigned long long *>(flags + ci0 - NXI);
    } else {                                                     // (rows that are not a multiple of 8 cells: byte by byte)
      for (int m = 0; m < MW_PATCH_CELLS; m++) {
        if (ie0 + m >= NXI) break;
        if (j + 1 < p.ny) wN |= (unsigned long long)flags[ci0 + NXI + m] << (8 * m);
        if (j >= 1)       wS |= (unsigned long long)flags[ci0 - NXI + m] << (8 * m);
      }
    }
  }
  wN &= 0x5555555555555555ull; wS &= 0xAAAAAAAAAAAAAAAAull;        // (the bits a receiver looks at)
  const unsigned long long busy = __ballot((wN | wS) != 0ull);     // lanes (= 8-cell groups) with a flagged cell
  if (busy == 0ull) return;
  const int lane = threadIdx.x & 63;
  const long long t0 = t - lane;                                 // the wavefront's first group
  const double cdt = (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;
#pragma unroll 1
  for (int s = 0; s < 8; s++) {                                  // pass s: the 64 cells of groups 8 s .. 8 s + 7
    if (((busy >> (8 * s)) & 0xFFull) == 0ull) continue;
    const int src = 8 * s + (lane >> 3), sh = 8 * (lane & 7);
    const unsigned fN = (unsigned)(__shfl(wN, src, 64) >> sh) & 0xFFu, fS = (unsigned)(__shfl(wS, src, 64) >> sh) & 0xFFu;
    if ((fN | fS) == 0u) continue;
    const long long ts = t0 + src;                               // the group this lane's cell belongs to (flagged => ts < nthr, ie < NXI)
    const int j = (int)(ts / G), ie = (int)(ts - (long long)j * G) * MW_PATCH_CELLS + (lane & 7);
    const long long ci = ((long long)k * p.ny + j) * NXI + ie;
    const int e = ie % p.nens;
    const long long so = (long long)(k + p.HZ) * p.sK + (long long)(j + p.HY) * p.sJ + (long long)p.HX * p.nens + ie;
    const double rho_new = Sout[so + idR * p.sV] + p.hyc[k * p.nens + e];
    const double inv_rho_new = fast_rcp(rho_new);
    for (int v = 0; v < p.nt && v < 4; v++) {
      double dN = 0, dS = 0;
      if ((fN >> (2 * v)) & 1u) dN = DS[(long long)(5 + v) * p.fxV + (long long)k * p.fxK + (long long)(j + 1) * p.fxJ + ie];
      if ((fS >> (2 * v)) & 2u) dS = DN[(long long)(5 + v) * p.fzV + (long long)k * p.fzK + (long long)(j - 1) * p.fzJ + ie];
      if (dN == 0 && dS == 0) continue;
      const double corr = cdt * ((dN - dS) * p.rdy);                // tend' = tend - (dN - dS)/dy
      if (MODE == 0) {
        const double qp = Sout[so + (5 + v) * p.sV] * rho_new;
        Sout[so + (5 + v) * p.sV] = fmax(0.0, qp - corr) * inv_rho_new;
      } else {
        c.tr[v][cpl(p, ci)] = fmax(0.0, c.tr[v][cpl(p, ci)] - corr);
      }
    }
    if (MODE == 1) {
      double rho_dry = rho_new, rho_v = 0;
      for (int v = 0; v < p.nt; v++) {
        const double qv = c.tr[v][cpl(p, ci)];
        if (v == p.idWV) rho_v = qv;
        if ((p.mass_mask >> v) & 1u) rho_dry -= qv;
      }
      const double press = Sout[so + idT * p.sV];                   // (k_xz_state<3, ., 1> left D13's pressure in the (rho theta)' slot)
      c.rho_d[cpl(p, ci)] = rho_dry;
      c.temp[cpl(p, ci)] = press / (rho_dry * p.R_d + rho_v * p.R_v);
    }
  }
}

} // namespace mw

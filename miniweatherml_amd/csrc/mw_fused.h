// =====================================================================================================
// mw_fused.h -- k_state_xyz: the state variables' COMPLETE stage (x, y and z faces, tendencies, SSPRK3 combine) in one z-marching
// kernel, the y stencil from an LDS-staged tile of the workgroup's rows.  Included by mw_dycore.hip (after mw_march.h).
//
// Round 3's review asked for this kernel (VERDICT.md "Next round" 1): k_y_all's state part and k_xz_state in one launch, so that the y
// tendencies (80 B per cell and stage written + read) never reach HBM.  Round 4 first bounded it with timing builds of k_xz_state
// (the bounds say "tie at best") and then built it literally, to have the measurement itself:
//   workgroup = 4 waves = rows j0 .. j0+3 of ONE x tile (each wave as in k_xz_state: 64 x lanes, 3 halo lanes per side, DPP shifts, a
//   5-level z window in registers), marching k.  Per level:
//     (1) every wave writes its level-k values into the tile (rows 3..6) and its share of the 6 halo rows j0-3..j0-1, j0+4..j0+6
//         (prefetched one level ahead: 30 row loads per level over 4 waves) into rows 0..2, 7..9                       | barrier
//     (2) y reconstruction of the own cell from the tile (4 LDS reads per variable) -> south / north edge values; the north edge goes
//         to LDS.  TILE-EDGE TASKS: the north edge of row j0-1 and the south edge of row j0+4 are needed too (the tile's outer
//         faces): 10 reconstructions per level spread over the 4 waves (3, 3, 2, 2)                                     | barrier
//     (3) Riemann solve of the own SOUTH face (L = the lower neighbour's north edge from LDS, R = own south edge), upwind mass flux +
//         selector -> M_y / UP_y (HBM: the tracers' y launch reads them), the five face fluxes -> LDS; wave 3 also solves the tile's
//         top face j0+4                                                                                                 | barrier
//     (4) y tendency = -(F(j+1) - F(j)) / dy with F(j+1) from the wave above; then x, z, finalise exactly as k_xz_state.
//   Same arithmetic as k_y_all + k_xz_state statement by statement: results are BITWISE those of the production path.
// Exists for the folded configurations (Cf<K>, K = 1 / 2: periodic y owned by one rank -> the row index wraps), nens = 1, WENO-5 and WENO-3,
// ny a multiple of W; selected with MW_FUSED_STATE=4 | 8 = W row-waves per workgroup (the stage is then: k_state_xyz -> k_y_tracers ->
// k_tracers_fused).  MEASURED: bitwise equal, 1.0-1.35 GB per stage less HBM traffic, and 30 % slower than the two launches it
// replaces (1.30 + 0.20 ms against 1.14 ms): 1.9-2.1 k instructions per cell at 0.64-0.69 VALU busy; a tie at WENO-3.  Not the default.
// reference: dynamics_euler_stratified_wenofv.h:271-388 (D6), :395-485 (D9), :519-551 (D11), :121-174 (D12) -- all three directions
// in one lambda there.
// =====================================================================================================
#pragma once

namespace mw {

template <int STAGE, int MODE, int K, int W, int ORD>
__global__ __launch_bounds__(64 * W, W == 4 ? 2 : 1) void k_state_xyz(DyP p, const double *__restrict__ S, const double *__restrict__ Sn,
                                                   double *__restrict__ Sout, double *__restrict__ MX, double *__restrict__ MZ,
                                                   unsigned char *__restrict__ UPX, unsigned char *__restrict__ UPZ,
                                                   double *__restrict__ MY, unsigned char *__restrict__ UPY, double dt_stage, double dt_dyn,
                                                   int chunk, int tiles_x, double *__restrict__ cu, double *__restrict__ cv,
                                                   double *__restrict__ cw) {
  static_assert(ORD == 5 || ORD == 3, "the marching kernels exist for WENO orders 5 and 3");
  constexpr int HS = (ORD - 1) / 2, HR = HS + 1;   // stencil half width; halo rows per side (hs + 1: the neighbour's edge value is rebuilt here)
  constexpr bool N1 = true;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __shared__ double lds_c[8];
  static_assert(W == 4 || W == 8, "4 row-waves (two workgroups per CU) or 8 (one)");
  constexpr int NPAIR = 2 * HR * 5;              // halo (row, variable) pairs per level: 30 / 20
  constexpr int NP = (NPAIR + W - 1) / W;        // ... per wave
  constexpr int NT = (10 + W - 1) / W;           // tile-edge reconstructions per wave
  __shared__ double lds_xpart[5][64 * W], lds_fzprev[5][64 * W];
  __shared__ double lds_tile[5][W + 2 * HR][64]; // [variable][tile row: 0..HR-1 = rows j0-HR..j0-1, HR..HR+W-1 = the workgroup's rows, then j0+W..j0+W+HR-1][x lane]
  __shared__ double lds_ne[5][W + 1][64];       // north edge values of rows j0-1 (slot 0) .. j0+W-1 (slot W)
  __shared__ double lds_set[5][64];             // south edge values of row j0+W (the tile's top face)
  __shared__ double lds_fy[5][W + 1][64];       // y face fluxes of faces j0 (slot 0) .. j0+W (slot W)
  extern __shared__ double lds_hp_all[];
  const BlockXY blk = xcd_block();
  const int ka = (int)blk.y * chunk, kb = min(ka + chunk, p.nz);
  XzGeom g = xz_geom<true, ORD>(p, blk.x, ka, kb, tiles_x, /*rows4*/ 1);
  g.j = (int)(blk.x / (unsigned)tiles_x) * W + wv;             // (W rows of one x tile per workgroup)
  g.valid = g.j < p.ny;
  double *lds_hp = lds_hp_all;
  {
    const int nrow = g.kb - g.kstart + 1;
    for (int i = threadIdx.x; i < nrow * 8; i += 64 * W) lds_hp[i] = p.hypk[(long long)g.kstart * 8 + i];
    if (threadIdx.x < 8) {
      const double cdt = (STAGE == 1) ? dt_dyn : (STAGE == 2) ? (1.0 / 4.0) * dt_dyn : (2.0 / 3.0) * dt_dyn;
      lds_c[threadIdx.x] = threadIdx.x == 0 ? p.rdx : threadIdx.x == 1 ? p.rdz : threadIdx.x == 2 ? cdt : threadIdx.x == 3 ? -p.grav : 0.0;
    }
    __syncthreads();
  }
  // (ny is a multiple of 4 here: every wave of a workgroup has a row, and all four take part in every barrier)
  const int n = 1, lane = g.lane, NXI = g.NXI, j = g.j, q = g.q;
  const int j0 = j - wv;
  const double *col = S + (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qa;           // level k at col + (k+HZ)*sK
  const long long cell0 = (long long)j * NXI + g.qc;                                           // + k*ny*NXI
  const long long slab0 = (long long)(j + p.HY) * p.sJ + (long long)p.HX * n + g.qc;           // + (k+HZ)*sK
  const long long planeC = (long long)p.ny * NXI;
  // the halo rows this wave brings in: pairs (halo row h = 0..5, variable v), pair id = wv + 4 i
  // (wave-uniform row / variable offsets + the lane's x offset: scalar base + one shared VGPR offset per load)
  long long hoff[NP];
  int hrow[NP], hvar[NP];
#pragma unroll
  for (int i = 0; i < NP; i++) {
    const int id = min(wv + W * i, NPAIR - 1);                  // (ids beyond the last pair repeat it: harmless duplicates)
    hrow[i] = id / 5; hvar[i] = id - hrow[i] * 5;
    const int jr = wrap_row(p, j0 + (hrow[i] < HR ? hrow[i] - HR : hrow[i] - HR + W));
    hoff[i] = (long long)hvar[i] * p.sV + (long long)(jr + p.HY) * p.sJ + (long long)p.HX * n;
  }
#define MW_HALO_LOAD(i_, kl_) (S + hoff[i_] + (long long)((kl_) + p.HZ) * p.sK)[g.qa]
  double w[5][ORD], nxt[5], ct[5], hal[NP];
#pragma unroll
  for (int v = 0; v < 5; v++) {
    ct[v] = 0; lds_fzprev[v][threadIdx.x] = 0; lds_xpart[v][threadIdx.x] = 0;
#pragma unroll
    for (int s = 0; s < ORD; s++) w[v][s] = load_zlevel<K>(p, col + (long long)v * p.sV, g.kstart - HS + s, v == idW);
  }
#pragma unroll
  for (int i = 0; i < NP; i++) hal[i] = MW_HALO_LOAD(i, min(max(g.kstart, g.ka), g.kb - 1));   // (levels of the chunk: interior, no z rule to apply)
#pragma unroll
  for (int v = 0; v < 5; v++) landed(w[v]);
  landed(hal);
  // The y phase of level kl (the window's centre holds it, `hal` the wave's share of the halo rows): tile, barriers, edges, faces -> the
  // five y tendencies of this lane's cell.  It runs at the END of iteration kl - 1 (for the next level) -- where the x / z temporaries
  // are dead -- and once in front of the loop when the chunk starts at the wall; placed first in the iteration it spilled 26-50 VGPRs.
  auto y_phase = [&](int kl, double (&tyo)[5]) __attribute__((always_inline)) {
#pragma unroll
      for (int v = 0; v < 5; v++) lds_tile[v][HR + wv][lane] = w[v][HS];
#pragma unroll
      for (int i = 0; i < NP; i++) lds_tile[hvar[i]][hrow[i] < HR ? hrow[i] : hrow[i] + W][lane] = hal[i];
      __syncthreads();
      const double *hq_ = lds_hp + (kl - g.kstart) * 8;
      const double hyr = hq_[0], hyt = hq_[1], p0 = hq_[2], ihyt = hq_[3];
      double se[5], ne[5];
#pragma unroll
      for (int v = 0; v < 5; v++) {                             // (one variable at a time: all LDS reads hoisted in front of the arithmetic would spill)
        MW_SCHED_FENCE();
        if (ORD == 3) weno3_edges_fast(lds_tile[v][HR - 1 + wv][lane], w[v][HS], lds_tile[v][HR + 1 + wv][lane], se[v], ne[v]);
        else          weno5_edges_fast(lds_tile[v][1 + wv][lane], lds_tile[v][2 + wv][lane], w[v][HS], lds_tile[v][4 + wv][lane], lds_tile[v][5 + wv][lane], se[v], ne[v]);
        lds_ne[v][1 + wv][lane] = ne[v];
      }
      MW_SCHED_FENCE();
      // tile-edge tasks: t < 5: north edge of row j0-1 (variable t); t >= 5: south edge of row j0+W (variable t-5).  wave w: t = w, w+W, ...
#pragma unroll
      for (int i = 0; i < NT; i++) {
        const int t = wv + W * i;
        if (t < 10) {                                          // (wave-uniform)
          const int v = t < 5 ? t : t - 5, r0 = t < 5 ? 0 : W + 1; // tile rows r0 .. r0+ORD-1: the stencil of tile row HR-1 (row j0-1) / HR+W (row j0+W)
          double s_, n_;
          if (ORD == 3) weno3_edges_fast(lds_tile[v][r0][lane], lds_tile[v][r0 + 1][lane], lds_tile[v][r0 + 2][lane], s_, n_);
          else          weno5_edges_fast(lds_tile[v][r0][lane], lds_tile[v][r0 + 1][lane], lds_tile[v][r0 + 2][lane], lds_tile[v][r0 + 3][lane], lds_tile[v][r0 + 4][lane], s_, n_);
          if (t < 5) lds_ne[v][0][lane] = n_; else lds_set[v][lane] = s_;
        }
        MW_SCHED_FENCE();
      }
      __syncthreads();
      // own south face j: L = north edge of row j-1, R = own south edge (k_y_all: face j between cn (cell j-1) and se (cell j))
      double fS[5];
      {
        const double Lr = lds_ne[idR][wv][lane], Lu = lds_ne[idV][wv][lane], Lt = lds_ne[idT][wv][lane];
        const double cU = lds_ne[idU][wv][lane], cW = lds_ne[idW][wv][lane];
        double fn, fT;
        FaceState fs = riemann_primary<K>(p, Lr + hyr, se[idR] + hyr, Lu, se[idV], Lt, se[idT], hyt, p0, ihyt, false, fn, fT);
        const int up = fs.ind;
        fS[idR] = fs.m_upw; fS[idV] = fn; fS[idT] = fT;
        { const double sU = se[idU], sW = se[idW];
          fS[idU] = fs.m_upw * (up ? sU : cU);
          fS[idW] = fs.m_upw * (up ? sW : cW); }
#pragma unroll
        for (int l = 0; l < 5; l++) lds_fy[l][wv][lane] = fS[l];
        if (g.owns_cell) { const long long fo = (long long)kl * p.fyK + (long long)j * p.fyJ + q; MY[fo] = fs.m_upw; UPY[fo] = (unsigned char)up; }
      }
      if (wv == W - 1) {                                       // the tile's top face j0+W: L = own north edge, R = south edge of row j0+W
        double fn, fT;
        const double Rr = lds_set[idR][lane], Ru = lds_set[idV][lane], Rt = lds_set[idT][lane], sU = lds_set[idU][lane], sW = lds_set[idW][lane];
        FaceState fs = riemann_primary<K>(p, ne[idR] + hyr, Rr + hyr, ne[idV], Ru, ne[idT], Rt, hyt, p0, ihyt, false, fn, fT);
        const int up = fs.ind;
        lds_fy[idR][W][lane] = fs.m_upw; lds_fy[idV][W][lane] = fn; lds_fy[idT][W][lane] = fT;
        { const double cU = ne[idU], cW = ne[idW];
          lds_fy[idU][W][lane] = fs.m_upw * (up ? sU : cU);
          lds_fy[idW][W][lane] = fs.m_upw * (up ? sW : cW); }
        if (g.owns_cell && j + 1 == p.ny) {                    // face ny has no tile above it to store it
          const long long fo = (long long)kl * p.fyK + (long long)(j + 1) * p.fyJ + q; MY[fo] = fs.m_upw; UPY[fo] = (unsigned char)up; }
      }
      __syncthreads();
#pragma unroll
      for (int l = 0; l < 5; l++) tyo[l] = -(lds_fy[l][wv + 1][lane] - fS[l]) * p.rdy;      // (= k_y_all: -(f - fprev) * rdy for row j)
    
  };
  double tyc[5] = {0, 0, 0, 0, 0};
  if (g.kstart >= g.ka && g.kstart < g.kb) y_phase(g.kstart, tyc);   // (the bottom chunk: its first iteration is a cell of the chunk)
  for (int k = g.kstart; k <= g.kb; k++) {
    const bool top = (k == p.nz);
    const bool xwork = (k >= g.ka) && (k < g.kb);              // cells of this chunk (ghost levels only do z)
    const bool zface = (k >= g.ka);
    const bool fin = (k > g.ka);
    // ---------------- issue this iteration's global loads
    {
      const int kn = min(k + HS + 1, p.nz + p.HZ - 1);
#pragma unroll
      for (int v = 0; v < 5; v++) nxt[v] = load_zlevel<K>(p, col + (long long)v * p.sV, kn, v == idW);
    }
    double snv[5], tyv[5], immv = 0;
    const int kfc = max(k - 1, g.ka), kxc = min(max(k, g.ka), g.kb - 1);
    const int khn = min(max(k + 1, g.ka), g.kb - 1);           // the level whose y phase runs at the end of this iteration
#pragma unroll
    for (int i = 0; i < NP; i++) hal[i] = MW_HALO_LOAD(i, khn);
#pragma unroll
    for (int l = 0; l < 5; l++) { snv[l] = 0; tyv[l] = tyc[l]; }
    if (STAGE != 1) {
#pragma unroll
      for (int l = 0; l < 5; l++) snv[l] = Sn[(long long)l * p.sV + slab0 + (long long)(kfc + p.HZ) * p.sK];
    }
    if (Cf<K>::immersed(p)) immv = p.imm[cpl(p, cell0 + (long long)kfc * planeC)];
    double hpl[8];
#pragma unroll
    for (int f = 0; f < 8; f++) hpl[f] = lds_hp[(k - g.kstart) * 8 + f];
    // ------------------------------------------------ X direction (cell k = window centre)
    double fxs[5];
    int upx = 0;
    if (xwork) {
      const double hyr = hpl[0], hyt = hpl[1], p0 = hpl[2], ihyt = hpl[3];
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
      double Lv[5];
#pragma unroll
      for (int v = 0; v < 5; v++) Lv[v] = from_west<N1>(ee[v], lane, n);       // west neighbour's east-edge values
      double fn, fT;
      FaceState fs = riemann_primary<K>(p, Lv[idR] + hyr, we[idR] + hyr, Lv[idU], we[idU], Lv[idT], we[idT], hyt, p0, ihyt, false, fn, fT);
      const int up = fs.ind;
      fxs[idR] = fs.m_upw; fxs[idU] = fn; fxs[idT] = fT;
      fxs[idV] = fs.m_upw * (up ? we[idV] : Lv[idV]);
      fxs[idW] = fs.m_upw * (up ? we[idW] : Lv[idW]);
      upx = up;
    }
    // ------------------------------------------------ Z direction: reconstruct cell k, solve face k
    double be[5], te[5];
#pragma unroll
    for (int v = 0; v < 5; v++) weno_window_edges<ORD>(w[v], be[v], te[v]);
    double fzs[5];
    int upz = 0;
    {
      const double hyr = hpl[4], hyt = hpl[5], p0 = hpl[6], ihyt = hpl[7];
      double Lr = ct[idR], Lu = ct[idW], Lt = ct[idT], Rr = be[idR], Ru = be[idW], Rt = be[idT];
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
    landed(nxt); landed(snv); landed(immv); landed(hal);
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
      const double hyc = lds_hp[(kc - g.kstart) * 8];
      constexpr int wi = HS - 1;
      const double rho_s = w[idR][wi] + hyc;
      const double rho_n = (STAGE == 1) ? rho_s : snv[idR] + hyc;
      double imm_coef = 0;
      if (Cf<K>::immersed(p)) { double tau = 1.e3 * dt_stage; imm_coef = -fmin(1.0, dt_stage / tau); }
      const double ru_s = w[idU][wi] * rho_s, rv_s = w[idV][wi] * rho_s;
      double inv_rho_new = 1.0;
      double *so = Sout + slab0 + (long long)(kc + p.HZ) * p.sK;
      double xpart[5], fzprev[5];
#pragma unroll
      for (int l = 0; l < 5; l++) { xpart[l] = lds_xpart[l][threadIdx.x]; fzprev[l] = lds_fzprev[l][threadIdx.x]; }
#pragma unroll
      for (int l = 0; l < 5; l++) {
        double raw_s = w[l][wi];
        double q_s = (l == idR || l == idT) ? raw_s : raw_s * rho_s;
        double q_n;
        if (STAGE == 1) q_n = q_s;
        else q_n = (l == idR || l == idT) ? snv[l] : snv[l] * rho_n;
        double tend = xpart[l] - (fzs[l] - fzprev[l]) * lds_c[1];
        if (l == idW && Cf<K>::gravity(p)) tend += lds_c[3] * rho_s;
        if (l == idU && Cf<K>::coriolis(p)) tend += p.fcor * rv_s;
        if (l == idV && Cf<K>::coriolis(p)) tend -= p.fcor * ru_s;
        if (l == idV && Cf<K>::sim2d(p)) tend = 0;
        if (Cf<K>::immersed(p)) { double imm_tend = imm_coef * q_s / dt_stage; tend = immv * imm_tend + (1 - immv) * tend; }
        double qnew;
        const double cdt = lds_c[2];
        if (STAGE == 1)      qnew = q_n + cdt * tend;
        else if (STAGE == 2) qnew = (3.0 / 4.0) * q_n + (1.0 / 4.0) * q_s + cdt * tend;
        else                 qnew = (1.0 / 3.0) * q_n + (2.0 / 3.0) * q_s + cdt * tend;
        if (l == idR) inv_rho_new = fast_rcp(qnew + hyc);
        double stored = (l == idR || l == idT) ? qnew : qnew * inv_rho_new;
        if (MODE == 1 && l == idT) {
          const double *hq = lds_hp + (kc - g.kstart) * 8;
          stored = pressure_fast<K>(p, qnew, hq[1], hq[2], hq[3]);
        }
        if (g.owns_cell && !(MODE == 1 && (l == idU || l == idV || l == idW))) so[(long long)l * p.sV] = stored;
        if (MODE == 1 && g.owns_cell && (l == idU || l == idV || l == idW))
          (l == idU ? cu : l == idV ? cv : cw)[cpl(p, cell0 + (long long)kc * planeC)] = stored;
      }
    }
    // ------------------------------------------------ carries for the next level
    if (xwork) {
#pragma unroll
      for (int l = 0; l < 5; l++) {
        double fe = from_east<N1>(fxs[l], lane, n);
        lds_xpart[l][threadIdx.x] = -(fe - fxs[l]) * lds_c[0] + tyv[l];
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
    // ------------------------------------------------ the y phase of the NEXT level (the window is centred on it now)
    if (k + 1 >= g.ka && k + 1 < g.kb) y_phase(k + 1, tyc);       // (workgroup-uniform: all four waves march the same levels)
    (void)kxc;
  }
#undef MW_HALO_LOAD
}

} // namespace mw

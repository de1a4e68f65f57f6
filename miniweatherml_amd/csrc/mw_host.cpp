// Host-only entry points of the C ABI: error string, decomposition, constants, CFL time step.
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include <cmath>
#include <algorithm>
#include <mutex>

namespace mw {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
}

extern "C" {

const char *mw_last_error(void) { return mw::g_err.c_str(); }

int mw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// core::Coupler::distribute_mpi_and_allocate_coupled_state  (model/core/coupler.h:127-179)
int mw_decompose(int nranks, int myrank, long long nx_glob, long long ny_glob, mw_grid_t *g) {
  if (!g || nranks < 1 || myrank < 0 || myrank >= nranks || nx_glob < 1 || ny_glob < 1) MW_FAIL("mw_decompose: bad arguments");
  int nproc_x, nproc_y;
  if (ny_glob == 1) { nproc_x = nranks; nproc_y = 1; }                      // :128-131
  else {                                                                      // :132-140
    nproc_y = (int) std::ceil( std::sqrt((double) nranks) );
    while (nproc_y >= 1) { if (nranks % nproc_y == 0) break; nproc_y--; }
    nproc_x = nranks / nproc_y;
  }
  int py = myrank / nproc_x, px = myrank % nproc_x;                           // :143-144
  double nper = ((double) nx_glob)/nproc_x;                                   // :147-153
  long long i_beg = (long long) std::round( nper* px    );
  long long i_end = (long long) std::round( nper*(px+1) ) - 1;
  nper = ((double) ny_glob)/nproc_y;
  long long j_beg = (long long) std::round( nper* py    );
  long long j_end = (long long) std::round( nper*(py+1) ) - 1;
  g->nx_glob = nx_glob; g->ny_glob = ny_glob;
  g->nproc_x = nproc_x; g->nproc_y = nproc_y; g->px = px; g->py = py;
  g->i_beg = i_beg; g->j_beg = j_beg;
  g->nx = (int)(i_end - i_beg + 1); g->ny = (int)(j_end - j_beg + 1);
  for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) {                   // :169-179 periodic neighbour matrix
    int pxloc = ((px+i-1) % nproc_x + nproc_x) % nproc_x;
    int pyloc = ((py+j-1) % nproc_y + nproc_y) % nproc_y;
    g->neigh[j*3+i] = pyloc * nproc_x + pxloc;
  }
  return 0;
}

// microphysics_kessler.h:29-41 (+ :86-95 set_option) then dynamics_euler_stratified_wenofv.h:1227-1249
int mw_default_constants(mw_grid_t *g) {
  if (!g) MW_FAIL("mw_default_constants: null grid");
  g->R_d = 287.; g->cp_d = 1003.; g->R_v = 461.; g->cp_v = 1859; g->p0 = 1.e5; g->grav = 9.81;
  double cv_d = g->cp_d - g->R_d;
  g->gamma_d = g->cp_d / cv_d;
  g->kappa_d = g->R_d / g->cp_d;
  g->C0 = pow( g->R_d * pow( g->p0 , -g->kappa_d ) , g->gamma_d );          // :1247
  g->earthrot = 7.292115e-5;
  g->latitude = 0;
  return 0;
}

// halo_exchange's four face neighbours (dynamics_euler_stratified_wenofv.h:651-663) and a FIFO-safe posting order
int mw_exchange_plan(const mw_grid_t *g, int *peers, int *send_order, int *recv_order, int *active) {
  if (!g || !peers || !send_order || !recv_order || !active) MW_FAIL("mw_exchange_plan: null argument");
  peers[0] = g->neigh[1 * 3 + 0]; peers[1] = g->neigh[1 * 3 + 2]; peers[2] = g->neigh[0 * 3 + 1]; peers[3] = g->neigh[2 * 3 + 1];
  const int so[4] = {0, 1, 2, 3}, ro[4] = {1, 0, 3, 2};
  for (int i = 0; i < 4; i++) { send_order[i] = so[i]; recv_order[i] = ro[i]; }
  bool sim2d = (g->ny_glob == 1);
  active[0] = active[1] = (g->nproc_x > 1);
  active[2] = active[3] = (g->nproc_y > 1) && !sim2d;
  return 0;
}

// dynamics_euler_stratified_wenofv.h:70-77
double mw_dycore_compute_time_step(const mw_grid_t *g) {
  double dx = g->xlen / g->nx_glob, dy = g->ylen / g->ny_glob, dz = g->zlen / g->nz;   // coupler.h:262-268
  const double maxwave = 350 + 80;
  double cfl = 0.6;
  return cfl * std::min( std::min( dx , dy ) , dz ) / maxwave;
}

} // extern "C"

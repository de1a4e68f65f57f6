// =====================================================================================================
// mw_output.hip -- SURVEY.md 8(f) rank 2 and 3:
//   * the device side of the reference's file output (dynamics_euler_stratified_wenofv.h:2019-2191, time_averager.h:82-141):
//     member 0 of a (nz,ny,nx,nens) field -> contiguous (nz,ny,nx) -> host -> this rank's hyperslab of the shared CDF-5 file
//   * the two element-wise modules of the simple_city loop (experiments/simple_city/driver.cpp:72-75):
//     custom_modules::Horizontal_Sponge::{init, apply}   custom_modules/horizontal_sponge.h:18-193
//     custom_modules::Time_Averager::accumulate          custom_modules/time_averager.h:37-78
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include "mw_glibc_pow.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace mw {

// cos with the bits of the host's glibc where its main path applies (|x| < 2.4263; mw_glibc_pow.h), else the device library's
__device__ __forceinline__ double cos_glibc(double x) { double r; if (glibc_cos_main(x, &r)) return r; return cos(x); }


struct Six { double *f[6]; };

__global__ __launch_bounds__(256) void k_extract_member(const double *__restrict__ in, long long ncell, int nens, int e,
                                                        double *__restrict__ out) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < ncell) out[t] = in[t * nens + e];
}

// horizontal_sponge.h:54-61: the reference column = cell (k, 0, 0, iens) of the main rank
__global__ __launch_bounds__(256) void k_hsponge_column(Six f, int nz, long long plane, int nens, double *__restrict__ col) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 6 * nz * nens) return;
  int e = t % nens, k = (t / nens) % nz, l = t / (nens * nz);
  col[t] = f.f[l][(long long)k * plane + e];
}

// horizontal_sponge.h:133-190.  The reference runs up to four full-domain passes (x1, x2, y1, y2) one after the other; each
// touches a cell with weight 0 outside its own strip (w*col + (1-w)*v = v exactly), so applying the four relaxations in the
// same order to the value in a register is the same arithmetic in one pass.
__global__ __launch_bounds__(256) void k_hsponge_apply(Six f, int nz, int ny, int nx, int nens, int sponge_cells, double time_factor,
                                                       int x1, int x2, int y1, int y2, const double *__restrict__ col) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y, l = blockIdx.z;
  if (t >= (long long)ny * nx * nens) return;
  const int e = (int)(t % nens);
  const int i = (int)((t / nens) % nx), j = (int)(t / ((long long)nens * nx));
  const int d[4] = {i, nx - 1 - i, j, ny - 1 - j};
  const int on[4] = {x1, x2, y1, y2};
  bool any = false;
  for (int s = 0; s < 4; s++) any = any || (on[s] && d[s] < sponge_cells);
  if (!any) return;                                                          // weight 0 in every enabled pass: value unchanged
  double *q = f.f[l] + (long long)k * ny * nx * nens + t;
  const double c = col[(l * nz + k) * nens + e];
  double v = *q;
  for (int s = 0; s < 4; s++) {
    if (!on[s]) continue;
    double loc = d[s] / (sponge_cells - 1.0);
    double weight = d[s] < sponge_cells ? (cos_glibc(M_PI * loc) + 1) / 2 : 0;
    weight *= time_factor;
    v = weight * c + (1 - weight) * v;
  }
  *q = v;
}

// time_averager.h:64-75
__global__ __launch_bounds__(256) void k_time_average(Six f, Six avg, long long n, double inertia) {
#pragma clang fp contract(off)
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int l = blockIdx.y;
  if (t >= n) return;
  avg.f[l][t] = inertia * avg.f[l][t] + (1 - inertia) * f.f[l][t];
}

// custom_modules::StatisticsGatherer::is_active, gather_micro_statistics.h:61-74 (member 0 only, :42-45)
struct Eight { const double *a[8]; };
__global__ __launch_bounds__(256) void k_active_count(Eight f, long long ncell, int nens, unsigned long long *count,
                                                      unsigned char *__restrict__ mask) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  int act = 0;
  if (t < ncell) {
    const double tol = 1.e-10;
    for (int v = 0; v < 4; v++) act |= (fabs(f.a[4 + v][t * nens] - f.a[v][t * nens]) > tol) ? 1 : 0;
    if (mask) mask[t] = (unsigned char)act;
  }
  unsigned long long b = __ballot(act);
  __shared__ int sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) { int n = sm[0] + sm[1] + sm[2] + sm[3]; if (n) atomicAdd(count, (unsigned long long)n); }
}

// custom_modules::DataGenerator::generate_samples_stencil, generate_micro_surrogate_data.h:83-101: which cells to sample.
// The reference draws yakl::Random(key).genFP<double>() with key = (seed+myrank)*nz*ny*nx + k*ny*nx + j*nx + i; YAKL's generator
// is not available, so the same key goes through the splitmix64 finaliser (53 random bits -> [0,1)).
__device__ __forceinline__ double u01_from_key(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__global__ __launch_bounds__(256) void k_sample_mask(Eight f, long long ncell, int nens, unsigned long long key0, double thr_active,
                                                     double thr_inactive, unsigned char *__restrict__ mask) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ncell) return;
  const double tol = 1.e-10;
  int act = 0;
  for (int v = 0; v < 4; v++) act |= (fabs(f.a[4 + v][t * nens] - f.a[v][t * nens]) > tol) ? 1 : 0;
  const double thresh = act ? thr_active : thr_inactive;
  mask[t] = u01_from_key(key0 + (unsigned long long)t) < thresh ? 1 : 0;
}
// :139-160: one sample = inputs (5 variables x 2-cell vertical stencil) + outputs (4), fp32.  Stencil slot 1 reproduces the
// reference's assignments (:147-150): (0,1) temp(k+1), (1,1) rho_v(k+1), (2,1) rho_c(k+1), (3,1) rho_p(k+1); (4,1) is never
// assigned there (uninitialised host memory) and is written as 0 here.
__global__ __launch_bounds__(256) void k_gather_samples(const double *__restrict__ rho_d, Eight f, const long long *__restrict__ cells,
                                                        long long n, int nz, long long plane, int nens, float *__restrict__ inputs,
                                                        float *__restrict__ outputs) {
  long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const long long c = cells[s];
  const int k = (int)(c / plane);
  const long long up = (k + 1 < nz ? c + plane : c);            // min(nz-1, k+1)
  float *in = inputs + s * 10, *out = outputs + s * 4;
  in[0] = (float)f.a[0][c * nens];  in[2] = (float)rho_d[c * nens];  in[4] = (float)f.a[1][c * nens];
  in[6] = (float)f.a[2][c * nens];  in[8] = (float)f.a[3][c * nens];
  in[1] = (float)f.a[0][up * nens]; in[3] = (float)f.a[1][up * nens]; in[5] = (float)f.a[2][up * nens];
  in[7] = (float)f.a[3][up * nens]; in[9] = 0.0f;
  for (int v = 0; v < 4; v++) out[v] = (float)f.a[4 + v][c * nens];
}

// mean(a - b) over n doubles, deterministic: 1024 workgroups each sum a fixed, interleaved share in a fixed order; one workgroup
// then adds the 1024 partial sums as a fixed tree (the "Relative diff" prints of the surrogate module, microphysics_kessler_ponni.h:266-269,
// are yakl::intrinsics::sum(a - b) / size in the reference).
__global__ __launch_bounds__(256) void k_diff_partial(long long n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ part) {
#pragma clang fp contract(off)
  double s = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i] - b[i];
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void k_diff_final(double *__restrict__ part, int np) {
#pragma clang fp contract(off)
  __shared__ double sm[256];
  double s = 0;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) part[0] = sm[0];
}

} // namespace mw

using namespace mw;

static int six_ptrs(double *const *p, Six &s, const char *what) {
  if (!p) { set_error(std::string(what) + ": null field list"); return 1; }
  for (int l = 0; l < 6; l++) { if (!p[l]) { set_error(std::string(what) + ": null field"); return 1; } s.f[l] = p[l]; }
  return 0;
}

namespace mw { int nc_staging(mw_nc_t nc, size_t bytes, double **dev, double **host); }   // mw_netcdf.cpp

// DataManager::validate_single_nan / _inf / _pos (model/core/DataManager.h:446-483) as ONE pass over the device array: the reference
// copies the array to the host and loops.  out[0..2] = how many NaN / inf / negative elements, out[3..5] = the lowest flat index of
// each kind (the first one the reference's loop would report; ~0 = none).  grid-stride, wave reduction, one atomic per wave and kind.
template <class T>
__global__ __launch_bounds__(256) void k_validate(const T *__restrict__ a, long long n, unsigned long long *__restrict__ out) {
  unsigned long long cnt[3] = {0, 0, 0}, first[3] = {~0ull, ~0ull, ~0ull};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const T v = a[i];
    const bool hit[3] = {v != v, (v - v != (T)0) && (v == v), v < (T)0};           // NaN; inf (finite - finite = 0); negative
#pragma unroll
    for (int c = 0; c < 3; c++) if (hit[c]) { cnt[c]++; if ((unsigned long long)i < first[c]) first[c] = (unsigned long long)i; }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    for (int off = 32; off > 0; off >>= 1) {
      cnt[c] += __shfl_down(cnt[c], off, 64);
      const unsigned long long o = __shfl_down(first[c], off, 64);
      if (o < first[c]) first[c] = o;
    }
    if ((threadIdx.x & 63) == 0 && cnt[c]) { atomicAdd(&out[c], cnt[c]); atomicMin(&out[3 + c], first[c]); }
  }
}

extern "C" {

int mw_output_put_field(mw_nc_t nc, int varid, long long record, const mw_grid_t *g, const double *field, void *stream) {
  if (!nc || !g || !field) MW_FAIL("output_put_field: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  const long long ncell = (long long)g->nz * g->ny * g->nx;
  double *dev = nullptr, *host = nullptr;
  if (nc_staging(nc, (size_t)ncell * sizeof(double), &dev, &host)) return 1;      // cached in the file handle, freed by mw_nc_close
  int rc = 0;
  hipLaunchKernelGGL(k_extract_member, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, field, ncell, g->nens, 0, dev);   // iens = 0 (:2035)
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(host, dev, (size_t)ncell * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { set_error("output_put_field: device to host copy failed"); rc = 1; }
  if (!rc) {
    if (record >= 0) { const long long start[4] = {record, 0, g->j_beg, g->i_beg}, count[4] = {1, g->nz, g->ny, g->nx};
                       rc = mw_nc_put_vara_double(nc, varid, start, count, host); }
    else             { const long long start[3] = {0, g->j_beg, g->i_beg}, count[3] = {g->nz, g->ny, g->nx};
                       rc = mw_nc_put_vara_double(nc, varid, start, count, host); }
  }
  return rc;
}

int mw_horizontal_sponge_column(const mw_grid_t *g, const double *const *fields6, double *column, void *stream) {
  if (!g || !column) MW_FAIL("horizontal_sponge_column: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  Six f; if (six_ptrs((double *const *)fields6, f, "horizontal_sponge_column")) return 1;
  const int n = 6 * g->nz * g->nens;
  hipLaunchKernelGGL(k_hsponge_column, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, g->nz,
                     (long long)g->ny * g->nx * g->nens, g->nens, column);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_horizontal_sponge_apply(const mw_grid_t *g, double *const *fields6, const double *column, int sponge_cells, double time_scale,
                               double dt, int x1, int x2, int y1, int y2, void *stream) {
  if (!g || !column) MW_FAIL("horizontal_sponge_apply: null argument");
  if (sponge_cells < 2) MW_FAIL("horizontal_sponge_apply: sponge_cells must be >= 2 (the reference divides by sponge_cells - 1)");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  Six f; if (six_ptrs(fields6, f, "horizontal_sponge_apply")) return 1;
  // a pass only runs on the ranks that own that domain edge (horizontal_sponge.h:133,148,163,178)
  x1 = x1 && g->px == 0; x2 = x2 && g->px == g->nproc_x - 1; y1 = y1 && g->py == 0; y2 = y2 && g->py == g->nproc_y - 1;
  if (!(x1 || x2 || y1 || y2)) return 0;
  const long long plane = (long long)g->ny * g->nx * g->nens;
  dim3 grid((unsigned)((plane + 255) / 256), (unsigned)g->nz, 6u);
  hipLaunchKernelGGL(k_hsponge_apply, grid, dim3(256), 0, (hipStream_t)stream, f, g->nz, g->ny, g->nx, g->nens, sponge_cells, dt / time_scale,
                     x1, x2, y1, y2, column);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_micro_active_count(const mw_grid_t *g, const double *const *in4, const double *const *out4, unsigned char *mask,
                          long long *count, void *stream) {
  if (!g || !in4 || !out4 || !count) MW_FAIL("micro_active_count: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  Eight f;
  for (int v = 0; v < 4; v++) { if (!in4[v] || !out4[v]) MW_FAIL("micro_active_count: null field"); f.a[v] = in4[v]; f.a[4 + v] = out4[v]; }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long *dev = nullptr, host = 0;
  MW_HIP(hipMalloc(&dev, sizeof(unsigned long long)));
  int rc = 0;
  const long long ncell = (long long)g->nz * g->ny * g->nx;
  if (hipMemsetAsync(dev, 0, sizeof(unsigned long long), st) != hipSuccess) rc = 1;
  if (!rc) { hipLaunchKernelGGL(k_active_count, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, f, ncell, g->nens, dev, mask);
             if (hipGetLastError() != hipSuccess) rc = 1; }
  if (!rc && (hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = 1;
  (void)hipFree(dev);
  if (rc) MW_FAIL("micro_active_count: device operation failed");
  *count = (long long)host;
  return 0;
}

int mw_micro_sample_mask(const mw_grid_t *g, const double *const *in4, const double *const *out4, unsigned long long key0,
                         double thr_active, double thr_inactive, unsigned char *mask, void *stream) {
  if (!g || !in4 || !out4 || !mask) MW_FAIL("micro_sample_mask: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  Eight f;
  for (int v = 0; v < 4; v++) { if (!in4[v] || !out4[v]) MW_FAIL("micro_sample_mask: null field"); f.a[v] = in4[v]; f.a[4 + v] = out4[v]; }
  const long long ncell = (long long)g->nz * g->ny * g->nx;
  hipLaunchKernelGGL(k_sample_mask, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, ncell, g->nens, key0,
                     thr_active, thr_inactive, mask);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_micro_gather_samples(const mw_grid_t *g, const double *rho_d, const double *const *in4, const double *const *out4,
                            const long long *cells, long long n, float *inputs, float *outputs, void *stream) {
  if (!g || !rho_d || !in4 || !out4 || (n > 0 && (!cells || !inputs || !outputs))) MW_FAIL("micro_gather_samples: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  if (n <= 0) return 0;
  Eight f;
  for (int v = 0; v < 4; v++) { if (!in4[v] || !out4[v]) MW_FAIL("micro_gather_samples: null field"); f.a[v] = in4[v]; f.a[4 + v] = out4[v]; }
  hipLaunchKernelGGL(k_gather_samples, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rho_d, f, cells, n, g->nz,
                     (long long)g->ny * g->nx, g->nens, inputs, outputs);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_mean_diff(long long n, const double *a, const double *b, double *workspace1024, double *mean_out, void *stream) {
  if (n < 1 || !a || !b || !workspace1024 || !mean_out) MW_FAIL("mean_diff: bad argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::min<long long>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(k_diff_partial, dim3((unsigned)nb), dim3(256), 0, st, n, a, b, workspace1024); MW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_diff_final, dim3(1), dim3(256), 0, st, workspace1024, nb); MW_LAUNCH_CHECK();
  double sum = 0;
  MW_HIP(hipMemcpyAsync(&sum, workspace1024, sizeof(double), hipMemcpyDeviceToHost, st));
  MW_HIP(hipStreamSynchronize(st));
  *mean_out = sum / (double)n;
  return 0;
}

static int validate_any(const void *field, long long n, int is_f32, long long *out6, void *stream) {
  if (!field || n < 1 || !out6) MW_FAIL("mw_validate: bad argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long *dev = nullptr, host[6] = {0, 0, 0, ~0ull, ~0ull, ~0ull};
  MW_HIP(hipMalloc(&dev, sizeof(host)));
  int rc = 0;
  if (hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, st) != hipSuccess) rc = 1;
  const unsigned nb = (unsigned)std::min<long long>(4096, (n + 255) / 256);
  if (!rc) {
    if (is_f32) hipLaunchKernelGGL(k_validate<float>, dim3(nb), dim3(256), 0, st, (const float *)field, n, dev);
    else        hipLaunchKernelGGL(k_validate<double>, dim3(nb), dim3(256), 0, st, (const double *)field, n, dev);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc && (hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = 1;
  (void)hipFree(dev);
  if (rc) MW_FAIL("mw_validate: device operation failed");
  for (int c = 0; c < 3; c++) { out6[c] = (long long)host[c]; out6[3 + c] = host[c] ? (long long)host[3 + c] : -1; }
  return 0;
}
int mw_validate_f64(const double *field, long long n, long long *out6, void *stream) { return validate_any(field, n, 0, out6, stream); }
int mw_validate_f32(const float *field, long long n, long long *out6, void *stream) { return validate_any(field, n, 1, out6, stream); }

int mw_time_average_accumulate(const mw_grid_t *g, const double *const *fields6, double *const *avg6, double etime, double dt, void *stream) {
  if (!g) MW_FAIL("time_average_accumulate: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  Six f, a;
  if (six_ptrs((double *const *)fields6, f, "time_average_accumulate") || six_ptrs(avg6, a, "time_average_accumulate")) return 1;
  const long long n = (long long)g->nz * g->ny * g->nx * g->nens;
  const double inertia = etime / (etime + dt);                               // time_averager.h:62
  hipLaunchKernelGGL(k_time_average, dim3((unsigned)((n + 255) / 256), 6u), dim3(256), 0, (hipStream_t)stream, f, a, n, inertia);
  MW_LAUNCH_CHECK();
  return 0;
}

} // extern "C"

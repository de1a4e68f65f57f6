// =====================================================================================================
// mw_column.hip -- the two remaining per-step modules of the supercell loop:
//   modules::sponge_layer(coupler, dt, time_scale)      model/modules/sponge_layer.h:8-77
//   modules::ColumnNudger::{set_column, nudge_to_column} model/modules/column_nudging.h:15-106
// Both need horizontal sums per (field, level, ensemble member).  The reference accumulates them with atomicAdd
// (order undefined on a GPU, sponge_layer.h:50, column_nudging.h:86); here they are DETERMINISTIC: fixed slices of a level
// are tree-reduced per workgroup, the slice partials are added in index order, ranks are combined by the caller's
// all-reduce (MPI_Allreduce in the reference, :53-63 / :89-99; ncclAllReduce / torch.distributed natively).
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include "mw_glibc_pow.h"
#include <cmath>

namespace mw {

// cos with the bits of the host's glibc (mw_glibc_pow.h; the device library's beyond |x| = 1e8)
__device__ __forceinline__ double cos_glibc(double x) { double r; if (glibc_cos_main(x, &r)) return r; return cos(x); }


struct FieldPtrs { double *f[5 + MW_MAX_TRACERS]; };

static constexpr int SLICE = 16384;       // cells per partial sum

// partial[((fld*nlev + lev)*nens + e)*S + s] = sum over slice s of level `lev0 + dir*lev` of field fld, member e
__global__ __launch_bounds__(256) void k_hsum_partial(FieldPtrs fp, int nlev, int lev0, int dir, long long ncell_lev, int nens, int S,
                                                      int skip_field, double *__restrict__ partial) {
  const int s = blockIdx.x, lev = blockIdx.y, fe = blockIdx.z;
  const int fld = fe / nens, e = fe - fld * nens;
  double acc = 0;
  if (fld != skip_field) {
    const double *src = fp.f[fld] + (long long)(lev0 + dir * lev) * ncell_lev * nens + e;
    const long long c0 = (long long)s * SLICE, c1 = min(c0 + SLICE, ncell_lev);
    // four independent chains per thread (four loads in flight instead of one: the kernel is a pure stream, 3.5 -> 4.7 TB/s); a fixed
    // association, so the sums stay reproducible run to run and layout to layout
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    long long c = c0 + threadIdx.x;
    for (; c + 768 < c1; c += 1024) { a0 += src[c * nens]; a1 += src[(c + 256) * nens]; a2 += src[(c + 512) * nens]; a3 += src[(c + 768) * nens]; }
    for (; c < c1; c += 256) a0 += src[c * nens];
    acc = (a0 + a1) + (a2 + a3);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[((long long)(fld * nlev + lev) * nens + e) * S + s] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// STRICT form (mw_column_set_strict(1)): one thread per (field, level, member) adds the level's cells in the reference's serial
// order (j, then i; sponge_layer.h:44-51, column_nudging.h:80-87 on the serial backend) -- the sums, and with them the two modules,
// are then bit-identical to the CPU restatement.  Not a performance path (ny * nx dependent additions per thread).
__global__ __launch_bounds__(64) void k_hsum_serial(FieldPtrs fp, int nlev, int lev0, int dir, long long ncell_lev, int nens, int nf,
                                                    int skip_field, double *__restrict__ out) {
#pragma clang fp contract(off)
  const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
  if (t >= (long long)nf * nlev * nens) return;
  const int e = (int)(t % nens);
  const int lev = (int)((t / nens) % nlev);
  const int fld = (int)(t / ((long long)nens * nlev));
  double acc = 0;
  if (fld != skip_field) {
    const double *src = fp.f[fld] + (long long)(lev0 + dir * lev) * ncell_lev * nens + e;
    for (long long c = 0; c < ncell_lev; c++) acc += src[c * nens];
  }
  out[t] = acc;
}
__global__ __launch_bounds__(256) void k_hsum_finish(const double *__restrict__ partial, long long n, int S, double *__restrict__ out) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  double a = 0;
  for (int s = 0; s < S; s++) a += partial[t * S + s];
  out[t] = a;
}

// sponge_layer.h:66-76
static constexpr int SPONGE_NPT = 4;
__global__ __launch_bounds__(256) void k_sponge_apply(FieldPtrs fp, int num_fields, int num_layers, int nz, long long ncell_lev, int nens,
                                                      double zlen, double dz, double time_factor, double nglob,
                                                      const double *__restrict__ havg) {
#pragma clang fp contract(off)
  const int kloc = blockIdx.y, ifld = blockIdx.z;
  const int k = nz - 1 - kloc;
  // the relaxation factor depends on the level alone and the target on (field, level, member): the reference's expressions, evaluated by
  // the block's first threads once instead of by every cell (a cosine and a division per cell: the pass took twice its bytes' time)
  __shared__ double sh_factor, sh_target[64];
  if (threadIdx.x == 0) {
    double z = (k + 0.5) * dz;
    double rel_dist = (zlen - z) / (num_layers * dz);
    double space_factor = (cos_glibc(M_PI * rel_dist) + 1) / 2;
    sh_factor = space_factor * time_factor;
  }
  if ((int)threadIdx.x < min(nens, 64)) sh_target[threadIdx.x] = havg[(ifld * num_layers + kloc) * nens + threadIdx.x] / nglob;
  __syncthreads();
  // four cells per thread, 256 apart, their loads issued together (one cell per thread kept one load in flight: 2 TB/s)
  const double factor = sh_factor;
  const long long n = ncell_lev * nens, t0 = (long long)blockIdx.x * (256 * SPONGE_NPT) + threadIdx.x;
  double *base = fp.f[ifld] + (long long)k * n;
  double v[SPONGE_NPT];
#pragma unroll
  for (int i = 0; i < SPONGE_NPT; i++) { const long long ti = t0 + (long long)i * 256; v[i] = (ti < n) ? base[ti] : 0.0; }
#pragma unroll
  for (int i = 0; i < SPONGE_NPT; i++) {
    const long long ti = t0 + (long long)i * 256;
    if (ti >= n) continue;
    const int e = (int)(ti % nens);
    const double target = (e < 64) ? sh_target[e] : havg[(ifld * num_layers + kloc) * nens + e] / nglob;
    base[ti] = v[i] + (target - v[i]) * factor;
  }
}

__global__ __launch_bounds__(256) void k_div_scalar(double *__restrict__ a, long long n, double d) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) a[t] = a[t] / d;
}

// column_nudging.h:62-65.  The increment dt (column - avg) / time_scale is one number per (field, level, member): a thread computes it
// once (the reference's expression, IEEE division) and adds it to NPT cells `nens * 256` apart -- rounds 1-4 ran one thread, and one
// division, per cell (0.45 ms per call on config 2, twice the time its 1.9 GB take).  Fields and levels are walked in REVERSE order:
// the sums in front of this pass read the arrays field 0 .. 4, level 0 .. nz - 1, so the last 256 MB they touched -- what the
// Infinity Cache still holds -- are the first this pass asks for.
static constexpr int NUDGE_NPT = 8;
__global__ __launch_bounds__(256) void k_nudge_apply(FieldPtrs fp, int nz, long long ncell_lev, int nens, double dt,
                                                     const double *__restrict__ column, const double *__restrict__ avg) {
#pragma clang fp contract(off)
  const int k = nz - 1 - (int)blockIdx.y, l = 4 - (int)blockIdx.z;
  const long long n = ncell_lev * nens;
  const long long stride = 256ll * nens;                       // (a multiple of nens: every cell of a thread belongs to one member)
  const long long t0 = (long long)blockIdx.x * (stride * NUDGE_NPT) + (long long)threadIdx.x * nens;
  const double time_scale = 900;
  double *q = fp.f[l] + (long long)k * n;
  for (int e = 0; e < nens; e++) {
    const long long m = ((long long)l * nz + k) * nens + e;
    const double inc = dt * (column[m] - avg[m]) / time_scale;
    double v[NUDGE_NPT];
#pragma unroll
    for (int c = 0; c < NUDGE_NPT; c++) { const long long t = t0 + c * stride + e; v[c] = (t < n) ? q[t] : 0.0; }
#pragma unroll
    for (int c = 0; c < NUDGE_NPT; c++) { const long long t = t0 + c * stride + e; if (t < n) q[t] = v[c] + inc; }
  }
}

} // namespace mw

using namespace mw;

static thread_local int g_column_strict = 0;        // per calling thread: a rank harness with one host thread per rank may use different modes side by side
// 1: the horizontal sums of sponge_layer / ColumnNudger in the reference's serial order (bit-identical to the serial backend);
// 0 (default): fixed-slice tree sums.  Process-wide.
extern "C" int mw_column_set_strict(int strict) { g_column_strict = strict ? 1 : 0; return 0; }

// horizontal sums of `nf` fields over levels lev0, lev0+dir, ... (nlev of them) -> out (nf, nlev, nens), all ranks combined
static int hsum(const mw_grid_t *g, const FieldPtrs &fp, int nf, int nlev, int lev0, int dir, int skip_field, double *out, double *partial,
                mw_allreduce_fn ar, void *ctx, hipStream_t st) {
  const long long ncell_lev = (long long)g->ny * g->nx;
  const int S = (int)((ncell_lev + SLICE - 1) / SLICE);
  const long long n = (long long)nf * nlev * g->nens;
  if (g_column_strict) {
    hipLaunchKernelGGL(k_hsum_serial, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, fp, nlev, lev0, dir, ncell_lev, g->nens, nf, skip_field, out);
    MW_LAUNCH_CHECK();
  } else {
    dim3 grid((unsigned)S, (unsigned)nlev, (unsigned)(nf * g->nens));
    hipLaunchKernelGGL(k_hsum_partial, grid, dim3(256), 0, st, fp, nlev, lev0, dir, ncell_lev, g->nens, S, skip_field, partial);
    MW_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hsum_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, n, S, out);
    MW_LAUNCH_CHECK();
  }
  if (ar && g->nproc_x * g->nproc_y > 1) { if (ar(ctx, out, n, st)) MW_FAIL("all-reduce callback failed"); }
  return 0;
}

extern "C" {

long long mw_column_workspace_bytes(const mw_grid_t *g, int num_fields) {
  if (!g || num_fields < 1) return -1;
  const long long ncell_lev = (long long)g->ny * g->nx;
  const long long S = (ncell_lev + SLICE - 1) / SLICE;
  const long long nlev = g->nz;                       // the nudger sums every level; the sponge only 10
  return (long long)sizeof(double) * ((long long)num_fields * nlev * g->nens * (S + 2) + 64);
}

int mw_sponge_layer(const mw_grid_t *g, double *const *fields, int num_fields, double dt, double time_scale, void *workspace,
                    mw_allreduce_fn allreduce, void *ctx, void *stream) {
  if (!g || !fields || !workspace) MW_FAIL("sponge_layer: null argument");
  if (num_fields < 5 || num_fields > 5 + MW_MAX_TRACERS) MW_FAIL("sponge_layer: num_fields out of range");
  const int num_layers = 10, WFLD = 3;                                       // sponge_layer.h:19-21
  if (g->nz < num_layers) MW_FAIL("sponge_layer: needs nz >= 10");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  FieldPtrs fp;
  for (int f = 0; f < num_fields; f++) { if (!fields[f]) MW_FAIL("sponge_layer: null field"); fp.f[f] = fields[f]; }
  double *havg = (double *)workspace;
  double *partial = havg + (long long)num_fields * num_layers * g->nens;
  if (hsum(g, fp, num_fields, num_layers, g->nz - 1, -1, WFLD, havg, partial, allreduce, ctx, st)) return 1;
  const long long ncell_lev = (long long)g->ny * g->nx;
  dim3 grid((unsigned)((ncell_lev * g->nens + 256 * SPONGE_NPT - 1) / (256 * SPONGE_NPT)), (unsigned)num_layers, (unsigned)num_fields);
  const double dz = g->zlen / g->nz;
  const double nglob = (double)((unsigned long long)g->nx_glob * (unsigned long long)g->ny_glob);      // size_t product (:75)
  hipLaunchKernelGGL(k_sponge_apply, grid, dim3(256), 0, st, fp, num_fields, num_layers, g->nz, ncell_lev, g->nens, g->zlen, dz,
                     dt / time_scale, nglob, havg);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_column_average(const mw_grid_t *g, const double *const *state5, double *column_out, void *workspace, mw_allreduce_fn allreduce,
                      void *ctx, void *stream) {
  if (!g || !state5 || !column_out || !workspace) MW_FAIL("column_average: null argument");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  hipStream_t st = (hipStream_t)stream;
  FieldPtrs fp;
  for (int f = 0; f < 5; f++) { if (!state5[f]) MW_FAIL("column_average: null field"); fp.f[f] = (double *)state5[f]; }
  if (hsum(g, fp, 5, g->nz, 0, +1, -1, column_out, (double *)workspace, allreduce, ctx, st)) return 1;
  const long long n = 5ll * g->nz * g->nens;
  const double nglob = (double)((int)g->nx_glob * (int)g->ny_glob);          // int product (column_nudging.h:73-74,103)
  hipLaunchKernelGGL(k_div_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, column_out, n, nglob);
  MW_LAUNCH_CHECK();
  return 0;
}

int mw_nudge_to_column(const mw_grid_t *g, double *const *state5, const double *column, double dt, void *workspace,
                       mw_allreduce_fn allreduce, void *ctx, void *stream) {
  if (!g || !state5 || !column || !workspace) MW_FAIL("nudge_to_column: null argument");
  hipStream_t st = (hipStream_t)stream;
  double *avg = (double *)workspace;
  double *rest = avg + 5ll * g->nz * g->nens;
  if (mw_column_average(g, state5, avg, rest, allreduce, ctx, stream)) return 1;
  FieldPtrs fp;
  for (int f = 0; f < 5; f++) fp.f[f] = state5[f];
  const long long ncell_lev = (long long)g->ny * g->nx;
  dim3 grid((unsigned)((ncell_lev + 256ll * NUDGE_NPT - 1) / (256ll * NUDGE_NPT)), (unsigned)g->nz, 5u);
  hipLaunchKernelGGL(k_nudge_apply, grid, dim3(256), 0, st, fp, g->nz, ncell_lev, g->nens, dt, column, avg);
  MW_LAUNCH_CHECK();
  return 0;
}

} // extern "C"

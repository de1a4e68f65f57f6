// =====================================================================================================
// mw_mlp.hip -- the ponni 5 -> 10 -> 4 surrogate MLP as a batched MFMA GEMM on gfx950
// reference call sites: experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h
//   :103-110 (Matvec, Bias, Relu(negative_slope 0.1), Matvec, Bias), :180-187 (input scaling -> float),
//   :189 (forward_batch_parallel), :196-201 (un-scaling, clip >= 0).
//
// One fused kernel: scale (fp64) -> layer 1 -> leaky ReLU -> layer 2 -> un-scale + clip (fp64), 72 B of HBM
// traffic per cell and nothing else.  Matrix work rides v_mfma_f32_16x16x4_f32 (exact f32 fma chain):
//   tile = 16 cells on the N (column = lane & 15) axis; the K axis (lane >> 4 = "group" g) carries input features.
//   Layer 1: D1[16 x 16cells] = W1p^T[16 x 8] * X[8 x 16cells] + b1p      (2 MFMAs, K = 5 padded to 8)
//   Layer 2: D2[16 x 16cells] = W2p^T[16 x 12] * H[12 x 16cells] + b2p    (3 MFMAs, K = 10 padded to 12)
// The C/D layout (row = 4*g + reg) is used as the next B operand WITHOUT any lane movement: hidden unit u is
// placed at row rho(u) with rho(u) % 4 < 3, so register j of group g *is* k-slot g of MFMA j.  Output n is
// placed at row 4n, so lane group n holds output n in register 0 and stores 16 contiguous doubles.
// Inputs are fetched the same way: lane group g reads feature g's array (128 contiguous bytes per group).
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>

namespace mw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpP {
  float a1[2][64];      // layer-1 A operand per MFMA m, per lane
  float c1[4][4];       // layer-1 C init (bias) [g][reg]
  float a2[3][64];      // layer-2 A operand per MFMA j, per lane
  float c2[4];          // layer-2 C init for reg 0 of group g (bias of output g)
  double in_min[5], in_rng[5];     // scl_in(:,0), scl_in(:,1)-scl_in(:,0)
  double out_min[4], out_rng[4];
};

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.1f * x; }

// TILES 16-cell tiles per wave per iteration
template <int TILES>
__global__ __launch_bounds__(256) void k_mlp(MlpP P, long long ncells, const double *__restrict__ temp,
                                             const double *__restrict__ rho_d, const double *__restrict__ rho_v,
                                             const double *__restrict__ rho_c, const double *__restrict__ rho_r,
                                             double *__restrict__ o_temp, double *__restrict__ o_rv,
                                             double *__restrict__ o_rc, double *__restrict__ o_rr) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, cidx = lane & 15;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * 256) >> 6;
  // per-lane constants
  const double *in_g = (g == 0) ? temp : (g == 1) ? rho_d : (g == 2) ? rho_v : rho_c;
  double *out_g = (g == 0) ? o_temp : (g == 1) ? o_rv : (g == 2) ? o_rc : o_rr;
  // min-max scaling as a multiply by the reciprocal range (the quotient is cast to fp32 right after: the <= 1 ulp fp64
  // difference is invisible at 24 bits except on exact rounding ties)
  const double imin = P.in_min[g], irng = 1.0 / P.in_rng[g], imin4 = P.in_min[4], irng4 = 1.0 / P.in_rng[4];
  const double omin = P.out_min[g], orng = P.out_rng[g];
  const float a10 = P.a1[0][lane], a11 = P.a1[1][lane];
  const float a20 = P.a2[0][lane], a21 = P.a2[1][lane], a22 = P.a2[2][lane];
  const f32x4 c1 = {P.c1[g][0], P.c1[g][1], P.c1[g][2], P.c1[g][3]};
  const f32x4 c2 = {P.c2[g], 0.f, 0.f, 0.f};
  const long long ntiles = (ncells + 15) / 16;
  for (long long t0 = wave * TILES; t0 < ntiles; t0 += nwaves * TILES) {
    double xin[TILES], xin4[TILES];
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      long long cell = (t0 + u) * 16 + cidx;
      bool ok = cell < ncells;
      xin[u]  = ok ? in_g[cell] : imin;
      xin4[u] = (ok && g == 0) ? rho_r[cell] : imin4;
    }
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      long long cell = (t0 + u) * 16 + cidx;
      float b0 = (float)((xin[u] - imin) * irng);                         // :182-186 (fp64 math, stored as float)
      float b1 = (g == 0) ? (float)((xin4[u] - imin4) * irng4) : 0.f;
      f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10, b0, c1, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, b1, d1, 0, 0, 0);
      float h0 = leaky(d1[0]), h1 = leaky(d1[1]), h2 = leaky(d1[2]);      // Relu(negative_slope = 0.1), :105
      f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, h0, c2, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, h1, d2, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a22, h2, d2, 0, 0, 0);
      double y = (double)d2[0] * orng + omin;                             // :198-201
      if (g != 0) y = fmax(0.0, y);
      if (cell < ncells) out_g[cell] = y;
    }
  }
}

// The same with 16-byte accesses: a lane owns the cell PAIR (2c, 2c+1) of a 32-cell span, loaded / stored as one double2 -- every
// lane group then moves 256 contiguous bytes per instruction and the wave issues half as many memory instructions.  The even
// cells of the span form one MFMA tile, the odd cells the next (a tile is any 16 cells).  ncells must be a multiple of 32 here;
// the launcher hands the remainder to k_mlp.
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <int PAIRS>
__global__ __launch_bounds__(256) void k_mlp_x2(MlpP P, long long ncells, const double *__restrict__ temp,
                                                const double *__restrict__ rho_d, const double *__restrict__ rho_v,
                                                const double *__restrict__ rho_c, const double *__restrict__ rho_r,
                                                double *__restrict__ o_temp, double *__restrict__ o_rv,
                                                double *__restrict__ o_rc, double *__restrict__ o_rr) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, cidx = lane & 15;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * 256) >> 6;
  const double *in_g = (g == 0) ? temp : (g == 1) ? rho_d : (g == 2) ? rho_v : rho_c;
  double *out_g = (g == 0) ? o_temp : (g == 1) ? o_rv : (g == 2) ? o_rc : o_rr;
  const double imin = P.in_min[g], irng = 1.0 / P.in_rng[g], imin4 = P.in_min[4], irng4 = 1.0 / P.in_rng[4];
  const double omin = P.out_min[g], orng = P.out_rng[g];
  const float a10 = P.a1[0][lane], a11 = P.a1[1][lane];
  const float a20 = P.a2[0][lane], a21 = P.a2[1][lane], a22 = P.a2[2][lane];
  const f32x4 c1 = {P.c1[g][0], P.c1[g][1], P.c1[g][2], P.c1[g][3]};
  const f32x4 c2 = {P.c2[g], 0.f, 0.f, 0.f};
  const long long nspans = ncells / 32;
  for (long long s0 = wave * PAIRS; s0 < nspans; s0 += nwaves * PAIRS) {
    f64x2 xin[PAIRS], xin4[PAIRS];
#pragma unroll
    for (int u = 0; u < PAIRS; u++) {
      const long long cell = min(s0 + u, nspans - 1) * 32 + 2 * cidx;          // (clamped: the tail spans are recomputed, not stored)
      xin[u] = *(const f64x2 *)(in_g + cell);
      xin4[u] = (g == 0) ? *(const f64x2 *)(rho_r + cell) : (f64x2){imin4, imin4};
    }
#pragma unroll
    for (int u = 0; u < PAIRS; u++) {
      f64x2 y2;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float b0 = (float)((xin[u][h] - imin) * irng);                    // :182-186 (fp64 math, stored as float)
        float b1 = (g == 0) ? (float)((xin4[u][h] - imin4) * irng4) : 0.f;
        f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10, b0, c1, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, b1, d1, 0, 0, 0);
        float h0 = leaky(d1[0]), h1 = leaky(d1[1]), h2 = leaky(d1[2]);    // Relu(negative_slope = 0.1), :105
        f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, h0, c2, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, h1, d2, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a22, h2, d2, 0, 0, 0);
        double y = (double)d2[0] * orng + omin;                           // :198-201
        if (g != 0) y = fmax(0.0, y);
        y2[h] = y;
      }
      if (s0 + u < nspans) *(f64x2 *)(out_g + (s0 + u) * 32 + 2 * cidx) = y2;
    }
  }
}


// ponni::Inference::forward_batch_parallel for the surrogate's stack on fp32 arrays in ponni's own layout -- in (5, batch), out (4, batch),
// batch fastest (microphysics_kessler_ponni.h:176-189: ponni_in(feature, iglob)) -- with the same MFMA tiles as k_mlp: lane group g
// reads feature g's row (64 contiguous bytes per group and tile), output n leaves from group n.  The activation slope is an argument
// (ponni::Relu<float>(n, negative_slope), :105).
template <int TILES>
__global__ __launch_bounds__(256) void k_mlp_f32(MlpP P, float slope, long long batch, const float *__restrict__ in, float *__restrict__ out) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, cidx = lane & 15;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * 256) >> 6;
  const float *in_g = in + (long long)g * batch, *in_4 = in + 4 * batch;
  float *out_g = out + (long long)g * batch;
  const float a10 = P.a1[0][lane], a11 = P.a1[1][lane];
  const float a20 = P.a2[0][lane], a21 = P.a2[1][lane], a22 = P.a2[2][lane];
  const f32x4 c1 = {P.c1[g][0], P.c1[g][1], P.c1[g][2], P.c1[g][3]};
  const f32x4 c2 = {P.c2[g], 0.f, 0.f, 0.f};
  const long long ntiles = (batch + 15) / 16;
  for (long long t0 = wave * TILES; t0 < ntiles; t0 += nwaves * TILES) {
    float x0[TILES], x4[TILES];
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      const long long cell = (t0 + u) * 16 + cidx;
      const bool ok = cell < batch;
      x0[u] = ok ? in_g[cell] : 0.f;
      x4[u] = (ok && g == 0) ? in_4[cell] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < TILES; u++) {
      const long long cell = (t0 + u) * 16 + cidx;
      f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10, x0[u], c1, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, x4[u], d1, 0, 0, 0);
      const float h0 = d1[0] > 0.f ? d1[0] : slope * d1[0], h1 = d1[1] > 0.f ? d1[1] : slope * d1[1], h2 = d1[2] > 0.f ? d1[2] : slope * d1[2];
      f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, h0, c2, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, h1, d2, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a22, h2, d2, 0, 0, 0);
      if (cell < batch) out_g[cell] = d2[0];
    }
  }
}

// Any other stack of ponni layers (Matvec / Bias / Relu), and the strict form of the surrogate's: thread = one batch element, every
// layer a plain fp32 loop in index order (Matvec: acc = 0; acc += x[i] * W[i][o] for i = 0 .. n_in - 1), no contraction.
#define MW_PONNI_MAX_LAYERS 10                                  // (microphysics_kessler_ponni.h:32)
#define MW_PONNI_MAX_WIDTH 32
#define MW_PONNI_MAX_PARAMS 960
struct PonniStack { int nlayers; int kind[MW_PONNI_MAX_LAYERS], n_in[MW_PONNI_MAX_LAYERS], n_out[MW_PONNI_MAX_LAYERS], off[MW_PONNI_MAX_LAYERS];
                    float slope[MW_PONNI_MAX_LAYERS]; float params[MW_PONNI_MAX_PARAMS]; };
__global__ __launch_bounds__(256) void k_ponni_generic(PonniStack P, long long batch, const float *__restrict__ in, float *__restrict__ out) {
#pragma clang fp contract(off)
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c >= batch) return;
  float a[MW_PONNI_MAX_WIDTH], b[MW_PONNI_MAX_WIDTH];
  int n = P.n_in[0];
  for (int i = 0; i < n; i++) a[i] = in[(long long)i * batch + c];
  for (int l = 0; l < P.nlayers; l++) {
    const float *w = P.params + P.off[l];
    if (P.kind[l] == 0) {                                        // Matvec: weights (n_in, n_out), y = x W
      for (int o = 0; o < P.n_out[l]; o++) { float acc = 0.f; for (int i = 0; i < n; i++) acc += a[i] * w[i * P.n_out[l] + o]; b[o] = acc; }
      n = P.n_out[l];
      for (int o = 0; o < n; o++) a[o] = b[o];
    } else if (P.kind[l] == 1) { for (int o = 0; o < n; o++) a[o] = a[o] + w[o]; }
    else { const float sl = P.slope[l]; for (int o = 0; o < n; o++) a[o] = a[o] > 0.f ? a[o] : sl * a[o]; }
  }
  for (int o = 0; o < n; o++) out[(long long)o * batch + c] = a[o];
}

} // namespace mw

using namespace mw;

// STRICT form (mw_mlp_set_strict(1)): thread = cell, plain fp32 loops in INDEX ORDER, no contraction -- the order in which the layers
// are defined (Matvec, Bias, Relu, Matvec, Bias; microphysics_kessler_ponni.h:103-110) and in which the CPU restatement accumulates:
// bit-identical to it.  (The MFMA kernels sum the same products in the matrix cores' order: 1e-5 on the fp32 outputs.)
struct MlpRef { float W1[50], b1[10], W2[40], b2[4]; double in_min[5], in_rng[5], out_min[4], out_rng[4]; };
__global__ __launch_bounds__(256) void k_mlp_strict(MlpRef P, long long n, const double *__restrict__ temp, const double *__restrict__ rho_d,
                                                    const double *__restrict__ rho_v, const double *__restrict__ rho_c, const double *__restrict__ rho_r,
                                                    double *__restrict__ temp_out, double *__restrict__ rho_v_out, double *__restrict__ rho_c_out,
                                                    double *__restrict__ rho_r_out) {
#pragma clang fp contract(off)
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const double in[5] = {temp[c], rho_d[c], rho_v[c], rho_c[c], rho_r[c]};
  float x[5], h[10], y[4];
#pragma unroll
  for (int i = 0; i < 5; i++) x[i] = (float)((in[i] - P.in_min[i]) / P.in_rng[i]);                 // :182-186 (fp64, stored to float)
#pragma unroll
  for (int o = 0; o < 10; o++) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) acc += x[i] * P.W1[i * 10 + o];
    acc = acc + P.b1[o];
    h[o] = acc > 0.f ? acc : 0.1f * acc;
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 10; i++) acc += h[i] * P.W2[i * 4 + o];
    y[o] = acc + P.b2[o];
  }
  temp_out[c]  =           y[0] * P.out_rng[0] + P.out_min[0];                                      // :198-201
  rho_v_out[c] = fmax(0.0, y[1] * P.out_rng[1] + P.out_min[1]);
  rho_c_out[c] = fmax(0.0, y[2] * P.out_rng[2] + P.out_min[2]);
  rho_r_out[c] = fmax(0.0, y[3] * P.out_rng[3] + P.out_min[3]);
}
// The MFMA operand images of the 5 -> 10 -> 4 stack (see the header of this file): A operands per lane, C initialisers = the biases.
static void build_operand_images(MlpP &P, const float *W1, const float *b1, const float *W2, const float *b2) {
  memset(&P, 0, sizeof(P));
  auto rho = [](int u) { return (u / 3) * 4 + (u % 3); };      // hidden unit u -> D1 row with row % 4 < 3
  for (int lane = 0; lane < 64; lane++) {
    int o = lane & 15, g = lane >> 4;
    int u = -1;
    for (int uu = 0; uu < 10; uu++) if (rho(uu) == o) u = uu;
    for (int m = 0; m < 2; m++) { int in = 4 * m + g; P.a1[m][lane] = (u >= 0 && in < 5) ? W1[in * 10 + u] : 0.f; }
    int n = (o % 4 == 0) ? o / 4 : -1;                          // output n lives at row 4n
    for (int j = 0; j < 3; j++) {
      int row = 4 * g + j, uk = -1;                             // k-slot g of MFMA j is hidden row 4g + j
      for (int uu = 0; uu < 10; uu++) if (rho(uu) == row) uk = uu;
      P.a2[j][lane] = (n >= 0 && uk >= 0) ? W2[uk * 4 + n] : 0.f;
    }
  }
  for (int g = 0; g < 4; g++) {
    for (int r = 0; r < 4; r++) { int row = 4 * g + r, u = -1; for (int uu = 0; uu < 10; uu++) if (rho(uu) == row) u = uu;
                                  P.c1[g][r] = (u >= 0) ? b1[u] : 0.f; }
    P.c2[g] = b2[g];
  }
}

static thread_local int g_mlp_strict = 0;        // per calling thread: a rank harness with one host thread per rank may use different modes side by side
extern "C" int mw_mlp_set_strict(int strict) { g_mlp_strict = strict ? 1 : 0; return 0; }

extern "C" int mw_mlp_forward(long long ncells, const double *temp, const double *rho_d, const double *rho_v, const double *rho_c,
                              const double *rho_r, const float *W1, const float *b1, const float *W2, const float *b2,
                              const double *scl_in, const double *scl_out, double *temp_out, double *rho_v_out,
                              double *rho_c_out, double *rho_r_out, void *stream) {
  if (ncells < 1) MW_FAIL("mlp: ncells must be >= 1");
  if (!temp || !rho_d || !rho_v || !rho_c || !rho_r || !W1 || !b1 || !W2 || !b2 || !scl_in || !scl_out || !temp_out ||
      !rho_v_out || !rho_c_out || !rho_r_out) MW_FAIL("mlp: null pointer");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  if (g_mlp_strict) {
    MlpRef R;
    memcpy(R.W1, W1, sizeof(R.W1)); memcpy(R.b1, b1, sizeof(R.b1)); memcpy(R.W2, W2, sizeof(R.W2)); memcpy(R.b2, b2, sizeof(R.b2));
    for (int i = 0; i < 5; i++) { R.in_min[i] = scl_in[i * 2 + 0]; R.in_rng[i] = scl_in[i * 2 + 1] - scl_in[i * 2 + 0]; }
    for (int i = 0; i < 4; i++) { R.out_min[i] = scl_out[i * 2 + 0]; R.out_rng[i] = scl_out[i * 2 + 1] - scl_out[i * 2 + 0]; }
    hipLaunchKernelGGL(k_mlp_strict, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, ncells, temp, rho_d, rho_v, rho_c,
                       rho_r, temp_out, rho_v_out, rho_c_out, rho_r_out);
    MW_LAUNCH_CHECK();
    return 0;
  }
  MlpP P;          // operand images built per call (cheap: 104 weights)
  build_operand_images(P, W1, b1, W2, b2);
  for (int i = 0; i < 5; i++) { P.in_min[i] = scl_in[i * 2 + 0]; P.in_rng[i] = scl_in[i * 2 + 1] - scl_in[i * 2 + 0]; }
  for (int i = 0; i < 4; i++) { P.out_min[i] = scl_out[i * 2 + 0]; P.out_rng[i] = scl_out[i * 2 + 1] - scl_out[i * 2 + 0]; }
  constexpr int TILES = 4, PAIRS = 2;
  // bulk: spans of 32 cells with 16-byte accesses (needs 16-byte aligned arrays); remainder (and unaligned callers): k_mlp
  bool aligned = true;
  for (const void *q : {(const void *)temp, (const void *)rho_d, (const void *)rho_v, (const void *)rho_c, (const void *)rho_r, (const void *)temp_out,
                        (const void *)rho_v_out, (const void *)rho_c_out, (const void *)rho_r_out}) aligned = aligned && (((size_t)q & 15) == 0);
  const long long bulk = aligned ? (ncells / 32) * 32 : 0;
  if (bulk > 0) {
    long long waves_needed = (bulk / 32 + PAIRS - 1) / PAIRS;
    long long blocks = (waves_needed + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;                   // grid-stride beyond 16 blocks per CU
    hipLaunchKernelGGL(k_mlp_x2<PAIRS>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P, bulk, temp, rho_d, rho_v, rho_c,
                       rho_r, temp_out, rho_v_out, rho_c_out, rho_r_out);
    MW_LAUNCH_CHECK();
  }
  const long long rest = ncells - bulk;
  if (rest > 0) {
    long long ntiles = (rest + 15) / 16;
    long long waves_needed = (ntiles + TILES - 1) / TILES;
    long long blocks = (waves_needed + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_mlp<TILES>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P, rest, temp + bulk, rho_d + bulk, rho_v + bulk,
                       rho_c + bulk, rho_r + bulk, temp_out + bulk, rho_v_out + bulk, rho_c_out + bulk, rho_r_out + bulk);
    MW_LAUNCH_CHECK();
  }
  return 0;
}

// ponni::Inference<...>::forward_batch_parallel (microphysics_kessler_ponni.h:189) for a stack of ponni layers on fp32 device arrays in
// ponni's layout: in (n_in of the first layer, batch), out (n_out of the last, batch), batch fastest.  layers / params: HOST memory
// (params = the layers' weights back to back: a Matvec's (n_in, n_out) kernel in Keras order, a Bias's vector; offsets in floats).
// The surrogate's stack Matvec(5,10), Bias(10), Relu(10), Matvec(10,4), Bias(4) runs on the MFMA tiles; any other stack that fits
// the limits (MW_PONNI_MAX_*), and every stack under mw_mlp_set_strict(1), on the thread-per-element kernel (index order, no contraction).
extern "C" int mw_ponni_forward(const mw_ponni_layer_t *layers, int nlayers, const float *params, int nparams, long long batch,
                                const float *in, float *out, void *stream) {
  if (!layers || nlayers < 1 || !params || nparams < 0 || !in || !out) MW_FAIL("ponni_forward: null argument");
  if (batch < 1) MW_FAIL("ponni_forward: batch must be >= 1");
  if (mw_device_count() < 1) MW_FAIL("no HIP device available: libmw_cdna4 has no CPU fallback");
  if (nlayers > MW_PONNI_MAX_LAYERS) MW_FAIL("ponni_forward: more than " + std::to_string(MW_PONNI_MAX_LAYERS) + " layers");
  if (nparams > MW_PONNI_MAX_PARAMS) MW_FAIL("ponni_forward: more than " + std::to_string(MW_PONNI_MAX_PARAMS) + " parameters");
  // Inference::validate(): every layer's input size is its predecessor's output size, parameters inside the buffer
  int width = layers[0].n_in;
  for (int l = 0; l < nlayers; l++) {
    const mw_ponni_layer_t &L = layers[l];
    if (L.kind < 0 || L.kind > 2) MW_FAIL("ponni_forward: unknown layer kind");
    if (L.n_in != width) MW_FAIL("ponni_forward: layer " + std::to_string(l) + " expects " + std::to_string(L.n_in) + " inputs but its predecessor provides " + std::to_string(width));
    if (L.kind != 0 && L.n_out != L.n_in) MW_FAIL("ponni_forward: Bias / Relu layers keep the size");
    if (L.n_in < 1 || L.n_out < 1 || L.n_in > MW_PONNI_MAX_WIDTH || L.n_out > MW_PONNI_MAX_WIDTH) MW_FAIL("ponni_forward: layer width outside [1, " + std::to_string(MW_PONNI_MAX_WIDTH) + "]");
    const long long need = L.kind == 0 ? (long long)L.n_in * L.n_out : L.kind == 1 ? L.n_out : 0;
    if (need && (L.offset < 0 || L.offset + need > nparams)) MW_FAIL("ponni_forward: layer parameters outside the buffer");
    width = L.n_out;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool surrogate = nlayers == 5 && layers[0].kind == 0 && layers[0].n_in == 5 && layers[0].n_out == 10 && layers[1].kind == 1 &&
                         layers[2].kind == 2 && layers[3].kind == 0 && layers[3].n_out == 4 && layers[4].kind == 1;
  if (surrogate && !g_mlp_strict) {
    MlpP P;
    build_operand_images(P, params + layers[0].offset, params + layers[1].offset, params + layers[3].offset, params + layers[4].offset);
    constexpr int TILES = 4;
    long long blocks = (((batch + 15) / 16 + TILES - 1) / TILES + 3) / 4;
    blocks = std::max<long long>(1, std::min<long long>(blocks, 256 * 16));
    hipLaunchKernelGGL(k_mlp_f32<TILES>, dim3((unsigned)blocks), dim3(256), 0, st, P, layers[2].negative_slope, batch, in, out);
    MW_LAUNCH_CHECK();
    return 0;
  }
  PonniStack S;
  memset(&S, 0, sizeof(S));
  S.nlayers = nlayers;
  for (int l = 0; l < nlayers; l++) { S.kind[l] = layers[l].kind; S.n_in[l] = layers[l].n_in; S.n_out[l] = layers[l].n_out; S.off[l] = layers[l].offset; S.slope[l] = layers[l].negative_slope; }
  memcpy(S.params, params, sizeof(float) * (size_t)nparams);
  hipLaunchKernelGGL(k_ponni_generic, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, S, batch, in, out);
  MW_LAUNCH_CHECK();
  return 0;
}

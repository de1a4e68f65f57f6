// WENO-5 reconstruction of the two cell-edge values from a 5-cell stencil, device side.
//
// Follows weno::WenoLimiter<5>::compute_limited_coefs (model/modules/helpers/WenoLimiter.h:53-93) with the
// polynomial fits / TV / convexify of helpers/WenoLimiter_recon.h:12-15,37-56,84-103,155-162 and the
// coefs -> 2 GLL points transform of TransformMatrices.h:1132-1144 as used by reconstruct_gll_values
// (dynamics_euler_stratified_wenofv.h:556-571).
//
// Two variants:
//   weno5_edges_strict : the reference's exact operation order, contraction off (diagnostic / parity proof)
//   weno5_edges_fast   : same mathematics re-associated for CDNA4 fp64 VALU -- 2 divisions instead of 16,
//                        FMA contraction on.  Differences are O(1e-16..1e-15) relative per call.
#pragma once
#include <hip/hip_runtime.h>

namespace mw {

// `x_fp` literals in the reference are long double -> double (main_header.h:61-63).  All constants used by
// WenoLimiter<5> are rationals whose double rounding is the same either way (tests/test_gpu_golden_vectors.py: the strict
// routine below reproduces vectors computed through the long-double route bit for bit), so plain double literals are exact here.
#define MW_C(x) (x)

// convexified ideal weights idl = (1,2,1,1000)/1004  (WenoLimiter.h:55-66); computed with the reference's
// division so that the bits match the oracle.
__device__ __forceinline__ void weno5_ideal(double &iL, double &iC, double &iR, double &iH) {
  const double tot = 1.0 + 2.0 + 1.0 + 1.e3;
  iL = 1.0 / tot; iC = 2.0 / tot; iR = 1.0 / tot; iH = 1.e3 / tot;
}

__device__ __forceinline__ void weno5_edges_strict(double s0, double s1, double s2, double s3, double s4,
                                                    double &left, double &right) {
#pragma clang fp contract(off)
  // coefs3_shift1/2/3, coefs5_shift3
  double L0 = -MW_C(0.041666666666666666666666666666666666667)*s0+MW_C(0.083333333333333333333333333333333333333)*s1+MW_C(0.95833333333333333333333333333333333333)*s2;
  double L1 = MW_C(0.5)*s0-MW_C(2.0)*s1+MW_C(1.5)*s2;
  double L2 = MW_C(0.5)*s0-MW_C(1.0)*s1+MW_C(0.5)*s2;
  double C0 = -MW_C(0.041666666666666666666666666666666666667)*s1+MW_C(1.0833333333333333333333333333333333333)*s2-MW_C(0.041666666666666666666666666666666666667)*s3;
  double C1 = -MW_C(0.5)*s1+MW_C(0.5)*s3;
  double C2 = MW_C(0.5)*s1-MW_C(1.0)*s2+MW_C(0.5)*s3;
  double R0 = MW_C(0.95833333333333333333333333333333333333)*s2+MW_C(0.083333333333333333333333333333333333333)*s3-MW_C(0.041666666666666666666666666666666666667)*s4;
  double R1 = -MW_C(1.5)*s2+MW_C(2.0)*s3-MW_C(0.5)*s4;
  double R2 = MW_C(0.5)*s2-MW_C(1.0)*s3+MW_C(0.5)*s4;
  double H0 = MW_C(0.0046875)*s0-MW_C(0.060416666666666666666666666666666666667)*s1+MW_C(1.1114583333333333333333333333333333333)*s2-MW_C(0.060416666666666666666666666666666666667)*s3+MW_C(0.0046875)*s4;
  double H1 = MW_C(0.10416666666666666666666666666666666667)*s0-MW_C(0.70833333333333333333333333333333333333)*s1+MW_C(0.70833333333333333333333333333333333333)*s3-MW_C(0.10416666666666666666666666666666666667)*s4;
  double H2 = -MW_C(0.0625)*s0+MW_C(0.75)*s1-MW_C(1.375)*s2+MW_C(0.75)*s3-MW_C(0.0625)*s4;
  double H3 = -MW_C(0.083333333333333333333333333333333333333)*s0+MW_C(0.16666666666666666666666666666666666667)*s1-MW_C(0.16666666666666666666666666666666666667)*s3+MW_C(0.083333333333333333333333333333333333333)*s4;
  double H4 = MW_C(0.041666666666666666666666666666666666667)*s0-MW_C(0.16666666666666666666666666666666666667)*s1+MW_C(0.25)*s2-MW_C(0.16666666666666666666666666666666666667)*s3+MW_C(0.041666666666666666666666666666666666667)*s4;
  // TV
  double wL = MW_C(1.0)*(L1*L1)+MW_C(4.3333333333333333333333333333333333333)*(L2*L2);
  double wC = MW_C(1.0)*(C1*C1)+MW_C(4.3333333333333333333333333333333333333)*(C2*C2);
  double wR = MW_C(1.0)*(R1*R1)+MW_C(4.3333333333333333333333333333333333333)*(R2*R2);
  double wH = MW_C(1.0)*(H1*H1)+MW_C(4.3333333333333333333333333333333333333)*(H2*H2)+MW_C(0.5)*H1*H3+MW_C(39.1125)*(H3*H3)
             +MW_C(4.2)*H2*H4+MW_C(625.83571428571428571428571428571428571)*(H4*H4);
  double tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  double iL, iC, iR, iH;  weno5_ideal(iL, iC, iR, iH);
  wL = iL / (wL*wL + 1.e-20);
  wC = iC / (wC*wC + 1.e-20);
  wR = iR / (wR*wR + 1.e-20);
  wH = iH / (wH*wH + 1.e-20);
  tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  // cutoff == 0: "if (w <= cutoff) w = 0" is the identity for w >= 0
  if (wL <= 0.0) wL = 0;
  if (wC <= 0.0) wC = 0;
  if (wR <= 0.0) wR = 0;
  tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  double c0 = H0*wH + L0*wL + C0*wC + R0*wR;
  double c1 = H1*wH + L1*wL + C1*wC + R1*wR;
  double c2 = H2*wH + L2*wL + C2*wC + R2*wR;
  double c3 = H3*wH;
  double c4 = H4*wH;
  // coefs_to_gll_lower<5,2>: tmp = 0 + 1*c0 + (-/+0.5)*c1 + 0.25*c2 + (-/+0.125)*c3 + 0.0625*c4 (in that order)
  left  = (((c0 + (-0.5)*c1) + 0.25*c2) + (-0.125)*c3) + 0.0625*c4;
  right = (((c0 + ( 0.5)*c1) + 0.25*c2) + ( 0.125)*c3) + 0.0625*c4;
}

__device__ __forceinline__ void weno5_edges_fast(double s0, double s1, double s2, double s3, double s4,
                                                  double &left, double &right) {
#pragma clang fp contract(fast)
  // First differences; everything else is built from them.  Names with a trailing 'p' are the reference's
  // coefficients times a power-of-two / small-integer factor that is folded into the constants further down.
  const double a = s1 - s0, b = s2 - s1, c = s3 - s2, d = s4 - s3;
  const double L2p = b - a, C2p = c - b, R2p = d - c;       // 2*coefs3_shift{1,2,3}(2)
  const double C1p = b + c;                                 // 2*coefs3_shift2(1) = s3 - s1
  // (the odd coefficients are carried DOUBLED and the four total variations times 4: powers of two, so every value below is the
  //  exact multiple of its plain form and the edge values come out bit for bit the same -- the halvings are simply never executed)
  const double L1d = 2.0*b + L2p;                           // 2 coefs3_shift1(1) =  s0 - 4 s1 + 3 s2
  const double R1d = 2.0*c - R2p;                           // 2 coefs3_shift3(1) = -3 s2 + 4 s3 - s4
  const double e = s4 - s0;
  const double sLR = L2p + R2p;
  const double H4p = sLR - 2.0*C2p;                         // 24*coefs5_shift3(4)
  const double H3p = e - 2.0*C1p;                           // 12*coefs5_shift3(3)
  const double H2p = 10.0*C2p - sLR;                        // 16*coefs5_shift3(2)
  const double H1d = (2.0*0.70833333333333333333333333333333333333)*C1p - (2.0*0.10416666666666666666666666666666666667)*e;  // 2 coefs5_shift3(1)
  // TV (WenoLimiter_recon.h:37-56) with the scale factors folded into the constants; all four times 4
  const double k133 = 4.3333333333333333333333333333333333333;              // 13/3
  const double tL = L1d*L1d + k133*(L2p*L2p);
  const double tC = C1p*C1p + k133*(C2p*C2p);               // (C1p = 2 coefs3_shift2(1))
  const double tR = R1d*R1d + k133*(R2p*R2p);
  const double tH = H1d*(H1d + (1.0/12.0)*H3p) + H2p*((4.3333333333333333333333333333333333333/64.0)*H2p + (4.2/96.0)*H4p)
                  + (39.1125/36.0)*(H3p*H3p) + (625.83571428571428571428571428571428571/144.0)*(H4p*H4p);
  // convexify #1 divides every t by S = sum(t) (when S > 1e-20); then w_i = idl_i / (t_i^2 + 1e-20).  Scaling all
  // four denominators by S^2 leaves the normalised weights unchanged:  d_i = t_i^2 + 1e-20 S^2   (no division).
  // (with the t_i carried times 4: S is 4 x the reference's sum -- its threshold 1e-20 becomes 4e-20 -- and the d_i are 16 x)
  const double S = (tL + tC) + (tR + tH);
  const double eS = (S > 4.0*1.e-20) ? (1.e-20*S)*S : 16.0*1.e-20;
  const double dL = tL*tL + eS, dC = tC*tC + eS, dR = tR*tR + eS, dH = tH*tH + eS;
  // normalised w_i = (idl_i / d_i) / sum_j (idl_j / d_j)  =  idl_i prod_{j != i} d_j / N   (idl = 1,2,1,1000; /1004 cancels)
  const double dLC = dL*dC, dRH = dR*dH;
  const double nL = dC*dRH;
  const double nC = (dL + dL)*dRH;
  const double nR = dLC*dH;
  const double nH = (1.e3*dLC)*dR;
  const double N = (nL + nC) + (nR + nH);
  // v_rcp_f64 (measured: 4.6e-8 relative) + ONE Newton step = 2.2e-15 relative (a second step would give the correctly
  // rounded quotient).  rN only scales the deviation of the edge values from the cell mean, so this is <= 2.2e-15 of that
  // deviation -- four orders below the parity tolerance -- and two VALU instructions less in each of the ~24 reconstructions
  // per cell and stage.
  double rN = __builtin_amdgcn_rcp(N);
  rN = rN + rN*(1.0 - N*rN);
  // (the reference's 2nd/3rd convexify only re-normalise weights that already sum to 1)
  // Limited polynomial (un-normalised weights n_i, sum N) evaluated at -1/2 and +1/2.  Every candidate preserves the cell mean:
  // its constant coefficient is s2 - c2/12 (- c4/80 for the 5th-order one), so the even part  c0 + c2/4 + c4/16  collapses to
  //   s2 N + (1/4 - 1/12) c2 + (1/16 - 1/80) c4  =  s2 N + c2/6 + c4/20
  // and the constant coefficients never have to be formed.
  // The 5th-order candidate's odd (c1/2 + c3/8) and even (c2/6 + c4/20) parts are combined before the weighting:
  // (the factors 1/2 and 1/12 of the two sums are applied once, to the normalised sums, instead of inside them: od2 = 2 od, ev12 = 12 ev)
  const double oH2 = H1d + (0.5/12.0)*H3p;                                 // (odd part, times 4 with the doubled coefficients)
  const double eH12 = 0.125*H2p + (1.0/40.0)*H4p;                          // H2 = H2p/16 -> c2/6 = H2p/96;  c4/20 = H4p/480
  const double od2 = oH2*nH + (L1d*nL + C1p*nC + R1d*nR);
  const double ev12 = eH12*nH + (L2p*nL + C2p*nC + R2p*nR);                // even part minus s2 N   (c2 = coefs3(2) = X2p/2)
  const double r12 = (1.0/12.0)*rN, rh = 0.25*rN;
  const double base = fma(ev12, r12, s2);                                  // (explicit: every instantiation contracts the same way)
  left  = fma(-od2, rh, base);
  right = fma(od2, rh, base);
}

// ---------------------------------------------------------------------------------------------------------------------
// WENO-3 (the reference's -DMW_ORD=3 build, build/machines/aws/aws_a100_gpu.env:21): weno::WenoLimiter<3>::compute_limited_coefs
// (helpers/WenoLimiter.h:12-50: ideal weights 1, 1, 500 convexified; candidates coefs2_shift1, coefs2_shift2, coefs3_shift2 of
// WenoLimiter_recon.h:72-96; TV :29-42) + coefs_to_gll_lower<3,2> (TransformMatrices.h:300-308).  STRICT: the reference's
// operation order, contraction off; otherwise the same statements with FMA contraction.
// ---------------------------------------------------------------------------------------------------------------------
// (The statements live in a macro and are expanded inside both functions: a `#pragma clang fp contract` binds lexically, it does
//  not follow a call into a helper.)
#define MW_WENO3_STATEMENTS(s0, s1, s2, left, right)                                                                          \
  const double L0 = 1.0 * s1, L1 = -1.0 * s0 + 1.0 * s1;                        /* coefs2_shift1(s0, s1) */                     \
  const double R0 = 1.0 * s1, R1 = -1.0 * s1 + 1.0 * s2;                        /* coefs2_shift2(s1, s2) */                     \
  const double H0 = -MW_C(0.041666666666666666666666666666666666667)*s0+MW_C(1.0833333333333333333333333333333333333)*s1-MW_C(0.041666666666666666666666666666666666667)*s2; \
  const double H1 = -MW_C(0.5)*s0+MW_C(0.5)*s2;                                                                                \
  const double H2 = MW_C(0.5)*s0-MW_C(1.0)*s1+MW_C(0.5)*s2;                                                                    \
  double wL = 1.0 * (L1 * L1), wR = 1.0 * (R1 * R1);                                                                           \
  double wH = 1.0 * (H1 * H1) + MW_C(4.3333333333333333333333333333333333333) * (H2 * H2);                                     \
  double tot = wL + wR + wH;                                                                                                   \
  if (tot > 1.e-20) { wL /= tot; wR /= tot; wH /= tot; }                                                                       \
  const double itot = 1.0 + 1.0 + 5.e2;                                          /* ctor: convexify(1, 1, 5e2) */               \
  const double iL = 1.0 / itot, iR = 1.0 / itot, iH = 5.e2 / itot;                                                             \
  wL = iL / (wL * wL + 1.e-20);                                                                                                \
  wR = iR / (wR * wR + 1.e-20);                                                                                                \
  wH = iH / (wH * wH + 1.e-20);                                                                                                \
  tot = wL + wR + wH;                                                                                                          \
  if (tot > 1.e-20) { wL /= tot; wR /= tot; wH /= tot; }                                                                       \
  if (wL <= 0.0) wL = 0;                                                         /* cutoff == 0 */                              \
  if (wR <= 0.0) wR = 0;                                                                                                       \
  tot = wL + wR + wH;                                                                                                          \
  if (tot > 1.e-20) { wL /= tot; wR /= tot; wH /= tot; }                                                                       \
  const double c0 = H0 * wH + L0 * wL + R0 * wR;                                                                               \
  const double c1 = H1 * wH + L1 * wL + R1 * wR;                                                                               \
  const double c2 = H2 * wH;                                                                                                   \
  left  = (c0 + (-0.5) * c1) + 0.25 * c2;                                                                                      \
  right = (c0 + ( 0.5) * c1) + 0.25 * c2;
__device__ __forceinline__ void weno3_edges_strict(double s0, double s1, double s2, double &left, double &right) {
#pragma clang fp contract(off)
  MW_WENO3_STATEMENTS(s0, s1, s2, left, right)
}
#undef MW_WENO3_STATEMENTS
// The same mathematics re-associated like weno5_edges_fast (one reciprocal instead of nine divisions, no constant coefficients:
// every candidate preserves the cell mean): 38 VALU instructions.
//   a = s1 - s0, b = s2 - s1;  L1 = a, R1 = b, H1 = (a + b)/2, H2 = (b - a)/2
//   t_L = a^2, t_R = b^2, t_H = H1^2 + 13/3 H2^2;  S = sum t;  d_i = t_i^2 + 1e-20 S^2 (convexify #1 folded in, as in WENO-5)
//   normalised weights = idl_i prod_{j != i} d_j / N  (idl = 1, 1, 500; the /502 cancels)
//   edge values = s1 + (c2/6 -+ c1/2) with c1 = H1 w_H + a w_L + b w_R, c2 = H2 w_H   (1/4 - 1/12 = 1/6)
__device__ __forceinline__ void weno3_edges_fast(double s0, double s1, double s2, double &left, double &right) {
#pragma clang fp contract(fast)
  const double a = s1 - s0, b = s2 - s1;
  const double C1p = a + b, H2p = b - a;                    // 2 H1, 2 H2
  const double hC = 0.5 * C1p;
  const double tL = a * a, tR = b * b;
  const double tH = hC * hC + 1.0833333333333333333333333333333333333 * (H2p * H2p);      // (13/3)/4
  const double S = (tL + tR) + tH;
  const double eS = (S > 1.e-20) ? (1.e-20 * S) * S : 1.e-20;
  const double dL = tL * tL + eS, dR = tR * tR + eS, dH = tH * tH + eS;
  const double nL = dR * dH, nR = dL * dH, nH = (5.e2 * dL) * dR;
  const double N = (nL + nR) + nH;
  double rN = __builtin_amdgcn_rcp(N);
  rN = rN + rN * (1.0 - N * rN);                            // (one Newton step: see weno5_edges_fast)
  const double od = 0.5 * (hC * nH + (a * nL + b * nR));
  const double ev = (1.0 / 12.0) * (H2p * nH);              // c2/6 = H2p/12
  left  = s1 + (ev - od) * rN;
  right = s1 + (ev + od) * rN;
}
// A marching kernel's register window of ORD cells (centre ORD/2) -> the centre cell's two edge values
template <int ORD>
__device__ __forceinline__ void weno_window_edges(const double (&w)[ORD], double &left, double &right) {
  static_assert(ORD == 3 || ORD == 5, "the marching kernels exist for WENO orders 3 and 5");
  if (ORD == 3) weno3_edges_fast(w[0], w[1], w[ORD - 1], left, right);
  else          weno5_edges_fast(w[0], w[1], w[ORD / 2], w[ORD - 2], w[ORD - 1], left, right);
}

// ---------------------------------------------------------------------------------------------------------------------
// WENO-7 and WENO-9 (MW_ORD = 7 / 9, dynamics_euler_stratified_wenofv.h:24-28): weno::WenoLimiter<7> / <9>::compute_limited_coefs
// (helpers/WenoLimiter.h:95-137, :141-194) -- the three 3-cell candidates around the centre as in WENO-5, the high-order
// candidate over all N cells, ideal weights 1, 2, 1, 1e5 (1e8) convexified -- + coefs_to_gll_lower<N,2>.  The polynomial-fit and
// total-variation constants come from mw_weno79.h, which tools/gen_weno_tables.py derives from their definitions.
// Always the reference's operation order with contraction off (see weno79_edges_fast).
// ---------------------------------------------------------------------------------------------------------------------
} // namespace mw
#include "mw_weno79.h"
namespace mw {

template <int N, bool STRICT>
__device__ __forceinline__ void weno79_edges_body(const double *s, double &left, double &right) {
#pragma clang fp contract(off)
  constexpr int c = (N - 1) / 2;                                               // centre cell of the stencil
  const double s0 = s[c - 1], s1 = s[c], s2 = s[c + 1];
  // coefs3_shift1(s[c-2], s[c-1], s[c]), coefs3_shift2(s[c-1], s[c], s[c+1]), coefs3_shift3(s[c], s[c+1], s[c+2]) (WenoLimiter_recon.h:84-103)
  const double m2 = s[c - 2], p2 = s[c + 2];
  double L[3], C[3], R[3], H[N];
  L[0] = -MW_C(0.041666666666666666666666666666666666667)*m2+MW_C(0.083333333333333333333333333333333333333)*s0+MW_C(0.95833333333333333333333333333333333333)*s1;
  L[1] = MW_C(0.5)*m2-MW_C(2.0)*s0+MW_C(1.5)*s1;
  L[2] = MW_C(0.5)*m2-MW_C(1.0)*s0+MW_C(0.5)*s1;
  C[0] = -MW_C(0.041666666666666666666666666666666666667)*s0+MW_C(1.0833333333333333333333333333333333333)*s1-MW_C(0.041666666666666666666666666666666666667)*s2;
  C[1] = -MW_C(0.5)*s0+MW_C(0.5)*s2;
  C[2] = MW_C(0.5)*s0-MW_C(1.0)*s1+MW_C(0.5)*s2;
  R[0] = MW_C(0.95833333333333333333333333333333333333)*s1+MW_C(0.083333333333333333333333333333333333333)*s2-MW_C(0.041666666666666666666666666666666666667)*p2;
  R[1] = -MW_C(1.5)*s1+MW_C(2.0)*s2-MW_C(0.5)*p2;
  R[2] = MW_C(0.5)*s1-MW_C(1.0)*s2+MW_C(0.5)*p2;
  if (N == 7) mw_coefs7(H, s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
  else        mw_coefs9(H, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[N - 1]);
  double wL = MW_C(1.0)*(L[1]*L[1])+MW_C(4.3333333333333333333333333333333333333)*(L[2]*L[2]);
  double wC = MW_C(1.0)*(C[1]*C[1])+MW_C(4.3333333333333333333333333333333333333)*(C[2]*C[2]);
  double wR = MW_C(1.0)*(R[1]*R[1])+MW_C(4.3333333333333333333333333333333333333)*(R[2]*R[2]);
  double wH = (N == 7) ? mw_tv7(H) : mw_tv9(H);
  double tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  double iL = 1.0, iC = 2.0, iR = 1.0, iH = (N == 7) ? 1.e5 : 1.e8;             // ctor: convexify(1, 2, 1, 1e5 | 1e8)
  { const double it = iL + iC + iR + iH; iL /= it; iC /= it; iR /= it; iH /= it; }
  wL = iL / (wL*wL + 1.e-20);
  wC = iC / (wC*wC + 1.e-20);
  wR = iR / (wR*wR + 1.e-20);
  wH = iH / (wH*wH + 1.e-20);
  tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  if (wL <= 0.0) wL = 0;                                                        // cutoff == 0
  if (wC <= 0.0) wC = 0;
  if (wR <= 0.0) wR = 0;
  tot = wL + wC + wR + wH;
  if (tot > 1.e-20) { wL /= tot; wC /= tot; wR /= tot; wH /= tot; }
  H[0] = H[0]*wH + L[0]*wL + C[0]*wC + R[0]*wR;
  H[1] = H[1]*wH + L[1]*wL + C[1]*wC + R[1]*wR;
  H[2] = H[2]*wH + L[2]*wL + C[2]*wC + R[2]*wR;
#pragma unroll
  for (int m = 3; m < N; m++) H[m] = H[m]*wH;
  // coefs_to_gll_lower<N,2>: tmp = 0 + sum_s (-+1/2)^s * coef_s, in that order
  double lo = H[0], hi = H[0], pw = 1.0;
#pragma unroll
  for (int m = 1; m < N; m++) { pw *= 0.5; lo = lo + ((m & 1) ? -pw : pw) * H[m]; hi = hi + pw * H[m]; }
  left = lo; right = hi;
}
template <int N>
__device__ __forceinline__ void weno79_edges_strict(const double *s, double &left, double &right) {
#pragma clang fp contract(off)
  weno79_edges_body<N, true>(s, left, right);
}
// (No contracted variant: the 9-cell fit and its total variation cancel over constants up to 1.8e9 with an ideal weight of 1e8, and
//  FMA contraction alone moved a supercell step by 1.3e-11 of the field's scale -- more than the 1e-11 the parity tests allow.
//  These orders are not a performance path; both run-time modes use the reference's operation order.)
template <int N>
__device__ __forceinline__ void weno79_edges_fast(const double *s, double &left, double &right) {
#pragma clang fp contract(off)
  weno79_edges_body<N, false>(s, left, right);
}

} // namespace mw

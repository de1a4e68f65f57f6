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
//                        FMA contraction on.  Differences are O(1e-16) relative per call.
#pragma once
#include <hip/hip_runtime.h>

namespace mw {

// `x_fp` literals in the reference are long double -> double (main_header.h:61-63).  All constants used by
// WenoLimiter<5> are rationals whose double rounding is the same either way (verified on the host by
// tests/test_constants.py against the oracle's long-double route), so plain double literals are exact here.
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
  // second differences are shared by the three quadratic fits (L2, C2, R2) and feed H2/H4
  const double d01 = s1 - s0, d12 = s2 - s1, d23 = s3 - s2, d34 = s4 - s3;
  const double L2 = 0.5*(d12 - d01);            // 0.5 s0 - s1 + 0.5 s2
  const double C2 = 0.5*(d23 - d12);
  const double R2 = 0.5*(d34 - d23);
  const double L1 = d12 + L2;                   // 0.5 s0 - 2 s1 + 1.5 s2  = (s2-s1) + L2
  const double C1 = 0.5*(s3 - s1);
  const double R1 = d23 - R2;                   // -1.5 s2 + 2 s3 - 0.5 s4 = (s3-s2) - R2
  const double k24 = 0.041666666666666666666666666666666666667;   // 1/24
  const double L0 = s2 - k24*(2.0*L2);          // -1/24 s0 + 1/12 s1 + 23/24 s2 = s2 - (1/12) L2
  const double C0 = s2 - k24*(2.0*C2);
  const double R0 = s2 - k24*(2.0*R2);
  // quartic fit
  const double H4 = (2.0*k24)*((L2 + R2) - 2.0*C2);   // 1/24 (s0 - 4 s1 + 6 s2 - 4 s3 + s4) = 1/12 (L2 + R2 - 2 C2)
  const double H3 = 0.083333333333333333333333333333333333333*((s4 - s0) - 2.0*(s3 - s1));
  const double H2x = -0.0625*(s0 + s4) + 0.75*(s1 + s3) - 1.375*s2;
  const double H1 = 0.10416666666666666666666666666666666667*(s0 - s4) + 0.70833333333333333333333333333333333333*(s3 - s1);
  const double H0 = 0.0046875*(s0 + s4) - 0.060416666666666666666666666666666666667*(s1 + s3) + 1.1114583333333333333333333333333333333*s2;
  const double k133 = 4.3333333333333333333333333333333333333;
  double tL = L1*L1 + k133*(L2*L2);
  double tC = C1*C1 + k133*(C2*C2);
  double tR = R1*R1 + k133*(R2*R2);
  double tH = H1*H1 + k133*(H2x*H2x) + 0.5*H1*H3 + 39.1125*(H3*H3) + 4.2*H2x*H4 + 625.83571428571428571428571428571428571*(H4*H4);
  // convexify #1: t_i / S  (skipped when S <= 1e-20, as the reference does)
  const double S = (tL + tC) + (tR + tH);
  const double rS = (S > 1.e-20) ? __builtin_amdgcn_rcp(S) : 1.0;
  // two Newton steps bring v_rcp_f64 to full fp64 accuracy
  double rS1 = rS;
  if (S > 1.e-20) { rS1 = rS + rS*(1.0 - S*rS); rS1 = rS1 + rS1*(1.0 - S*rS1); }
  tL *= rS1; tC *= rS1; tR *= rS1; tH *= rS1;
  // w_i = idl_i / (t_i^2 + eps), then normalised: the common 1/prod(d) cancels ->
  //   w_i  ~  idl_i * prod_{j != i} d_j        (d in [1e-20, 1]: no under/overflow in fp64)
  const double dL = tL*tL + 1.e-20, dC = tC*tC + 1.e-20, dR = tR*tR + 1.e-20, dH = tH*tH + 1.e-20;
  const double dLC = dL*dC, dRH = dR*dH;
  double nL = dC*dRH;            // idl_L = 1/1004 ; the common 1/1004 cancels in the normalisation
  double nC = 2.0*(dL*dRH);
  double nR = dLC*dH;
  double nH = 1.e3*(dLC*dR);
  const double N = (nL + nC) + (nR + nH);
  double rN = __builtin_amdgcn_rcp(N);
  rN = rN + rN*(1.0 - N*rN);
  rN = rN + rN*(1.0 - N*rN);
  const double wL = nL*rN, wC = nC*rN, wR = nR*rN, wH = nH*rN;
  // (third convexify: weights already sum to 1 within rounding)
  const double c0 = H0*wH + L0*wL + C0*wC + R0*wR;
  const double c1 = H1*wH + L1*wL + C1*wC + R1*wR;
  const double c2 = H2x*wH + L2*wL + C2*wC + R2*wR;
  const double c3 = H3*wH;
  const double c4 = H4*wH;
  const double ev = c0 + 0.25*c2 + 0.0625*c4;     // even part
  const double od = 0.5*c1 + 0.125*c3;            // odd part
  left  = ev - od;
  right = ev + od;
}

} // namespace mw

// oracle/pow_check.cpp -- TEST INFRASTRUCTURE ONLY (like everything under oracle/; the product never loads it).
//
// The checker of miniweatherml_amd/csrc/mw_glibc_pow.h, the strict kernel path's restatement of glibc's pow:
//   mwo_libm_pow            = std::pow of the host's C library -- what the CPU oracle (and a reference built with the YAKL serial
//                             backend) computes pow with (dynamics_euler_stratified_wenofv.h:401, :1935, :2009);
//   mwo_glibc_pow_restated  = the product header compiled for the HOST, same source the device compiles.
// tests/test_glibc_pow.py compares the two bit for bit on millions of arguments (no GPU needed), and the device routine against
// mwo_libm_pow on the GPU.
#include "../miniweatherml_amd/csrc/mw_glibc_pow.h"
#include <cmath>

extern "C" {

void mwo_libm_pow(long long n, const double *x, const double *y, double *out) {
  for (long long i = 0; i < n; i++) out[i] = std::pow(x[i], y[i]);
}

// returns the number of arguments on the restated main path (ok[i] = 1); out[i] is only defined there
long long mwo_glibc_pow_restated(long long n, const double *x, const double *y, double *out, unsigned char *ok) {
  long long cnt = 0;
  for (long long i = 0; i < n; i++) {
    double r = 0;
    const bool m = mw::glibc_pow_main(x[i], y[i], &r);
    ok[i] = m ? 1 : 0; out[i] = r; cnt += m;
  }
  return cnt;
}


void mwo_libm_exp(long long n, const double *x, double *out) {
  for (long long i = 0; i < n; i++) out[i] = std::exp(x[i]);
}
long long mwo_glibc_exp_restated(long long n, const double *x, double *out, unsigned char *ok) {
  long long cnt = 0;
  for (long long i = 0; i < n; i++) {
    double r = 0;
    const bool m = mw::glibc_exp_main(x[i], &r);
    ok[i] = m ? 1 : 0; out[i] = r; cnt += m;
  }
  return cnt;
}


void mwo_libm_cos(long long n, const double *x, double *out) {
  for (long long i = 0; i < n; i++) out[i] = std::cos(x[i]);
}
long long mwo_glibc_cos_restated(long long n, const double *x, double *out, unsigned char *ok) {
  long long cnt = 0;
  for (long long i = 0; i < n; i++) {
    double r = 0;
    const bool m = mw::glibc_cos_main(x[i], &r);
    ok[i] = m ? 1 : 0; out[i] = r; cnt += m;
  }
  return cnt;
}

}

// Issue-rate probe of a few fp64 VALU instructions on gfx950: N dependent-free chains per lane, 256 threads x 1024 blocks, cycles per wave-instruction.
//   hipcc -O2 --offload-arch=gfx950 tools/oneoff/rate_probe.cpp -o tools/oneoff/rate_probe && tools/oneoff/rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
  double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                                                         \
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));                                    \
    else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));                                            \
    else if (OP == 2) asm volatile("v_ldexp_f64 %0, %0, -2" : "+v"(x));                                                   \
    else if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));                                            \
    else if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));                                                         \
    else if (OP == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));                                            \
    else if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int *)&x) : "v"(*(int *)&b));                 \
    else if (OP == 7) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));                                                \
    else if (OP == 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
    STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int OP> double run(const char *name, double *d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;                                   // 8 blocks of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, 16);
  hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 8;        // wave-instructions
  const double per_simd = winstr / (256.0 * 4);
  printf("%-16s %8.3f ms   %.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  return ms;
}
int main() {
  double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
  const int it = 20000;
  run<0>("v_fma_f64", d, it); run<1>("v_mul_f64", d, it); run<3>("v_add_f64", d, it); run<2>("v_ldexp_f64", d, it); run<5>("v_max_f64", d, it);
  run<8>("v_cmp_lt_f64", d, it); run<6>("v_cndmask_b32", d, it); run<7>("v_mov_b64", d, it); run<4>("v_rcp_f64", d, it / 4);
  return 0;
}

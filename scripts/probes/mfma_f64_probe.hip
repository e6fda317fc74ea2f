// Probe: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on gfx950 (MI355X).
// Question behind it: is there an MFMA crossover for the fp64 contraction
// K = J^T M J at n <= 32?  Only if the matrix pipe does more fp64 FMAs per cycle than the VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
  double x0 = 1, x1 = 2, x2 = 3, x3 = 4, x4 = 5, x5 = 6, x6 = 7, x7 = 8;
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
    x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
  double* out; hipMalloc(&out, 1024 * 256 * 8 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, iters = 20000;           // 8 blocks of 4 waves per CU: 8 waves per SIMD
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); k_mfma<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * 4 * iters * 4;                 // wave-instructions
    printf("mfma_f64_16x16x4: %.3f ms, %.1f TFLOP/s (2048 flop each), %.2f ns per wave-instruction per SIMD\n", ms,
           n_mfma * 2048 / ms / 1e9, ms * 1e6 / (n_mfma / 1024));
    hipEventRecord(e0); k_fma<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double n_fma = (double)blocks * 4 * iters * 8;
    printf("v_fma_f64       : %.3f ms, %.1f TFLOP/s (128 flop each),  %.2f ns per wave-instruction per SIMD\n", ms,
           n_fma * 128 / ms / 1e9, ms * 1e6 / (n_fma / 1024));
  }
  return 0;
}

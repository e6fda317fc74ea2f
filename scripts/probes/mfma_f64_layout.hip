// Which lane/register holds which element for v_mfma_f64_16x16x4_f64 on gfx950?
// hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
  double ha[64], hb[64], hd[256], *da, *db, *dd;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + 0.37 * l + 0.01 * l * l; hb[l] = 2.0 - 0.11 * l + 0.003 * l * l; }
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dd);
  hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  // input hypothesis: A[i][kk] in lane 16*kk+i, B[kk][j] in lane 16*kk+j
  double D[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += ha[16 * kk + i] * hb[16 * kk + j]; D[i][j] = s; }
  int h1 = 0, h2 = 0, h3 = 0, h4 = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = hd[l * 4 + r];
    h1 += std::fabs(v - D[4 * (l / 16) + r][l % 16]) < 1e-9;     // i = 4*(lane/16) + r, j = lane%16
    h2 += std::fabs(v - D[4 * r + l / 16][l % 16]) < 1e-9;       // i = 4*r + lane/16,  j = lane%16
    h3 += std::fabs(v - D[l % 16][4 * (l / 16) + r]) < 1e-9;     // transposed variants
    h4 += std::fabs(v - D[l % 16][4 * r + l / 16]) < 1e-9;
  }
  printf("matches of 256: H1 i=4*(l/16)+r,j=l%%16: %d | H2 i=4*r+l/16,j=l%%16: %d | H3 i=l%%16,j=4*(l/16)+r: %d | H4 i=l%%16,j=4*r+l/16: %d\n", h1, h2, h3, h4);
  return 0;
}

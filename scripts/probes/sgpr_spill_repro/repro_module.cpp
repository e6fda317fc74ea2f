// The same experiment with the EXACT code objects hiprtc produced inside libhamk (round-1 device library):
// random8_hiprtc_default.hsaco (hamk_rkf45_k: 101 spilled SGPRs) and random8_hiprtc_nolicm.hsaco (0).
// Loads one with hipModuleLoadData and launches hamk_rkf45_k repeatedly on identical inputs, the
// register-scribbling kernel (and, as libhamk's callers did, the module's hamEqs and RK4 kernels) in between.
//   ./repro_module random8_hiprtc_default.hsaco      ./repro_module random8_hiprtc_nolicm.hsaco
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>
#include "scribble.inc"
#include "scribble_bisect.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const long long B = 4096; const int n = 3, reps = 10;
  std::ifstream fi(argv[1], std::ios::binary);
  std::vector<char> co((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
  std::vector<double> in(2 * (size_t)n * B);
  FILE* f = std::fopen("inputs_4096.bin", "rb");
  if (!f || std::fread(in.data(), 8, in.size(), f) != in.size()) return 2;
  std::fclose(f);
  hipModule_t mod; hipFunction_t rkf, ham, rk4;
  CK(hipModuleLoadData(&mod, co.data()));
  CK(hipModuleGetFunction(&rkf, mod, "hamk_rkf45_k")); CK(hipModuleGetFunction(&ham, mod, "hamk_hameqs_k")); CK(hipModuleGetFunction(&rk4, mod, "hamk_rk4_steps_k"));
  double *q0, *p0, *q, *p, *dq, *dp; int *st, *ns;
  const size_t bytes = (size_t)n * B * 8;
  CK(hipMalloc(&q0, bytes)); CK(hipMalloc(&p0, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&p, bytes)); CK(hipMalloc(&dq, bytes)); CK(hipMalloc(&dp, bytes));
  CK(hipMalloc(&st, B * 4)); CK(hipMalloc(&ns, B * 4));
  CK(hipMemcpy(q0, in.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(p0, in.data() + (size_t)n * B, bytes, hipMemcpyHostToDevice));
  std::vector<std::vector<double>> res(reps, std::vector<double>(2 * (size_t)n * B));
  std::vector<std::vector<int>> nsub(reps, std::vector<int>(B));
  long long b = B; double dt = 0.02, eps = 1.49012e-08, t0 = 0.0, h0 = dt / 100.0; int nt = 2, row0 = 1, inplace = 1, maxsub = 2000;
  const double* ts = nullptr;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemcpy(q, q0, bytes, hipMemcpyDeviceToDevice)); CK(hipMemcpy(p, p0, bytes, hipMemcpyDeviceToDevice));
    const char mode = argc > 2 ? argv[2][0] : 'x';           // x: all register files, v / a / s: one of them, 0: none
    const unsigned seed = 12345u + 7919u * r;
    if (mode == 'x') hamk_scribble_k<<<dim3(4096), dim3(256)>>>(seed, nullptr);
    if (mode == 'v') hamk_scribble_v_k<<<dim3(4096), dim3(256)>>>(seed, nullptr);
    if (mode == 'a') hamk_scribble_a_k<<<dim3(4096), dim3(256)>>>(seed, nullptr);
    if (mode == 's') hamk_scribble_s_k<<<dim3(4096), dim3(256)>>>(seed, nullptr);
    if (mode == 'b') scribble_blocks[atoi(argv[2] + 1) & 15]<<<dim3(4096), dim3(256)>>>(seed, nullptr);   // b<k>: VGPR block k only
    if ((r & 1) && !(argc > 3)) {                                            // every other time also the module's own other kernels
      const double *cq = q0, *cp = p0; void* a1[] = {&cq, &cp, &dq, &dp, &b, &st};
      CK(hipModuleLaunchKernel(ham, 16, 1, 1, 256, 1, 1, 0, nullptr, a1, nullptr));
      double d2 = 0.01; int two = 2; void* a2[] = {&dq, &dp, &b, &d2, &two, &st};
      CK(hipModuleLaunchKernel(rk4, 16, 1, 1, 256, 1, 1, 0, nullptr, a2, nullptr));
    }
    const double *cq = q, *cp = p;
    void* args[] = {&cq, &cp, &q, &p, &b, &nt, &ts, &t0, &dt, &h0, &eps, &eps, &row0, &inplace, &maxsub, &st, &ns};
    CK(hipModuleLaunchKernel(rkf, 16, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(res[r].data(), q, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(res[r].data() + (size_t)n * B, p, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nsub[r].data(), ns, B * 4, hipMemcpyDeviceToHost));
  }
  long long worst = 0;
  for (int r = 1; r < reps; ++r) {
    long long bad = 0;
    for (long long i = 0; i < B; ++i) {
      bool same = nsub[r][i] == nsub[0][i];
      for (int j = 0; j < 2 * n && same; ++j) same = std::memcmp(&res[r][(size_t)j * B + i], &res[0][(size_t)j * B + i], 8) == 0;
      bad += !same;
    }
    std::printf("launch %d vs launch 0: %lld of %lld lanes differ\n", r, bad, B);
    if (bad > worst) worst = bad;
  }
  std::printf("RESULT %s %s: worst %lld lanes\n", argv[1], worst ? "NONDETERMINISTIC" : "deterministic", worst);
  return 0;
}

// Standalone reproducer of the SGPR-spill defect of DESIGN.md section 6b (hipcc / ROCm 7.2, gfx950).
//
// system_random8.hip is the generated source of one random test system (tests/test_gpu_random_systems.py,
// seed 8; System 3 3) with the UNROLLED adaptive stepper, hamk_device_r01.hpp the device library exactly as
// it was when the defect was found (round 1).  Built with the library's own options the kernel
// hamk_rkf45_k spills 101 SGPRs and gives RUN-TO-RUN DIFFERENT results on a few percent of the lanes --
// although every lane is independent and the inputs are identical; built with
// `-mllvm -disable-machine-licm` it spills none and is deterministic.  The same source is correct on
// the host (tests/test_host_emulation.py::test_random_systems_on_host[8]).
//
//   ./run.sh        builds both variants and runs each: prints the number of lanes whose result
//                   differs between launches, and the spill counts of both builds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "system_random8.hip"
#include "scribble.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const long long B = 4096;
  const int n = HamkSys::N, reps = 8;
  std::vector<double> in(2 * (size_t)n * B);
  FILE* f = std::fopen(argc > 1 ? argv[1] : "inputs_4096.bin", "rb");
  if (!f || std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) { std::fprintf(stderr, "inputs_4096.bin?\n"); return 2; }
  std::fclose(f);
  double *q0, *p0, *q, *p; int *st, *ns;
  const size_t bytes = (size_t)n * B * 8;
  CK(hipMalloc(&q0, bytes)); CK(hipMalloc(&p0, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&p, bytes));
  CK(hipMalloc(&st, B * 4)); CK(hipMalloc(&ns, B * 4));
  CK(hipMemcpy(q0, in.data(), bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(p0, in.data() + (size_t)n * B, bytes, hipMemcpyHostToDevice));
  std::vector<std::vector<double>> res(reps, std::vector<double>(2 * (size_t)n * B));
  std::vector<std::vector<int>> nsub(reps, std::vector<int>(B));
  const double dt = 0.02, eps = 1.49012e-08;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemcpy(q, q0, bytes, hipMemcpyDeviceToDevice)); CK(hipMemcpy(p, p0, bytes, hipMemcpyDeviceToDevice));
    // other kernels leave other register contents behind: a kernel that fills every VGPR, AGPR and most
    // SGPRs of every SIMD with launch-dependent values runs before each launch (argv[2] = "0" skips it)
    if (!(argc > 2 && argv[2][0] == '0')) { hamk_scribble_k<<<dim3(4096), dim3(256)>>>(12345u + 7919u * r, nullptr); CK(hipDeviceSynchronize()); }
    // stepHam dt, in place: nt = 2, times (0, dt) as arguments, h0 = dt/100, row0 = 1, inplace = 1
    hamk_rkf45_k<<<dim3((unsigned)((B + 255) / 256)), dim3(256)>>>(q, p, q, p, B, 2, nullptr, 0.0, dt, dt / 100.0, eps, eps, 1, 1, 2000, st, ns);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(res[r].data(), q, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(res[r].data() + (size_t)n * B, p, bytes, hipMemcpyDeviceToHost));
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
  std::printf("RESULT %s: worst %lld lanes\n", worst ? "NONDETERMINISTIC" : "deterministic", worst);
  return 0;
}

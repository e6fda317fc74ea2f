// TEST INFRASTRUCTURE -- never part of libhamk.so.
// hamilton_amd/csrc/hamk_sample.hpp (the device sampler behind hamk_sample_batch) compiled for the host: the CPU suite
// checks its bits against the numpy sampler (hamilton_amd/examples.py) without a GPU.  Built with -ffp-contract=off;
// lerp_unfused's opaque statement is left out (HAMK_HOST_EMULATION).
#include "hip_shim.hpp"
#include "hamk_sample.hpp"

extern "C" void emu_sample(double* q, double* qd, long long B, long long first, unsigned long long seed, int n,
                           const double* qlo, const double* qhi, const double* dlo, const double* dhi) {
  HamkBoxes bx;
  std::memset(&bx, 0, sizeof bx);
  for (int j = 0; j < n; ++j) { bx.q_lo[j] = qlo[j]; bx.q_hi[j] = qhi[j]; bx.qd_lo[j] = dlo[j]; bx.qd_hi[j] = dhi[j]; }
  blockDim.x = 256;
  for (long long b0 = 0; b0 < B; b0 += 256) {
    blockIdx.x = (unsigned)(b0 / 256);
    for (unsigned t = 0; t < 256; ++t) { threadIdx.x = t; hamk_sample_k(q, qd, B, first, seed, n, bx); }
  }
}

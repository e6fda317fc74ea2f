// hamk_sample.hpp -- initial conditions of an ensemble, generated ON THE DEVICE from the global trajectory index
// ("per-index counter-based RNG -- splitmix64(seed ^ global_index k + field) -> U[0,1) -- so any
// GPU count / shard reproduces bit-identical inputs ... generated on-device from the global index -> no scatter needed").
// The reference has no counterpart (its demo starts ONE trajectory from a CLI-given Config, app/Examples.hs:230-359);
// this is the ensemble version of "choose an initial Config", one uniform box per coordinate.
//
// Bit-identical to hamilton_amd/examples.py sample_config / uniform01 (numpy, the CPU tests' sampler):
//   key = seed ^ (index * 0xD1342543DE82EF95);  z = splitmix64(key + field * 0x2545F4914F6CDD1D);
//   u = (z >> 11) * 2^-53;  value = lo + (hi - lo) * u            -- field 2j: q_j, field 2j + 1: qd_j
// The last line is three separately rounded IEEE operations in numpy: lerp_unfused below keeps the compiler from fusing
// the multiply and the add (the module is built with -ffp-contract=fast).
// One thread per trajectory, component-major stores q[j * B + i]: a wavefront writes 512 contiguous bytes per
// component.  Pure HBM-write work: 16 n bytes per trajectory.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

struct HamkBoxes { double q_lo[64], q_hi[64], qd_lo[64], qd_hi[64]; };   // n <= 64 (hamk.h); passed by value: 2 KiB of kernel arguments

namespace hamk {
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01(unsigned long long key, int field) {
  const unsigned long long z = splitmix64(key + (unsigned long long)field * 0x2545F4914F6CDD1Dull);
  return (double)(z >> 11) * 0x1p-53;                       // 53 bits: the conversion and the scaling are exact
}
// lo + (hi - lo) * u as numpy evaluates it: three separately rounded operations.  HIP's __dmul_rn / __dadd_rn are plain
// `*` and `+` to the compiler, and the module is built with -ffp-contract=fast: the first GPU run of this kernel differed
// from numpy in the last bit on some trajectories (a fused multiply-add).  Contraction is switched off for this function
// AND the product goes through an opaque statement, so no later pass can fuse it either.
__device__ __forceinline__ double lerp_unfused(double lo, double hi, double u) {
#pragma clang fp contract(off)
  double t = (hi - lo) * u;
#ifndef HAMK_HOST_EMULATION
  asm volatile("" : "+v"(t));
#endif
  return lo + t;
}
}  // namespace hamk

extern "C" __global__ void __launch_bounds__(256) hamk_sample_k(double* q, double* qd, long long B, long long first_index,
                                                                unsigned long long seed, int n, HamkBoxes bx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const unsigned long long key = seed ^ ((unsigned long long)(first_index + i) * 0xD1342543DE82EF95ull);
  for (int j = 0; j < n; ++j) {
    const double u = hamk::uniform01(key, 2 * j), w = hamk::uniform01(key, 2 * j + 1);
    q[(long long)j * B + i] = hamk::lerp_unfused(bx.q_lo[j], bx.q_hi[j], u);
    qd[(long long)j * B + i] = hamk::lerp_unfused(bx.qd_lo[j], bx.qd_hi[j], w);
  }
}

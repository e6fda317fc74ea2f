// TEST INFRASTRUCTURE -- never part of libhamk.so.
// Just enough of the HIP device environment to compile the library's device header
// (hamilton_amd/csrc/hamk_device.hpp) and a generated system for the HOST, so that the CPU test
// suite can run the generated coordinate map / potential, the jets of every opcode, the three AD
// strategies (incl. the generated reverse sweep) and the RK4 / RKF45 bodies against the oracle
// without a GPU.  What it cannot cover is the GPU compiler (those hazards are the GPU suite's job).
#pragma once
#define HAMK_HOST_EMULATION 1
#define __HIPCC_RTC__ 1            /* keeps hamk_device.hpp from including hip_runtime.h */
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict
#ifndef __shared__
#define __shared__ static                /* one OS thread at a time per block (lane kernels): a static array indexed by threadIdx */
#endif

struct emu_dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local emu_dim3 threadIdx, blockIdx, blockDim;

static inline int __double2hiint(double x) { int64_t b; std::memcpy(&b, &x, 8); return (int)(b >> 32); }
static inline int __double2loint(double x) { int64_t b; std::memcpy(&b, &x, 8); return (int)(b & 0xffffffff); }
static inline double __hiloint2double(int hi, int lo) {
  const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double x; std::memcpy(&x, &b, 8); return x;
}
static inline double __longlong_as_double(long long b) { double x; std::memcpy(&x, &b, 8); return x; }
static inline long long __double_as_longlong(double x) { long long b; std::memcpy(&b, &x, 8); return b; }
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __builtin_amdgcn_logf(x) ::log2f(x)
#define __builtin_amdgcn_exp2f(x) ::exp2f(x)

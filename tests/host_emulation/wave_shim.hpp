// TEST INFRASTRUCTURE -- never part of libhamk.so.
// Host emulation of the wave-cooperative kernels (hamilton_amd/csrc/hamk_wave.hpp): ONE OS THREAD PER
// LANE, 64 threads per wavefront, the wavefronts of a block run one after the other (the kernels only
// communicate within a wavefront).  Cross-lane primitives go through small exchange arrays with real
// barriers; v_mfma_f64_16x16x4_f64 is emulated with the operand/result layout measured on the MI355X
// (scripts/probes/mfma_f64_layout.hip): A[i][k] and B[k][j] in lane 16 k + i / 16 k + j, D[i][j] in
// lane l, register r with i = 4 r + l/16, j = l%16.
#pragma once
#include "hip_shim.hpp"
#include <barrier>
#include <thread>
#include <vector>

#ifndef __shared__
#define __shared__ static
#endif

struct EmuQuadBarrier { std::barrier<> b{4}; };
struct EmuWave {
  std::barrier<> bar{64};
  int xi[64];
  double xd[64], xa[64], xb[64];
  EmuQuadBarrier qbar[16];                        // hamk_quad.hpp: four lanes per trajectory, quads run independently
  double qd[64];
  int qi[64];
};
static EmuWave* emu_wave = nullptr;              // the wavefront being executed
static thread_local int emu_lane = 0;

static inline void emu_wave_barrier() { emu_wave->bar.arrive_and_wait(); }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()

// quad primitives of hamk_quad.hpp (on the device: DPP quad_perm moves, no memory)
static inline void emu_quad_barrier() { emu_wave->qbar[emu_lane >> 2].b.arrive_and_wait(); }
static inline double emu_quad_read(double x, int sel, int xor_mode) {
  emu_wave->qd[emu_lane] = x;
  emu_quad_barrier();
  const double r = emu_wave->qd[xor_mode ? (emu_lane ^ sel) : ((emu_lane & ~3) | sel)];
  emu_quad_barrier();
  return r;
}
static inline int emu_quad_read_i(int x, int mask) {
  emu_wave->qi[emu_lane] = x;
  emu_quad_barrier();
  const int r = emu_wave->qi[emu_lane ^ mask];
  emu_quad_barrier();
  return r;
}
static inline int __builtin_amdgcn_ds_bpermute(int byte_index, int v) {
  emu_wave->xi[emu_lane] = v;
  emu_wave_barrier();
  const int r = emu_wave->xi[(byte_index >> 2) & 63];
  emu_wave_barrier();
  return r;
}
static inline double emu_read_lane(double x, int lane) {     // v_readlane_b32 x 2: lane `lane`'s value for every lane
  emu_wave->xd[emu_lane] = x;
  emu_wave_barrier();
  const double r = emu_wave->xd[lane & 63];
  emu_wave_barrier();
  return r;
}
static inline double __shfl_xor(double x, int off, int width) {
  emu_wave->xd[emu_lane] = x;
  emu_wave_barrier();
  const double r = emu_wave->xd[(emu_lane & ~(width - 1)) | ((emu_lane ^ off) & (width - 1))];
  emu_wave_barrier();
  return r;
}
static inline int __shfl_xor(int x, int off, int width) {
  emu_wave->xi[emu_lane] = x;
  emu_wave_barrier();
  const int r = emu_wave->xi[(emu_lane & ~(width - 1)) | ((emu_lane ^ off) & (width - 1))];
  emu_wave_barrier();
  return r;
}
static inline int __any(int pred) {
  emu_wave->xi[emu_lane] = pred;
  emu_wave_barrier();
  int r = 0;
  for (int l = 0; l < 64; ++l) r |= (emu_wave->xi[l] != 0);
  emu_wave_barrier();
  return r;
}
typedef double emu_d4 __attribute__((vector_size(32)));
static inline emu_d4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emu_d4 acc, int, int, int) {
  emu_wave->xa[emu_lane] = a;
  emu_wave->xb[emu_lane] = b;
  emu_wave_barrier();
  const int j = emu_lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + (emu_lane >> 4);
    double s = acc[r];
    for (int k = 0; k < 4; ++k) s += emu_wave->xa[16 * k + i] * emu_wave->xb[16 * k + j];
    acc[r] = s;
  }
  emu_wave_barrier();
  return acc;
}

// run `kernel` for every lane of every wavefront that owns at least one of the B trajectories
// (G trajectories per wavefront, 4 wavefronts per 256-thread block)
template <class F> static void emu_launch_wave(long long B, int G, F kernel) {
  const long long blocks = (B + 4 * G - 1) / (4 * G);
  for (long long b = 0; b < blocks; ++b)
    for (int w = 0; w < 4; ++w) {
      if ((b * 4 + w) * G >= B) continue;                  // a wavefront of padding only: nothing to check
      EmuWave wave;
      emu_wave = &wave;
      std::vector<std::thread> lanes;
      for (int l = 0; l < 64; ++l)
        lanes.emplace_back([=, &kernel]() {
          emu_lane = l;
          blockDim.x = 256; blockIdx.x = (unsigned)b; threadIdx.x = (unsigned)(w * 64 + l);
          kernel();
        });
      for (auto& t : lanes) t.join();
    }
  emu_wave = nullptr;
}

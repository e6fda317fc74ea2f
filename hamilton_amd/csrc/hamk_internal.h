// hamk_internal.h -- shared between hamk_codegen.cpp and the rest of the host side (hamk_host.h) of libhamk.so
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// libhamk.so is built with -fvisibility=hidden: what it exports is exactly the C ABI of include/hamk.h (the host code is three
// translation units since round 4, and their shared C++ symbols are nobody else's business)
#pragma GCC visibility push(default)
#include "hamk.h"
#pragma GCC visibility pop

namespace hamk_host {

struct SystemDesc {
  int m = 0, n = 0, u_space = 0;
  bool mode_h = true;
  bool mode_r = false;          // second sweep in reverse mode (generated adjoint code) instead of Jet2<N>
  bool rk4_stage_loop = false;
  bool rkf_stage_loop = false;
  int rk4_min_waves = 1;        // __launch_bounds__ second argument of the RK4 kernel (waves per SIMD)
  int use_lut = 2;              // stepping kernels: sincos through the LDS table (hamk_device.hpp StageTrig: 0, 1, 2)
  int mapping = 1;              // HAMK_MAP_* (hamk.h)
  bool wave = false;            // mapping == HAMK_MAP_WAVE: wave-cooperative kernels (hamk_wave.hpp) instead of one trajectory per lane
  bool k_reassoc = true;        // mass_matrix summed with re-association allowed (hamk_device.hpp)
  bool k_symbolic = true;       // lane mapping, n <= 7: K derived symbolically where f is polynomial in sincos of polynomial arguments (hamk_codegen.cpp symbolic_mass_matrix)
  bool rkf_park = false;        // lane / quad mapping: the RKF45 stepper's vectors in a run-time-indexed private array
  bool rk4_park = false;        // lane mapping: RK4 stage loop parks y / acc in LDS across the right-hand side
  bool trig_const_vgpr = false; // lane mapping, 8 <= n <= 14: sincos_lut's fp64 literals live in vector registers (hamk_device.hpp LutK)
  bool rkf_two_waves = false;   // lane mapping, n <= 7: the parked RKF45 stepper at two wavefronts per SIMD (rows beyond a halved LDS share in registers)
  bool pair_rows = false;       // lane mapping, n = 8, 9: the parked stepper's LDS rows hold components in pairs (16-byte accesses; hamk_device.hpp HAMK_PAIR_ROWS)
  bool quad_dense = false;      // quad mapping, a coordinate map with a dense Jacobian: K accumulated in passes (hamk_quad.hpp assemble_dense)
  bool rkf_two_waves_off = false;  // ... built that way, its 256-register cap made THIS system's stepper spill: one wavefront (hamk_dispatch.cpp variant_for)
  std::vector<double> inertia;
  std::vector<hamk_op> f_ops;
  std::vector<int32_t> f_outs;
  std::vector<hamk_op> u_ops;
  int32_t u_out = 0;
};

// empty string = valid
std::string validate_tape(const hamk_op* ops, int nops, int n_in, const int32_t* outs, int n_out, const char* what);
std::string generate_source(const SystemDesc& d);
// distinct non-zero entries of the coordinate map's Jacobian, as expressions (hamk_codegen.cpp)
int distinct_jacobian_entries(const SystemDesc& d);
// fp64 operations of ONE first-order forward sweep of the coordinate map with compile-time seeds: per tape operation the number of
// gradient entries it has to COMPUTE (structural zeros and pass-through copies are free) -- what a lane of the quad mapping pays per pass
long long forward_gradient_work(const SystemDesc& d);
// a lane module of this system takes K and dT/dq from the symbolic mass matrix (hamk_codegen.cpp symbolic_mass_matrix)
bool symbolic_rhs_applies(const SystemDesc& d);

}  // namespace hamk_host

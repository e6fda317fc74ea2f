// hamk_host.h -- what the three host translation units of libhamk.so share (round 4: hamk_api.cpp split):
//   hamk_build.cpp     generated source -> gfx950 code objects: hiprtc, the on-disk cache, the two builds per module
//   hamk_dispatch.cpp  options -> specialisations (which lanes serve a trajectory, bodies, sincos policy ...), device
//                      binding, launches, the first-use self-check
//   hamk_api.cpp       the C ABI of include/hamk.h: argument checks, host-pointer staging, checkpoints
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "hamk_internal.h"

namespace hamk_host {

// Environment overrides of the library's own choices (which mapping, which body, which sincos ...) exist for the test suites and
// the A/B scripts.  They are read ONLY when HAMK_TEST_OVERRIDES=1 is set (tests/conftest.py and the scripts set it): a host
// process that happens to carry a HAMK_QUAD or HAMK_AD_MODE variable runs what hamk_options says, nothing else.  Not gated:
// HAMK_CACHE / HAMK_CACHE_DIR (where compiled code objects are kept) and HAMK_SELFCHECK_VERBOSE (diagnostics).
const char* test_env(const char* name);                    // getenv(name) under HAMK_TEST_OVERRIDES=1, else nullptr
int fail(int code, const std::string& msg);                // sets the thread's hamk_last_error text, returns code
const std::string& last_error_text();

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return hamk_host::fail(HAMK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define TRY0(expr) do { int rc0_ = (expr); if (rc0_ != HAMK_OK) return rc0_; } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_ != HAMK_OK) return rc_; } while (0)

enum KernelId { K_RK4, K_HAMEQS, K_COORDS, K_TO_PHASE, K_FROM_PHASE, K_OBSERVE, K_OBSERVE_CFG, K_RKF45, K_SCRIBBLE, K__COUNT };
extern const char* const kKernelNames[K__COUNT];

struct Sha256 {
  uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  unsigned char buf[64];
  size_t fill = 0;
  uint64_t total = 0;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const unsigned char* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
        0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
        0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
        0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
        0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
        0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      const uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    total += n;
    while (n) {
      const size_t k = std::min(n, sizeof buf - fill);
      std::memcpy(buf + fill, p, k);
      fill += k; p += k; n -= k;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(unsigned char out[32]) {
    const uint64_t bits = total * 8;
    const unsigned char one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    unsigned char len[8];
    for (int i = 0; i < 8; ++i) len[i] = (unsigned char)(bits >> (56 - 8 * i));
    update(len, 8);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (unsigned char)(h[i] >> 24); out[4 * i + 1] = (unsigned char)(h[i] >> 16); out[4 * i + 2] = (unsigned char)(h[i] >> 8); out[4 * i + 3] = (unsigned char)h[i]; }
  }
};

// One specialisation of the device library for a system: which lanes serve a trajectory (hamk.h HAMK_MAP_*) and the
// choices that go with it.  A handle builds the one its options name -- or, with mapping = HAMK_AUTO, the one a large
// ensemble uses -- when it is created, and the others the first time a launch asks for them.
struct Variant {
  int mapping = HAMK_MAP_LANE;
  SystemDesc desc;
  std::string source;
  std::vector<char> code;      // gfx950 code object (default options)
  std::vector<char> code2;     // the same source built without MachineLICM; empty unless some kernel is taken from it
  bool use2[K__COUNT] = {};    // kernel k comes from code2 (it spills no / fewer SGPRs there)
  std::string build_log;
  std::string build_info;      // per kernel: which build it comes from, bytes, spilled SGPRs
  int generation = 0;          // bumped whenever `code` is rebuilt (the self-check's recovery path)
  int self_check_rebuilds = 0;
  bool forced_rk4_body = false, forced_rkf_body = false;
  bool has[K__COUNT] = {};     // the kernels this module provides (the quad module: four of the eight)
};
constexpr int kMaxMap = 4;     // HAMK_MAP_* ids are 1..3

// The modules of one Variant on one device.
struct DevModule {
  hipModule_t module = nullptr;
  hipModule_t module2 = nullptr;
  hipFunction_t fn[K__COUNT] = {};
  bool self_checked = false;
  int code_generation = -1;     // Variant::generation the loaded modules were built from
  void unload() {
    if (module) { hipModuleUnload(module); module = nullptr; }
    if (module2) { hipModuleUnload(module2); module2 = nullptr; }
  }
};

// What a handle owns on ONE device.  A handle used from several devices (one process driving every
// GPU of a node, or a torch program whose tensors live on cuda:1 while cuda:0 is current) keeps one
// of these per device: modules stay loaded and staging buffers stay allocated when the calls
// alternate between devices.
struct DevState {
  int device = -1;
  DevModule mod[kMaxMap];
  // the stream this handle launches on ON THIS DEVICE (hamk_set_stream binds it to the device that is current
  // at the time: a stream belongs to one device, and a handle may be used from several)
  hipStream_t stream = nullptr;
  // grow-only device staging for HAMK_MEM_HOST calls (slot i serves the i-th staged array of a
  // call): the reference's own usage pattern is one small stepHam per frame (Examples.hs:429),
  // where a hipMalloc/hipFree pair per array per call would dominate
  std::vector<void*> stage_buf;
  std::vector<size_t> stage_cap;
  // a LARGE host-array ensemble is stepped piece by piece, the transfers of one piece under the kernel of another
  // (hamk_api.cpp rk4_steps_host_pieces): the copies' own stream and the events that order it against `stream`
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> piece_events;
  // pinned, device-mapped arena for SMALL host-pointer calls (the reference's one-trajectory
  // stepHam per frame): the kernel reads and writes host memory directly over PCIe -- a launch
  // and a stream synchronisation per call, no hipMemcpy at all.  The block is allocated COHERENT
  // (fine-grained, uncached on the device): the CPU rewrites it before every call, and
  // hipHostMallocMapped alone gives non-coherent memory whose lines the device may keep in L2
  // across launches.
  char* pin = nullptr;          // host address
  char* pin_dev = nullptr;      // the same block as the device sees it
  bool pin_failed = false;
  // hamk_sample_batch's kernel (hamk_sample.hpp; one module per process, loaded per device on first use)
  hipModule_t sample_module = nullptr;
  hipFunction_t sample_fn = nullptr;
  // small device scratch for evolveHam's time grid
  double* d_ts = nullptr;
  size_t d_ts_cap = 0;
  std::vector<double> h_ts;
  void release() {
    for (DevModule& m : mod) m.unload();
    if (sample_module) { hipModuleUnload(sample_module); sample_module = nullptr; sample_fn = nullptr; }
    if (d_ts) { hipFree(d_ts); d_ts = nullptr; d_ts_cap = 0; }
    for (void*& b : stage_buf) if (b) { hipFree(b); b = nullptr; }
    stage_cap.assign(stage_cap.size(), 0);
    if (pin) { hipHostFree(pin); pin = pin_dev = nullptr; }
    for (hipEvent_t e : piece_events) hipEventDestroy(e);
    piece_events.clear();
    if (copy_stream) { hipStreamDestroy(copy_stream); copy_stream = nullptr; }
  }
};

}  // namespace hamk_host

struct hamk_system {
  using SystemDesc = hamk_host::SystemDesc; using Variant = hamk_host::Variant; using DevState = hamk_host::DevState; using DevModule = hamk_host::DevModule;
  static constexpr int kMaxMap = hamk_host::kMaxMap;
  SystemDesc base;             // m, n, inertia, tapes: what every specialisation shares
  hamk_options opt;            // as the caller gave them (HAMK_AUTO where the choice is the library's)
  Variant* var[kMaxMap] = {};  // by mapping id; built on demand
  Variant* curv = nullptr;     // the specialisation the call in progress uses
  Variant* info = nullptr;     // the one the introspection entry points describe (hamk_system_describe_batch)
  // lazily bound to the calling thread's current device, one DevState per device ever used
  std::vector<DevState*> devs;
  DevState* cur = nullptr;
  int gsl_api = 2;             // which binding of hmatrix-gsl's gsl-ode.c stepHam/evolveHam follow (hamk.h)
  int max_substeps = 1 << 24;
  bool self_check_on = true, cache_on = true;
  int quad_eligible = -1;      // -1: not analysed yet
  int quad_dense_eligible = -1;  // -1: not analysed yet (hamk_dispatch.cpp quad_dense_eligible)
  int64_t ensemble_size = 0;   // hamk_options::ensemble_size: AUTO picks the mapping for THIS size instead of a launch's own B
  DevModule& mod() { return cur->mod[curv->mapping]; }
};

namespace hamk_host {

// ---- hamk_build.cpp ---------------------------------------------------------------------------------------------------
int compile_module(Variant* s, bool cache_on, bool no_machine_licm, std::vector<char>& code);
int build_code(Variant* s, bool cache_on, int force);      // force: 0 default build only, 1 without MachineLICM only, -1 per kernel
size_t kernel_code_bytes(const std::vector<char>& elf, const char* name);
size_t chosen_kernel_bytes(const Variant* s, int k);
int sgpr_spill_count(const std::vector<char>& elf, const char* kernel);
int vgpr_spill_count(const std::vector<char>& elf, const char* kernel);
int sample_code(bool cache_on, const std::vector<char>** out);        // hamk_sample.hpp's code object, compiled once per process

// ---- hamk_dispatch.cpp ------------------------------------------------------------------------------------------------
std::string check_options(const hamk_options& o, int n);
int build_force(const hamk_system* s);
int variant_for(hamk_system* s, int64_t B, int kernel, Variant** out);
int current_device_state(hamk_system* s);
int bind_device(hamk_system* s, int64_t B, int kernel);
int launch(hamk_system* s, KernelId k, int64_t B, void** args);
int rkf_flags(int row0, int inplace, int gsl_api);

}  // namespace hamk_host

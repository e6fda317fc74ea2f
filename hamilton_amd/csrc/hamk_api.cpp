// hamk_api.cpp -- C ABI of libhamk.so (include/hamk.h): system construction
// (tape -> hiprtc-specialised gfx950 module) and ensemble launches.
//
// Host-side counterpart of the reference's `System` record and its callers
// (Hamilton.hs:160-169, :201-254, :262-462): there a `System` is a bundle of
// Haskell closures re-entering `ad` on every call; here it is an opaque handle
// owning one compiled code object whose kernels evaluate the whole ensemble.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "hamk_host.h"

using namespace hamk_host;

static thread_local std::string g_last_error;
namespace hamk_host {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const std::string& last_error_text() { return g_last_error; }
const char* const kKernelNames[K__COUNT] = {"hamk_rk4_steps_k", "hamk_hameqs_k",  "hamk_coords_k",         "hamk_to_phase_k",
                                            "hamk_from_phase_k", "hamk_observe_k", "hamk_observe_config_k", "hamk_rkf45_k",
                                            "hamk_scribble_k"};
}  // namespace hamk_host

// ---- host-pointer staging -----------------------------------------------------
namespace {
struct Staged {
  void* dev = nullptr;
  void* host = nullptr;
  void* pinned = nullptr;      // non-null: `dev` aliases this pinned host block
  size_t bytes = 0;
  bool out = false;
};
constexpr size_t kPinArena = 256 << 10;    // bytes of pinned arena per handle
constexpr size_t kPinMaxBuf = 32 << 10;    // larger arrays go through device staging
class Stager {
 public:
  explicit Stager(hamk_system* s, int mem) : s_(s), host_(mem == HAMK_MEM_HOST) {}
  // returns the device pointer to use for `p` (nullptr stays nullptr)
  template <class T> int in(const T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), true, false, (void**)dev); }
  template <class T> int out(T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), false, true, (void**)dev); }
  template <class T> int inout(T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), true, true, (void**)dev); }
  // small read-only side input (evolveHam's time grid): pinned copy, or nullptr if it does not fit
  const double* side_input(const double* p, size_t count) {
    if (!host_) return nullptr;
    void* host = nullptr; void* dev = nullptr;
    if (!pin_alloc(count * sizeof(double), &host, &dev)) return nullptr;
    std::memcpy(host, p, count * sizeof(double));
    return (const double*)dev;
  }
  int finish() {
    if (!host_) return HAMK_OK;
    for (auto& b : bufs_)
      if (b.out && !b.pinned) HIP_TRY(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, s_->cur->stream));
    HIP_TRY(hipStreamSynchronize(s_->cur->stream));
    for (auto& b : bufs_)
      if (b.out && b.pinned) std::memcpy(b.host, b.pinned, b.bytes);
    return HAMK_OK;
  }

 private:
  int add(void* p, size_t bytes, bool copy_in, bool copy_out, void** dev) {
    if (!p || !host_ || bytes == 0) { *dev = p; return HAMK_OK; }
    Staged b; b.host = p; b.bytes = bytes; b.out = copy_out;
    if (bytes <= kPinMaxBuf && pin_alloc(bytes, &b.pinned, &b.dev)) {
      if (copy_in) std::memcpy(b.pinned, p, bytes);
      bufs_.push_back(b);
      *dev = b.dev;
      return HAMK_OK;
    }
    const size_t slot = nstaged_++;
    if (slot >= s_->cur->stage_buf.size()) { s_->cur->stage_buf.push_back(nullptr); s_->cur->stage_cap.push_back(0); }
    if (s_->cur->stage_cap[slot] < bytes) {
      HIP_TRY(hipStreamSynchronize(s_->cur->stream));          // nobody may still be using the old block
      if (s_->cur->stage_buf[slot]) hipFree(s_->cur->stage_buf[slot]);
      s_->cur->stage_buf[slot] = nullptr; s_->cur->stage_cap[slot] = 0;
      HIP_TRY(hipMalloc(&s_->cur->stage_buf[slot], bytes));
      s_->cur->stage_cap[slot] = bytes;
    }
    b.dev = s_->cur->stage_buf[slot];
    bufs_.push_back(b);
    if (copy_in) HIP_TRY(hipMemcpyAsync(b.dev, p, bytes, hipMemcpyHostToDevice, s_->cur->stream));
    *dev = b.dev;
    return HAMK_OK;
  }
  // bump allocation in the handle's pinned arena; every host-pointer call ends with a stream
  // synchronisation (finish), so the arena is free again when the next call starts
  bool pin_alloc(size_t bytes, void** host, void** dev) {
    if (s_->cur->pin_failed) return false;
    if (!s_->cur->pin) {
      static const bool off = [] { const char* e = test_env("HAMK_PINNED"); return e && e[0] == '0'; }();
      void* h = nullptr; void* d = nullptr;
      static const bool noncoh = [] { const char* e = test_env("HAMK_PINNED"); return e && e[0] == 'n'; }();   // test hook: the broken variant
      if (off || hipHostMalloc(&h, kPinArena, hipHostMallocMapped | (noncoh ? 0u : hipHostMallocCoherent)) != hipSuccess ||
          hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        if (h) hipHostFree(h);
        (void)hipGetLastError();
        s_->cur->pin_failed = true;
        return false;
      }
      s_->cur->pin = (char*)h; s_->cur->pin_dev = (char*)d;
    }
    const size_t at = (pin_used_ + 255) & ~(size_t)255;
    if (at + bytes > kPinArena) return false;
    pin_used_ = at + bytes;
    *host = s_->cur->pin + at; *dev = s_->cur->pin_dev + at;
    return true;
  }
  hamk_system* s_;
  bool host_;
  size_t pin_used_ = 0;
  size_t nstaged_ = 0;
  std::vector<Staged> bufs_;
};
}  // namespace


static int check_call(hamk_system* s, int64_t B, int32_t mem) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (B < 0) return fail(HAMK_ERR_INVALID, "negative ensemble size");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  return HAMK_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char* hamk_last_error(void) { return g_last_error.c_str(); }
const char* hamk_version(void) { return "hamk 0.2 (gfx950; hiprtc-specialised jets; RK4 + GSL-semantics RKF45, gsl_odeiv2 / gsl_odeiv)"; }

int hamk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hamk_set_device(int32_t device) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return HAMK_OK;
}

int hamk_get_device(int32_t* device) {
  if (!device) return fail(HAMK_ERR_INVALID, "null device");
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  *device = d;
  return HAMK_OK;
}

int hamk_device_malloc(void** ptr, int64_t bytes) {
  if (!ptr || bytes < 0) return fail(HAMK_ERR_INVALID, "hamk_device_malloc: null ptr / negative size");
  *ptr = nullptr;
  if (bytes == 0) return HAMK_OK;
  HIP_TRY(hipMalloc(ptr, (size_t)bytes));
  return HAMK_OK;
}

int hamk_device_free(void* ptr) {
  if (ptr) HIP_TRY(hipFree(ptr));
  return HAMK_OK;
}

int hamk_memcpy(void* dst, const void* src, int64_t bytes, int32_t kind) {
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(HAMK_ERR_INVALID, "hamk_memcpy: null pointer / negative size");
  if (bytes == 0) return HAMK_OK;
  hipMemcpyKind k;
  switch (kind) {
    case HAMK_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case HAMK_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case HAMK_COPY_D2D: k = hipMemcpyDefault; break;        // unified addressing: same device or a peer
    default: return fail(HAMK_ERR_INVALID, "hamk_memcpy: kind must be HAMK_COPY_H2D / D2H / D2D");
  }
  HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, k));
  return HAMK_OK;
}

int hamk_gather_batch(int32_t nparts, int32_t n, const int64_t* B_parts, const double* const* parts, double* out,
                      int32_t out_mem) {
  if (nparts < 0 || n <= 0 || (nparts > 0 && (!B_parts || !parts)))
    return fail(HAMK_ERR_INVALID, "hamk_gather_batch: bad nparts / n / null arrays");
  if (out_mem != HAMK_MEM_HOST && out_mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "out_mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  int64_t total = 0;
  for (int g = 0; g < nparts; ++g) {
    if (B_parts[g] < 0 || (B_parts[g] > 0 && !parts[g])) return fail(HAMK_ERR_INVALID, "hamk_gather_batch: negative size / null part");
    total += B_parts[g];
  }
  if (total == 0) return HAMK_OK;
  if (!out) return fail(HAMK_ERR_INVALID, "hamk_gather_batch: null out");
  int here = 0;
  HIP_TRY(hipGetDevice(&here));
  std::string peer_note;                                   // why a direct xGMI path could not be set up, if so
  int64_t at = 0;
  for (int g = 0; g < nparts; ++g) {
    const int64_t Bg = B_parts[g];
    if (Bg == 0) continue;
    if (out_mem == HAMK_MEM_DEVICE) {                       // direct xGMI DMA where the devices are peers
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, parts[g]) == hipSuccess && attr.device != here) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, here, attr.device) == hipSuccess && can) {
          hipError_t e = hipDeviceEnablePeerAccess(attr.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            peer_note = std::string(" [hipDeviceEnablePeerAccess(") + std::to_string(attr.device) + "): " + hipGetErrorString(e) + "]";
        } else {
          peer_note = " [device " + std::to_string(attr.device) + " is not a peer of device " + std::to_string(here) + "]";
        }
      }
      (void)hipGetLastError();                             // the copy below still works without peer access (staged by the runtime)
    }
    // row j of the part goes to columns [at, at + Bg) of row j of the output
    {
      hipError_t e = hipMemcpy2DAsync(out + at, (size_t)total * sizeof(double), parts[g], (size_t)Bg * sizeof(double),
                                      (size_t)Bg * sizeof(double), (size_t)n, hipMemcpyDefault, nullptr);
      if (e != hipSuccess) return fail(HAMK_ERR_HIP, std::string("hamk_gather_batch: copy of part ") + std::to_string(g) + ": " + hipGetErrorString(e) + peer_note);
    }
    at += Bg;
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
  return HAMK_OK;
}

void hamk_options_init(hamk_options* opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof *opt);
  opt->size = (uint32_t)sizeof *opt;
  opt->version = HAMK_OPTIONS_VERSION;
}

int hamk_system_create(int32_t m, int32_t n, const double* inertia, const hamk_op* f_ops, int32_t f_nops,
                       const int32_t* f_outs, const hamk_op* u_ops, int32_t u_nops, int32_t u_out, int32_t u_space,
                       hamk_system** out) {
  return hamk_system_create_ex(m, n, inertia, f_ops, f_nops, f_outs, u_ops, u_nops, u_out, u_space, nullptr, out);
}

int hamk_system_create_ex(int32_t m, int32_t n, const double* inertia, const hamk_op* f_ops, int32_t f_nops,
                          const int32_t* f_outs, const hamk_op* u_ops, int32_t u_nops, int32_t u_out, int32_t u_space,
                          const hamk_options* opt, hamk_system** out) {
  if (!out) return fail(HAMK_ERR_INVALID, "out is null");
  *out = nullptr;
  if (m <= 0 || n <= 0) return fail(HAMK_ERR_INVALID, "m and n must be positive");
  if (!inertia || !f_outs) return fail(HAMK_ERR_INVALID, "null inertia / f_outs");
  if (u_space != HAMK_U_GENERALIZED && u_space != HAMK_U_CARTESIAN) return fail(HAMK_ERR_INVALID, "bad u_space");
  if (n > 64 || m > 128)
    return fail(HAMK_ERR_UNSUPPORTED, "supported sizes: n <= 16 (one trajectory per lane), 17 <= n <= 64 with m <= 128 (wave-cooperative kernels)");
  std::string err = validate_tape(f_ops, f_nops, n, f_outs, m, "coordinate map");
  if (!err.empty()) return fail(HAMK_ERR_TAPE, err);
  const int nu = (u_space == HAMK_U_CARTESIAN) ? m : n;
  err = validate_tape(u_ops, u_nops, nu, &u_out, 1, "potential");
  if (!err.empty()) return fail(HAMK_ERR_TAPE, err);
  hamk_options o;
  hamk_options_init(&o);
  if (opt) {
    if (opt->size < 8 || opt->size > 4096) return fail(HAMK_ERR_INVALID, "hamk_options: size is not set (hamk_options_init)");
    if (opt->version != HAMK_OPTIONS_VERSION)
      return fail(HAMK_ERR_INVALID, "hamk_options: layout revision mismatch (built against another include/hamk.h, or hamk_options_init "
                                    "was not called): this library reads HAMK_OPTIONS_VERSION 0x484b0005");
    std::memcpy(&o, opt, std::min((size_t)opt->size, sizeof o));      // a caller built against an older header of THIS revision: the tail stays AUTO
    o.size = (uint32_t)sizeof o;
  }
  err = check_options(o, n);
  if (err.empty() && o.mapping == HAMK_MAP_QUAD)
    for (int k = 0; k < m; ++k)
      if (!(inertia[k] > 0.0))
        err = "unsupported: HAMK_MAP_QUAD factorises K = J^T M J without pivoting and needs every inertia positive; this system has "
              "inertia[" + std::to_string(k) + "] = " + std::to_string(inertia[k]) + " (leave the mapping to the library: the lane and "
              "wave-cooperative kernels pivot as the reference's `inv` does)";
  if (!err.empty()) return fail(err.rfind("unsupported:", 0) == 0 ? HAMK_ERR_UNSUPPORTED : HAMK_ERR_INVALID, "hamk_options: " + err);

  hamk_system* s = new hamk_system();
  s->opt = o;
  s->base.m = m; s->base.n = n; s->base.u_space = u_space;
  s->base.inertia.assign(inertia, inertia + m);
  s->base.f_ops.assign(f_ops, f_ops + f_nops);
  s->base.f_outs.assign(f_outs, f_outs + m);
  s->base.u_ops.assign(u_ops, u_ops + u_nops);
  s->base.u_out = u_out;
  s->gsl_api = o.gsl_api ? o.gsl_api : 2;
  if (o.gsl_api == HAMK_AUTO) if (const char* e = test_env("HAMK_GSL_API")) s->gsl_api = (e[0] == '1') ? 1 : 2;
  s->self_check_on = o.self_check != HAMK_OFF;
  if (o.self_check == HAMK_AUTO) if (const char* e = test_env("HAMK_SELFCHECK")) if (e[0] == '0') s->self_check_on = false;
  s->cache_on = o.cache != HAMK_OFF;
  s->ensemble_size = o.ensemble_size;
  s->max_substeps = o.max_substeps > 0 ? o.max_substeps : (1 << 24);
  if (o.max_substeps == HAMK_AUTO)                          // test suites: a kernel gone wrong must end, not spin through 16M attempts per lane
    if (const char* e = test_env("HAMK_MAX_SUBSTEPS")) { const long k = std::atol(e); if (k > 0 && k < (1L << 24)) s->max_substeps = (int)k; }
  // the specialisation a large ensemble uses is built now: a tape the kernels cannot be specialised for fails here
  Variant* v = nullptr;
  const int rc = variant_for(s, INT64_MAX, K_RK4, &v);
  if (rc != HAMK_OK) { hamk_system_destroy(s); return rc; }
  s->info = v;
  *out = s;
  return HAMK_OK;
}

void hamk_system_destroy(hamk_system* s) {
  if (!s) return;
  for (DevState* d : s->devs) { d->release(); delete d; }
  for (Variant* v : s->var) delete v;
  (void)hipGetLastError();
  delete s;
}

int hamk_system_dims(const hamk_system* s, int32_t* m, int32_t* n) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (m) *m = s->base.m;
  if (n) *n = s->base.n;
  return HAMK_OK;
}

int hamk_system_get_options(hamk_system* s, int64_t B, hamk_options* r) {
  if (!s || !r) return fail(HAMK_ERR_INVALID, "null system handle / options");
  Variant* v = nullptr;
  TRY(variant_for(s, B < 0 ? INT64_MAX : B, K_RK4, &v));
  hamk_options_init(r);
  const SystemDesc& d = v->desc;
  r->mapping = v->mapping;
  r->ad_mode = d.mode_h ? HAMK_AD_H : (d.mode_r ? HAMK_AD_R : HAMK_AD_D);
  r->rk4_body = d.rk4_stage_loop ? HAMK_BODY_STAGE_LOOP : HAMK_BODY_UNROLLED;
  r->rkf_body = d.rkf_stage_loop ? HAMK_BODY_STAGE_LOOP : HAMK_BODY_UNROLLED;
  r->trig = d.use_lut == 0 ? HAMK_TRIG_DIRECT : (d.use_lut == 1 ? HAMK_TRIG_TABLE : HAMK_TRIG_TABLE_ROTATE);
  r->gsl_api = s->gsl_api;
  r->self_check = s->self_check_on ? HAMK_ON : HAMK_OFF;
  const int bf = build_force(s);
  r->build = bf == 0 ? HAMK_BUILD_DEFAULT : (bf == 1 ? HAMK_BUILD_NOLICM : HAMK_AUTO);
  r->rk4_min_waves = d.rk4_min_waves;
  r->k_reassoc = d.k_reassoc ? HAMK_ON : HAMK_OFF;
  r->rk4_park = d.rk4_park ? HAMK_ON : HAMK_OFF;
  r->rkf_park = d.rkf_park ? HAMK_ON : HAMK_OFF;
  r->max_substeps = s->max_substeps;
  r->cache = s->cache_on ? HAMK_ON : HAMK_OFF;
  r->ensemble_size = s->ensemble_size;
  r->lanes_per_trajectory = v->mapping == HAMK_MAP_LANE ? 1 : (v->mapping == HAMK_MAP_QUAD ? 4 : (d.n <= 16 ? 16 : d.n <= 32 ? 32 : 64));
  return HAMK_OK;
}

int hamk_system_describe_batch(hamk_system* s, int64_t B) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  Variant* v = nullptr;
  TRY(variant_for(s, B < 0 ? INT64_MAX : B, K_RK4, &v));
  s->info = v;
  return HAMK_OK;
}

int hamk_system_set_ensemble_size(hamk_system* s, int64_t B_total) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (B_total < 0) return fail(HAMK_ERR_INVALID, "negative ensemble size");
  s->ensemble_size = B_total;
  return HAMK_OK;
}

int hamk_set_stream(hamk_system* s, void* hip_stream) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (current_device_state(s) != HAMK_OK) return HAMK_OK;  // no device to bind to: the first launch reports that
  s->cur->stream = (hipStream_t)hip_stream;
  return HAMK_OK;
}

int hamk_synchronize(hamk_system* s) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  TRY(current_device_state(s));
  HIP_TRY(hipStreamSynchronize(s->cur->stream));
  return HAMK_OK;
}

int hamk_system_set_gsl_api(hamk_system* s, int32_t api) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (api != 1 && api != 2) return fail(HAMK_ERR_INVALID, "gsl api must be 1 (gsl_odeiv) or 2 (gsl_odeiv2)");
  s->gsl_api = api;
  return HAMK_OK;
}

int32_t hamk_system_get_gsl_api(const hamk_system* s) { return s ? s->gsl_api : 0; }

int64_t hamk_system_code_object(const hamk_system* s, int32_t which, void* buf, int64_t cap) {
  if (!s || !s->info || (which != 0 && which != 1)) return 0;
  const std::vector<char>& c = which ? s->info->code2 : s->info->code;
  if (buf && cap >= (int64_t)c.size() && !c.empty()) std::memcpy(buf, c.data(), c.size());
  return (int64_t)c.size();
}

const char* hamk_system_source(const hamk_system* s) { return (s && s->info) ? s->info->source.c_str() : nullptr; }
const char* hamk_system_build_info(const hamk_system* s) { return (s && s->info) ? s->info->build_info.c_str() : ""; }
int64_t hamk_system_code_size(const hamk_system* s) { return (s && s->info) ? (int64_t)(s->info->code.size() + s->info->code2.size()) : 0; }
int64_t hamk_system_kernel_bytes(const hamk_system* s, const char* kernel_name) {
  if (!s || !s->info) return 0;
  if (kernel_name)
    for (int k = 0; k < K__COUNT; ++k)
      if (std::strcmp(kernel_name, kKernelNames[k]) == 0) return (int64_t)chosen_kernel_bytes(s->info, k);
  return (int64_t)kernel_code_bytes(s->info->code, kernel_name);
}

int hamk_coords_batch(hamk_system* s, int64_t B, const double* q, double* x, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !x) return fail(HAMK_ERR_INVALID, "null q / x");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_COORDS));
  Stager st(s, mem);
  const double* dq; double* dx;
  TRY(st.in(q, (size_t)s->base.n * B, (double**)&dq));
  TRY(st.out(x, (size_t)s->base.m * B, &dx));
  long long b = B;
  void* args[] = {&dq, &dx, &b};
  TRY(launch(s, K_COORDS, B, args));
  return st.finish();
}

int hamk_to_phase_batch(hamk_system* s, int64_t B, const double* q, const double* qd, double* p, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !qd || !p) return fail(HAMK_ERR_INVALID, "null q / qd / p");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_TO_PHASE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dqd; double* dp;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(qd, cnt, (double**)&dqd));
  TRY(st.out(p, cnt, &dp));
  long long b = B;
  void* args[] = {&dq, &dqd, &dp, &b};
  TRY(launch(s, K_TO_PHASE, B, args));
  return st.finish();
}

int hamk_from_phase_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* qd, int32_t* status,
                          int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p || !qd) return fail(HAMK_ERR_INVALID, "null q / p / qd");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_FROM_PHASE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dp; double* dqd; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(p, cnt, (double**)&dp));
  TRY(st.out(qd, cnt, &dqd));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&dq, &dp, &dqd, &b, &dst};
  TRY(launch(s, K_FROM_PHASE, B, args));
  return st.finish();
}

int hamk_observe_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* ke, double* pe, double* h,
                       int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q) return fail(HAMK_ERR_INVALID, "null q");
  if (!p && (ke || h)) return fail(HAMK_ERR_INVALID, "ke / h need momenta p");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_OBSERVE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dp; double *dke, *dpe, *dh; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(p, cnt, (double**)&dp));
  TRY(st.out(ke, (size_t)B, &dke));
  TRY(st.out(pe, (size_t)B, &dpe));
  TRY(st.out(h, (size_t)B, &dh));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&dq, &dp, &dke, &dpe, &dh, &b, &dst};
  TRY(launch(s, K_OBSERVE, B, args));
  return st.finish();
}

int hamk_observe_config_batch(hamk_system* s, int64_t B, const double* q, const double* qd, double* ke, double* lag,
                              int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !qd) return fail(HAMK_ERR_INVALID, "null q / qd");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_OBSERVE_CFG));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dqd; double *dke, *dlag;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(qd, cnt, (double**)&dqd));
  TRY(st.out(ke, (size_t)B, &dke));
  TRY(st.out(lag, (size_t)B, &dlag));
  long long b = B;
  void* args[] = {&dq, &dqd, &dke, &dlag, &b};
  TRY(launch(s, K_OBSERVE_CFG, B, args));
  return st.finish();
}

int hamk_hameqs_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* dq, double* dp,
                      int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p || !dq || !dp) return fail(HAMK_ERR_INVALID, "null q / p / dq / dp");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_HAMEQS));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *xq, *xp; double *xdq, *xdp; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&xq));
  TRY(st.in(p, cnt, (double**)&xp));
  TRY(st.out(dq, cnt, &xdq));
  TRY(st.out(dp, cnt, &xdp));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&xq, &xp, &xdq, &xdp, &b, &dst};
  TRY(launch(s, K_HAMEQS, B, args));
  return st.finish();
}

namespace { struct HamkBoxes { double q_lo[64], q_hi[64], qd_lo[64], qd_hi[64]; }; }     // = hamk_sample.hpp

// The sampler is the one JIT product that every other check of the library takes on trust (shard invariance, bench parity and the
// CPU tests' numpy sampler all rest on "index -> bits" being what hamk_sample.hpp says): on its first use per device, eight
// trajectories of a two-coordinate box are drawn and compared BIT FOR BIT with the same arithmetic on the host -- splitmix64 of
// (seed, global index, field), 53 bits to [0, 1), lo + (hi - lo) u in three separately rounded operations.
static uint64_t host_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double host_sample(uint64_t seed, uint64_t index, int field, double lo, double hi) {
  const uint64_t key = seed ^ (index * 0xD1342543DE82EF95ull);
  const uint64_t z = host_splitmix64(key + (uint64_t)field * 0x2545F4914F6CDD1Dull);
  const double u = (double)(z >> 11) * 0x1p-53;
  volatile double t = (hi - lo) * u;                        // (volatile: the product is rounded before the sum, whatever the host compiler contracts)
  return lo + t;
}
static int sample_self_check(DevState* d) {
  constexpr int n = 2, B = 8;
  const long long first = 123456789012ll;
  const unsigned long long seed = 0x5eedull;
  HamkBoxes bx;
  std::memset(&bx, 0, sizeof bx);
  bx.q_lo[0] = -1.25; bx.q_hi[0] = 2.5; bx.q_lo[1] = 0.5; bx.q_hi[1] = 0.75;
  bx.qd_lo[0] = -3.0; bx.qd_hi[0] = 3.0; bx.qd_lo[1] = 1e-3; bx.qd_hi[1] = 1e3;
  double* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, 2 * n * B * sizeof(double)));
  double *xq = dev, *xqd = dev + n * B;
  long long b = B, f0 = first;
  unsigned long long sd = seed;
  int nn = n;
  void* args[] = {&xq, &xqd, &b, &f0, &sd, &nn, &bx};
  double got[2 * n * B];
  hipError_t e = hipModuleLaunchKernel(d->sample_fn, 1, 1, 1, 256, 1, 1, 0, d->stream, args, nullptr);
  if (e == hipSuccess) e = hipMemcpyAsync(got, dev, sizeof got, hipMemcpyDeviceToHost, d->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
  hipFree(dev);
  if (e != hipSuccess) return fail(HAMK_ERR_HIP, std::string("sampler self-check: ") + hipGetErrorString(e));
  if (const char* f = test_env("HAMK_SELFCHECK_FAULT")) if (std::strstr(f, "sample")) got[3] = std::nextafter(got[3], 4.0);      // test hook: one bit off
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < B; ++i) {
      const double q = host_sample(seed, (uint64_t)(first + i), 2 * j, bx.q_lo[j], bx.q_hi[j]);
      const double v = host_sample(seed, (uint64_t)(first + i), 2 * j + 1, bx.qd_lo[j], bx.qd_hi[j]);
      if (std::memcmp(&q, &got[j * B + i], 8) != 0 || std::memcmp(&v, &got[n * B + j * B + i], 8) != 0)
        return fail(HAMK_ERR_COMPILE, "self-check failed: hamk_sample_k does not draw the bits of the host's evaluation of the same "
                                      "(seed, index, field) -> value map (miscompiled sampler module?)");
    }
  return HAMK_OK;
}

int hamk_sample_batch(hamk_system* s, int64_t B, int64_t first_index, uint64_t seed, const double* q_lo, const double* q_hi,
                      const double* qd_lo, const double* qd_hi, double* q, double* qd, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (first_index < 0) return fail(HAMK_ERR_INVALID, "negative first_index");
  if (first_index > INT64_MAX - B) return fail(HAMK_ERR_INVALID, "first_index + B overflows the trajectory index");
  if (B == 0) return HAMK_OK;                               // an empty shard (its arrays may be null) is a no-op
  if (!q_lo || !q_hi || !qd_lo || !qd_hi || !q || !qd) return fail(HAMK_ERR_INVALID, "null box / q / qd");
  const std::vector<char>* code = nullptr;
  TRY(sample_code(s->cache_on, &code));                     // (before the device is looked at: hiprtc needs no GPU, so a build
  TRY(current_device_state(s));                             //  machine without one still proves the kernel compiles for gfx950)
  DevState* d = s->cur;
  if (!d->sample_fn) {
    HIP_TRY(hipModuleLoadData(&d->sample_module, code->data()));
    HIP_TRY(hipModuleGetFunction(&d->sample_fn, d->sample_module, "hamk_sample_k"));
    if (s->self_check_on) {                                 // first use on this device: a few draws against the host's own evaluation
      const int rc0 = sample_self_check(d);
      if (rc0 != HAMK_OK) { hipModuleUnload(d->sample_module); d->sample_module = nullptr; d->sample_fn = nullptr; return rc0; }
    }
  }
  HamkBoxes bx;
  std::memset(&bx, 0, sizeof bx);
  const int n = s->base.n;
  for (int j = 0; j < n; ++j) { bx.q_lo[j] = q_lo[j]; bx.q_hi[j] = q_hi[j]; bx.qd_lo[j] = qd_lo[j]; bx.qd_hi[j] = qd_hi[j]; }
  Stager st(s, mem);
  const size_t cnt = (size_t)n * B;
  double *xq, *xqd;
  TRY(st.out(q, cnt, &xq));
  TRY(st.out(qd, cnt, &xqd));
  long long b = B, first = first_index;
  unsigned long long sd = seed;
  int nn = n;
  void* args[] = {&xq, &xqd, &b, &first, &sd, &nn, &bx};
  const int64_t grid = (B + 255) / 256;
  if (grid > 0x7fffffffLL) return fail(HAMK_ERR_INVALID, "ensemble too large for one launch");
  HIP_TRY(hipModuleLaunchKernel(d->sample_fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, d->stream, args, nullptr));
  return st.finish();
}

// ---- a large ensemble in HOST arrays: its transfers under its own kernels ----------------------------------------------
// Copy in -> kernel -> copy out back to back leaves the GPU idle for the PCIe time (C2's 64 MiB both ways: 1.3 ms around
// an 11 ms launch of 1000 steps, around a 1.3 ms launch of 100).  Trajectories do not interact, so the ensemble is stepped
// in pieces of piece_size() trajectories: piece c + 1 travels in and piece c - 1 travels out on the copy stream while piece c
// steps on the handle's stream; only the first copy in and the last copy out are exposed.  Every piece runs the variant
// bound for the WHOLE ensemble (bind_device saw B), one trajectory's arithmetic does not depend on its neighbours', so
// the results are bitwise those of the single launch (tests/test_gpu_configs.py).  The host arrays are pageable, where
// hipMemcpyAsync holds the calling thread for the length of the copy: the order of the calls below is the overlap.
// Piece size, measured on MI355X (C2, B = 2^20, host-array call in ms at 32 / 100 / 1000 steps; one launch 1.75 / 2.66 / 13.6):
// 2^17 4.28 / 4.32 / 13.2, 2^18 1.93 / 2.16 / 13.1, 2^19 1.53 / 2.12 / 12.8 (profiles/r05_host_pieces_sweep.txt).  A pageable
// copy costs tens of microseconds before its first byte moves, and a piece of 2^18 trajectories runs its kernel at 0.91 of
// the full launch's rate (2^19: 0.97): few, large pieces.
static int64_t piece_size() {
  static const int64_t v = [] {
    const char* e = test_env("HAMK_HOST_PIECE");                  // (test override: another size, a multiple of 256)
    const long long x = e ? std::atoll(e) : 0;
    return (int64_t)((x >= 256 && x % 256 == 0) ? x : (1 << 19));
  }();
  return v;
}
static bool host_pieces_on() {
  static const bool off = [] { const char* e = test_env("HAMK_HOST_PIECES"); return e && e[0] == '0'; }();
  return !off;
}

static int stage_slot(hamk_system* s, size_t slot, size_t bytes, void** dev) {
  DevState* d = s->cur;
  while (slot >= d->stage_buf.size()) { d->stage_buf.push_back(nullptr); d->stage_cap.push_back(0); }
  if (d->stage_cap[slot] < bytes) {
    HIP_TRY(hipStreamSynchronize(d->stream));                  // nobody may still be using the old block
    if (d->stage_buf[slot]) hipFree(d->stage_buf[slot]);
    d->stage_buf[slot] = nullptr; d->stage_cap[slot] = 0;
    HIP_TRY(hipMalloc(&d->stage_buf[slot], bytes));
    d->stage_cap[slot] = bytes;
  }
  *dev = d->stage_buf[slot];
  return HAMK_OK;
}

static int rk4_steps_host_pieces(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t nsteps, double drift_tol,
                                 int32_t* status) {
  DevState* d = s->cur;
  const int n = s->base.n;
  const int64_t kPiece = piece_size();
  const int npieces = (int)((B + kPiece - 1) / kPiece);
  if (!d->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
  while ((int)d->piece_events.size() < 2 * npieces + 1) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    d->piece_events.push_back(e);
  }
  void *vq = nullptr, *vp = nullptr, *vs = nullptr;
  TRY(stage_slot(s, 0, sizeof(double) * (size_t)n * B, &vq));
  TRY(stage_slot(s, 1, sizeof(double) * (size_t)n * B, &vp));
  if (status) TRY(stage_slot(s, 2, sizeof(int32_t) * (size_t)B, &vs));
  double *dq = (double*)vq, *dp = (double*)vp;
  int32_t* dst = (int32_t*)vs;
  hipStream_t S = d->stream, C = d->copy_stream;
  // what the handle's stream was doing with the staging blocks comes first
  hipEvent_t start = d->piece_events[2 * npieces];
  HIP_TRY(hipEventRecord(start, S));
  HIP_TRY(hipStreamWaitEvent(C, start, 0));
  // piece c: trajectories [c kPiece, c kPiece + bc) -- on the device a compact [n][bc] block at offset n c kPiece
  auto count = [&](int c) { return std::min<int64_t>(kPiece, B - (int64_t)c * kPiece); };
  auto copy_in = [&](int c) -> int {
    const int64_t at = (int64_t)c * kPiece, bc = count(c);
    for (int j = 0; j < n; ++j) {
      HIP_TRY(hipMemcpyAsync(dq + (size_t)n * at + (size_t)j * bc, q + (size_t)j * B + at, sizeof(double) * bc, hipMemcpyHostToDevice, C));
      HIP_TRY(hipMemcpyAsync(dp + (size_t)n * at + (size_t)j * bc, p + (size_t)j * B + at, sizeof(double) * bc, hipMemcpyHostToDevice, C));
    }
    HIP_TRY(hipEventRecord(d->piece_events[2 * c], C));
    return HAMK_OK;
  };
  auto step = [&](int c) -> int {
    const int64_t at = (int64_t)c * kPiece;
    double *xq = dq + (size_t)n * at, *xp = dp + (size_t)n * at;
    int32_t* xs = dst ? dst + at : nullptr;
    long long b = count(c);
    int ns = nsteps;
    void* args[] = {&xq, &xp, &b, &dt, &ns, &drift_tol, &xs};
    HIP_TRY(hipStreamWaitEvent(S, d->piece_events[2 * c], 0));
    TRY(launch(s, K_RK4, b, args));
    HIP_TRY(hipEventRecord(d->piece_events[2 * c + 1], S));
    return HAMK_OK;
  };
  auto copy_out = [&](int c) -> int {
    const int64_t at = (int64_t)c * kPiece, bc = count(c);
    HIP_TRY(hipStreamWaitEvent(C, d->piece_events[2 * c + 1], 0));
    for (int j = 0; j < n; ++j) {
      HIP_TRY(hipMemcpyAsync(q + (size_t)j * B + at, dq + (size_t)n * at + (size_t)j * bc, sizeof(double) * bc, hipMemcpyDeviceToHost, C));
      HIP_TRY(hipMemcpyAsync(p + (size_t)j * B + at, dp + (size_t)n * at + (size_t)j * bc, sizeof(double) * bc, hipMemcpyDeviceToHost, C));
    }
    if (status) HIP_TRY(hipMemcpyAsync(status + at, dst + at, sizeof(int32_t) * bc, hipMemcpyDeviceToHost, C));
    return HAMK_OK;
  };
  int rc = copy_in(0);
  if (rc == HAMK_OK) rc = step(0);
  for (int c = 1; c < npieces && rc == HAMK_OK; ++c) {
    rc = copy_in(c);
    if (rc == HAMK_OK) rc = step(c);
    if (rc == HAMK_OK) rc = copy_out(c - 1);
  }
  if (rc == HAMK_OK) rc = copy_out(npieces - 1);
  // the call returns with both streams drained, on the failure paths too: the host arrays are the caller's again
  const hipError_t e1 = hipStreamSynchronize(C), e2 = hipStreamSynchronize(S);
  if (rc != HAMK_OK) return rc;
  HIP_TRY(e1);
  HIP_TRY(e2);
  return HAMK_OK;
}

int hamk_rk4_steps_checked(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t nsteps, double drift_tol,
                           int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p) return fail(HAMK_ERR_INVALID, "null q / p");
  if (nsteps < 0) return fail(HAMK_ERR_INVALID, "negative nsteps");
  if (drift_tol != drift_tol) return fail(HAMK_ERR_INVALID, "drift_tol is NaN");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RK4));
  if (mem == HAMK_MEM_HOST && B >= 2 * piece_size() && (int64_t)nsteps * s->base.n >= 64 && host_pieces_on()) return rk4_steps_host_pieces(s, B, q, p, dt, nsteps, drift_tol, status);
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  double *xq, *xp; int32_t* dst;
  TRY(st.inout(q, cnt, &xq));
  TRY(st.inout(p, cnt, &xp));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  int ns = nsteps;
  void* args[] = {&xq, &xp, &b, &dt, &ns, &drift_tol, &dst};
  TRY(launch(s, K_RK4, B, args));
  return st.finish();
}

int hamk_rk4_steps(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t nsteps, int32_t* status,
                   int32_t mem) {
  return hamk_rk4_steps_checked(s, B, q, p, dt, nsteps, 0.0, status, mem);
}

static int upload_times(hamk_system* s, int32_t nt, const double* ts) {
  if ((size_t)nt > s->cur->d_ts_cap) {
    if (s->cur->d_ts) hipFree(s->cur->d_ts);
    s->cur->d_ts = nullptr; s->cur->d_ts_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->cur->d_ts, sizeof(double) * nt));
    s->cur->d_ts_cap = nt;
  }
  // the previous launch may still be reading d_ts / h_ts: order behind it on the stream
  HIP_TRY(hipStreamSynchronize(s->cur->stream));
  s->cur->h_ts.assign(ts, ts + nt);
  HIP_TRY(hipMemcpyAsync(s->cur->d_ts, s->cur->h_ts.data(), sizeof(double) * nt, hipMemcpyHostToDevice, s->cur->stream));
  return HAMK_OK;
}

static const double kRefEps = 1.49012e-08;   // Hamilton.hs:448
int hamk_evolve_ham_batch(hamk_system* s, int64_t B, const double* q0, const double* p0, int32_t nt, const double* ts,
                          double* qout, double* pout, double h0, double eps_abs, double eps_rel, int32_t* status,
                          int32_t* nsub, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q0 || !p0 || !qout || !pout || !ts) return fail(HAMK_ERR_INVALID, "null q0 / p0 / ts / qout / pout");
  if (nt < 2) return fail(HAMK_ERR_INVALID, "evolveHam needs at least two times (2 <= s, Hamilton.hs:435)");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RKF45));
  if (!(h0 > 0.0)) h0 = (ts[1] - ts[0]) / 100.0;          // Hamilton.hs:447
  if (!(eps_abs > 0.0)) eps_abs = kRefEps;
  if (!(eps_rel > 0.0)) eps_rel = kRefEps;
  if (s->gsl_api == 2) {
    // gsl_odeiv2_driver_apply: the direction is the sign of the initial step (h > 0 ? +1 : -1) and a
    // target time on the wrong side of t is GSL_EINVAL ("integration limits and/or step direction not
    // consistent"); t equals the previous grid time exactly whenever a trajectory gets this far
    const double sgn = h0 > 0.0 ? 1.0 : -1.0;
    for (int32_t r = 1; r < nt; ++r)
      if (sgn * (ts[r] - ts[r - 1]) < 0.0)
        return fail(HAMK_ERR_INVALID, "evolveHam (gsl_odeiv2 semantics): integration limits and/or step direction not consistent "
                                      "(the time grid must be monotone in the direction of ts[1] - ts[0])");
  }
  Stager st(s, mem);
  // time grid: two times travel as kernel arguments; a longer grid as a pinned side input of a
  // small host-pointer call, else through the handle's device scratch
  double ts0 = ts[0], ts1 = ts[1];
  const double* dts = nullptr;
  if (nt > 2) {
    dts = (size_t)nt * sizeof(double) <= kPinMaxBuf ? st.side_input(ts, (size_t)nt) : nullptr;
    if (!dts) { TRY(upload_times(s, nt, ts)); dts = s->cur->d_ts; }
  }
  const size_t cnt = (size_t)s->base.n * B;
  const double *xq, *xp; double *xqo, *xpo; int32_t *dst, *dns;
  TRY(st.in(q0, cnt, (double**)&xq));
  TRY(st.in(p0, cnt, (double**)&xp));
  TRY(st.out(qout, cnt * nt, &xqo));
  TRY(st.out(pout, cnt * nt, &xpo));
  TRY(st.out(status, (size_t)B, &dst));
  TRY(st.out(nsub, (size_t)B, &dns));
  long long b = B;
  int nt_ = nt, flags = rkf_flags(0, 0, s->gsl_api), max_sub = s->max_substeps;
  int ncalls = 1, it_every = 0;
  void* args[] = {&xq, &xp, &xqo, &xpo, &b, &nt_, &dts, &ts0, &ts1, &h0, &eps_abs, &eps_rel, &flags, &max_sub, &dst, &dns, &ncalls, &it_every};
  TRY(launch(s, K_RKF45, B, args));
  return st.finish();
}

int hamk_step_ham_iterate(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t ncalls, int32_t out_every,
                          double* qout, double* pout, int32_t* status, int32_t* nsub, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p) return fail(HAMK_ERR_INVALID, "null q / p");
  if (ncalls < 0 || out_every < 0) return fail(HAMK_ERR_INVALID, "negative ncalls / out_every");
  if (out_every > 0 && (!qout || !pout)) return fail(HAMK_ERR_INVALID, "out_every > 0 needs qout / pout");
  if (B == 0 || ncalls == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RKF45));
  double ts0 = 0.0, ts1 = dt;                               // evolveHam over (0, r), Hamilton.hs:401
  double h0 = dt / 100.0, eps_abs = kRefEps, eps_rel = kRefEps;
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const size_t rows = out_every > 0 ? (size_t)(ncalls / out_every) : 0;
  double *xq, *xp, *xqo = nullptr, *xpo = nullptr; int32_t *dst, *dns;
  TRY(st.inout(q, cnt, &xq));
  TRY(st.inout(p, cnt, &xp));
  if (rows > 0) { TRY(st.out(qout, cnt * rows, &xqo)); TRY(st.out(pout, cnt * rows, &xpo)); }
  TRY(st.out(status, (size_t)B, &dst));
  TRY(st.out(nsub, (size_t)B, &dns));
  long long b = B;
  // in place on (q, p); the kernel's qout/pout receive the frames (hamk_device.hpp rkf45_body, flags)
  int nt_ = 2, flags = rkf_flags(1, 2, s->gsl_api), max_sub = s->max_substeps;
  int nc = ncalls, every = rows > 0 ? out_every : 0;
  const double* dts = nullptr;
  const double *cq = xq, *cp = xp;
  void* args[] = {&cq, &cp, &xqo, &xpo, &b, &nt_, &dts, &ts0, &ts1, &h0, &eps_abs, &eps_rel, &flags, &max_sub, &dst, &dns, &nc, &every};
  TRY(launch(s, K_RKF45, B, args));
  return st.finish();
}

int hamk_step_ham_batch(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t* status, int32_t* nsub,
                        int32_t mem) {
  return hamk_step_ham_iterate(s, B, q, p, dt, 1, 0, nullptr, nullptr, status, nsub, mem);
}


// ---- ensemble checkpoint: flat binary dump / restore of the SoA state ----------
namespace {
struct CkHeader {              // 64 bytes, little endian
  char magic[8];               // "HAMKCKP1"
  int32_t n, version;
  int64_t B, steps_done;
  uint64_t seed;
  double t;
  unsigned char reserved[16];
};
static_assert(sizeof(CkHeader) == 64, "checkpoint header layout");
constexpr char kCkMagic[8] = {'H', 'A', 'M', 'K', 'C', 'K', 'P', '1'};
constexpr size_t kCkChunk = 8u << 20;   // staging chunk for device-resident state
}  // namespace

int hamk_checkpoint_write(const char* path, int32_t n, int64_t B, const double* q, const double* p, int32_t mem,
                          int64_t steps_done, uint64_t seed, double t) {
  if (!path || n <= 0 || B < 0 || (B > 0 && (!q || !p))) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_write: bad path / n / B / null state");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  CkHeader h;
  std::memset(&h, 0, sizeof h);
  std::memcpy(h.magic, kCkMagic, 8);
  h.n = n; h.version = 1; h.B = B; h.steps_done = steps_done; h.seed = seed; h.t = t;
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  std::FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_write: cannot create ") + tmp);
  Sha256 sha;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  sha.update(&h, sizeof h);
  std::vector<char> stage;
  const size_t bytes = (size_t)n * (size_t)B * sizeof(double);
  for (const double* arr : {q, p}) {
    for (size_t off = 0; off < bytes && ok; off += kCkChunk) {
      const size_t k = std::min(kCkChunk, bytes - off);
      const char* src = (const char*)arr + off;
      if (mem == HAMK_MEM_DEVICE) {
        stage.resize(k);
        hipError_t e = hipMemcpy(stage.data(), src, k, hipMemcpyDeviceToHost);    // synchronises with the null stream
        if (e != hipSuccess) { std::fclose(f); std::remove(tmp.c_str()); return fail(HAMK_ERR_HIP, std::string("hamk_checkpoint_write: ") + hipGetErrorString(e)); }
        src = stage.data();
      }
      ok = std::fwrite(src, 1, k, f) == k;
      sha.update(src, k);
    }
  }
  unsigned char dg[32];
  sha.finish(dg);
  ok = ok && std::fwrite(dg, 1, 32, f) == 32;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_write: write to ") + path + " failed"); }
  return HAMK_OK;
}

static int read_ck_header(std::FILE* f, const char* path, CkHeader* h) {
  if (std::fread(h, sizeof *h, 1, f) != 1 || std::memcmp(h->magic, kCkMagic, 8) != 0 || h->version != 1 || h->n <= 0 || h->B < 0)
    return fail(HAMK_ERR_INVALID, std::string(path) + " is not a hamk checkpoint");
  // the header must describe THIS file before anything is sized from it: 64 + 2 n B 8 + 32 bytes, without overflow
  // (a corrupt or crafted header would otherwise ask for an arbitrary allocation before the digest is looked at)
  struct stat st;
  if (::fstat(fileno(f), &st) != 0) return fail(HAMK_ERR_INVALID, std::string(path) + ": cannot stat");
  const uint64_t n = (uint64_t)h->n, B = (uint64_t)h->B, lim = (uint64_t)1 << 60;
  if (n > 64 || (B != 0 && n * 16 > lim / B) || (uint64_t)st.st_size != sizeof(CkHeader) + 2 * n * B * 8 + 32)
    return fail(HAMK_ERR_INVALID, std::string(path) + ": header does not match the file size (truncated or corrupted checkpoint)");
  return HAMK_OK;
}

int hamk_checkpoint_info(const char* path, int32_t* n, int64_t* B, int64_t* steps_done, uint64_t* seed, double* t) {
  if (!path) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_info: null path");
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_info: cannot open ") + path);
  CkHeader h;
  const int rc = read_ck_header(f, path, &h);
  std::fclose(f);
  if (rc != HAMK_OK) return rc;
  if (n) *n = h.n;
  if (B) *B = h.B;
  if (steps_done) *steps_done = h.steps_done;
  if (seed) *seed = h.seed;
  if (t) *t = h.t;
  return HAMK_OK;
}

int hamk_checkpoint_read(const char* path, int32_t n, int64_t B, double* q, double* p, int32_t mem) {
  if (!path || (B > 0 && (!q || !p))) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_read: null path / state");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_read: cannot open ") + path);
  CkHeader h;
  int rc = read_ck_header(f, path, &h);
  if (rc == HAMK_OK && (h.n != n || h.B != B)) rc = fail(HAMK_ERR_INVALID, "hamk_checkpoint_read: the file holds a different ensemble (n, B)");
  if (rc != HAMK_OK) { std::fclose(f); return rc; }
  Sha256 sha;
  sha.update(&h, sizeof h);
  const size_t bytes = (size_t)n * (size_t)B * sizeof(double);
  // the digest is checked BEFORE anything reaches the caller's arrays (host or device)
  std::vector<char> all(2 * bytes);
  bool ok = bytes == 0 || std::fread(all.data(), 1, 2 * bytes, f) == 2 * bytes;
  unsigned char want[32], got[32];
  ok = ok && std::fread(want, 1, 32, f) == 32;
  std::fclose(f);
  if (ok) { sha.update(all.data(), 2 * bytes); sha.finish(got); ok = std::memcmp(want, got, 32) == 0; }
  if (!ok) return fail(HAMK_ERR_INVALID, std::string(path) + ": truncated or corrupted checkpoint (digest mismatch)");
  if (bytes == 0) return HAMK_OK;
  if (mem == HAMK_MEM_HOST) {
    std::memcpy(q, all.data(), bytes);
    std::memcpy(p, all.data() + bytes, bytes);
  } else {
    HIP_TRY(hipMemcpy(q, all.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p, all.data() + bytes, bytes, hipMemcpyHostToDevice));
  }
  return HAMK_OK;
}

}  // extern "C"

// hamk_dispatch.cpp -- options -> specialisations, device binding, launches, first-use self-check (host side of
// libhamk.so, see hamk_host.h).
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

namespace hamk_host {
const char* test_env(const char* name) {
  const char* on = std::getenv("HAMK_TEST_OVERRIDES");
  return (on && on[0] == '1' && on[1] == 0) ? std::getenv(name) : nullptr;
}

// hamk_options::build, else HAMK_NOLICM (tests), else per kernel
int build_force(const hamk_system* s) {                      // hamk_options::build, else HAMK_NOLICM (tests), else per kernel
  if (s->opt.build == HAMK_BUILD_DEFAULT) return 0;
  if (s->opt.build == HAMK_BUILD_NOLICM) return 1;
  if (const char* e = test_env("HAMK_NOLICM")) return (e[0] == '1') ? 1 : (e[0] == '0' ? 0 : -1);
  return -1;
}

int launch(hamk_system* s, KernelId k, int64_t B, void** args);
// the flags argument of hamk_rkf45_k (hamk_device.hpp rkf45_body)
int rkf_flags(int row0, int inplace, int gsl_api) { return (row0 & 1) | ((inplace & 3) << 8) | ((gsl_api & 3) << 16); }
static int load_modules(hamk_system* s);

// ---------------------------------------------------------------------------
// First-use self-check of the stepping kernels against the (small, separately compiled) hamEqs
// kernel: one RK4 step of the fused kernel must equal four hamEqs launches combined on the host,
// one accepted RKF45 sub-step must equal its six stage evaluations combined on the host.  A JIT
// product cannot take the code generator's word for it: on this toolchain one large unrolled
// stepping kernel was observed to be silently wrong.  On a mismatch the
// module is rebuilt once with the stage-loop bodies; if that does not help, the system is refused.
// HAMK_SELFCHECK=0 skips it.
// ---------------------------------------------------------------------------
static const double kRefEpsilon = 1.49012e-08;   // Hamilton.hs:448
static thread_local std::string g_selfcheck_detail;
static int self_check_once(hamk_system* s, bool* rk4_ok, bool* rkf_ok) {
  g_selfcheck_detail.clear();
  const int n = s->base.n;
  const int64_t B = 64;
  const size_t cnt = (size_t)n * B;
  std::vector<double> q(cnt), p(cnt), k(2 * cnt), acc(2 * cnt), yt(2 * cnt);
  for (int j = 0; j < n; ++j)
    for (int64_t i = 0; i < B; ++i) {
      q[(size_t)j * B + i] = 0.31 + 0.07 * j + 0.011 * (double)i;
      p[(size_t)j * B + i] = 0.23 - 0.05 * j + 0.007 * (double)i;
    }
  // device scratch of the check, released on every path out of this function
  struct Scratch {
    void* p[5] = {};
    ~Scratch() { for (void* x : p) if (x) hipFree(x); }
  } scratch;
  HIP_TRY(hipMalloc(&scratch.p[0], cnt * 8)); HIP_TRY(hipMalloc(&scratch.p[1], cnt * 8));
  HIP_TRY(hipMalloc(&scratch.p[2], cnt * 8)); HIP_TRY(hipMalloc(&scratch.p[3], cnt * 8));
  HIP_TRY(hipMalloc(&scratch.p[4], B * 4));
  double *d_q = (double*)scratch.p[0], *d_p = (double*)scratch.p[1], *d_dq = (double*)scratch.p[2], *d_dp = (double*)scratch.p[3];
  int32_t* d_st = (int32_t*)scratch.p[4];
  long long b = B;
  bool flagged = false;
  auto rhs = [&](const std::vector<double>& y, std::vector<double>& out) -> int {   // out = hamEqs(y), y = [q; p]
    HIP_TRY(hipMemcpy(d_q, y.data(), cnt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p, y.data() + cnt, cnt * 8, hipMemcpyHostToDevice));
    const double *cq = d_q, *cp = d_p; int32_t* st = d_st;
    void* args[] = {&cq, &cp, &d_dq, &d_dp, &b, &st};
    int rc = launch(s, K_HAMEQS, B, args);
    if (rc != HAMK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->cur->stream));
    HIP_TRY(hipMemcpy(out.data(), d_dq, cnt * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out.data() + cnt, d_dp, cnt * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> hst((size_t)B);
    HIP_TRY(hipMemcpy(hst.data(), d_st, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int32_t v : hst) if (v != 0) flagged = true;    // singular / non-finite at the test points: cannot judge
    return HAMK_OK;
  };
  auto close_enough = [&](const std::vector<double>& a, const std::vector<double>& ref, bool* usable, const char* what) {
    double worst = 0.0; *usable = true;
    for (size_t i = 0; i < 2 * cnt; ++i) {
      if (!std::isfinite(ref[i])) { *usable = false; return true; }     // test points outside the system's domain: cannot judge
      const double e = std::fabs(a[i] - ref[i]) / std::fmax(1.0, std::fabs(ref[i]));
      if (!(e <= worst)) worst = e;
    }
    if (!(worst <= 1e-9)) {
      char msg[160];
      std::snprintf(msg, sizeof msg, "%s: worst deviation %.3g from the same step built from hamEqs launches", what, worst);
      g_selfcheck_detail = msg;
    }
    return worst <= 1e-9;
  };
  std::vector<double> y0(2 * cnt), got(2 * cnt), ref(2 * cnt);
  std::copy(q.begin(), q.end(), y0.begin()); std::copy(p.begin(), p.end(), y0.begin() + cnt);
  int rc = HAMK_OK;
  bool usable = true;
  // ---- RK4: one step of dt -------------------------------------------------------------------
  // dt = 1e-3 unless the right-hand side at the test points is large (a 64-link chain with these
  // momenta turns at 1e4 rad/s): the two ways of computing one step agree to roundoff only while the
  // step is a small perturbation of the state, so dt keeps the largest increment at 0.05.
  double dt = 1e-3;
  {
    rc = rhs(y0, k);
    double kmax = 0.0;
    for (size_t i = 0; i < 2 * cnt; ++i) if (std::isfinite(k[i]) && std::fabs(k[i]) > kmax) kmax = std::fabs(k[i]);
    if (kmax * dt > 0.05) dt = 0.05 / kmax;
  }
  if (rc == HAMK_OK) {
    const double a[4] = {0.0, 0.5 * dt, 0.5 * dt, dt}, w[4] = {dt / 6, dt / 3, dt / 3, dt / 6};
    ref = y0;
    std::fill(k.begin(), k.end(), 0.0);
    for (int sg = 0; sg < 4 && rc == HAMK_OK; ++sg) {
      for (size_t i = 0; i < 2 * cnt; ++i) yt[i] = y0[i] + a[sg] * k[i];
      rc = rhs(yt, k);
      for (size_t i = 0; i < 2 * cnt; ++i) ref[i] += w[sg] * k[i];
    }
    if (rc == HAMK_OK) {
      hipMemcpy(d_q, q.data(), cnt * 8, hipMemcpyHostToDevice); hipMemcpy(d_p, p.data(), cnt * 8, hipMemcpyHostToDevice);
      double ddt = dt, no_drift = 0.0; int ns = 1; int32_t* st = d_st;
      void* args[] = {&d_q, &d_p, &b, &ddt, &ns, &no_drift, &st};
      rc = launch(s, K_RK4, B, args);
      if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RK4 kernel failed");
      hipMemcpy(got.data(), d_q, cnt * 8, hipMemcpyDeviceToHost); hipMemcpy(got.data() + cnt, d_p, cnt * 8, hipMemcpyDeviceToHost);
      *rk4_ok = close_enough(got, ref, &usable, "one RK4 step");
    }
  }
  // ---- RKF45: one accepted sub-step (h = dt, huge tolerances, t: 0 -> dt) ---------------------
  const bool has_rkf = s->curv->has[K_RKF45];              // (the quad module leaves the adaptive stepper to the wave module)
  if (rc == HAMK_OK && usable && has_rkf) {
    static const double A[5][5] = {{1.0 / 4, 0, 0, 0, 0},
                                   {3.0 / 32, 9.0 / 32, 0, 0, 0},
                                   {1932.0 / 2197, -7200.0 / 2197, 7296.0 / 2197, 0, 0},
                                   {8341.0 / 4104, -32832.0 / 4104, 29440.0 / 4104, -845.0 / 4104, 0},
                                   {-6080.0 / 20520, 41040.0 / 20520, -28352.0 / 20520, 9295.0 / 20520, -5643.0 / 20520}};
    static const double C[6] = {902880.0 / 7618050, 0, 3953664.0 / 7618050, 3855735.0 / 7618050, -1371249.0 / 7618050, 277020.0 / 7618050};
    std::vector<std::vector<double>> ks(6, std::vector<double>(2 * cnt));
    rc = rhs(y0, ks[0]);
    for (int sg = 1; sg < 6 && rc == HAMK_OK; ++sg) {
      for (size_t i = 0; i < 2 * cnt; ++i) {
        double t = 0.0;
        for (int j2 = 0; j2 < sg; ++j2) t += A[sg - 1][j2] * ks[j2][i];
        yt[i] = y0[i] + dt * t;
      }
      rc = rhs(yt, ks[sg]);
    }
    if (rc == HAMK_OK) {
      for (size_t i = 0; i < 2 * cnt; ++i) {
        double t = 0.0;
        for (int j2 = 0; j2 < 6; ++j2) t += C[j2] * ks[j2][i];
        ref[i] = y0[i] + dt * t;
      }
      hipMemcpy(d_q, q.data(), cnt * 8, hipMemcpyHostToDevice); hipMemcpy(d_p, p.data(), cnt * 8, hipMemcpyHostToDevice);
      double h0 = dt, ea = 1e30, er = 1e30, t0 = 0.0, t1 = dt;
      int nt = 2, flags = rkf_flags(1, 1, s->gsl_api), max_sub = 8;
      const double *cq = d_q, *cp = d_p, *cts = nullptr; int32_t* st = d_st; int32_t* ns = nullptr;
      int ncalls = 1, it_every = 0;
      void* args[] = {&cq, &cp, &d_q, &d_p, &b, &nt, &cts, &t0, &t1, &h0, &ea, &er, &flags, &max_sub, &st, &ns, &ncalls, &it_every};
      rc = launch(s, K_RKF45, B, args);
      if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RKF45 kernel failed");
      hipMemcpy(got.data(), d_q, cnt * 8, hipMemcpyDeviceToHost); hipMemcpy(got.data() + cnt, d_p, cnt * 8, hipMemcpyDeviceToHost);
      *rkf_ok = close_enough(got, ref, &usable, "one accepted RKF45 sub-step");
    }
  }
  if (!usable || flagged) { *rk4_ok = true; *rkf_ok = true; }
  // ---- adaptive stepper end to end ---------------------------------------------------------------
  // stepHam(T) with the reference's tolerances and step-size control on 256 trajectories, TWICE:
  // the two runs must agree bit for bit (lanes are independent: a difference is a broken kernel, and
  // exactly that was seen once -- an unrolled RKF45 body whose results changed from run to run),
  // and both must agree with 64 fixed RK4 steps of T/64 (the kernel checked above) to well within
  // what the controller's tolerance allows.
  if (rc == HAMK_OK && *rk4_ok && *rkf_ok && usable && !flagged) {
    const int64_t B3 = 4096;                                   // 64 wavefronts: many divergence patterns
    const size_t c3 = (size_t)n * B3;
    const double T = 0.02;
    std::vector<double> q3(c3), p3(c3), ref3(2 * c3), run[2] = {std::vector<double>(2 * c3), std::vector<double>(2 * c3)};
    std::vector<int32_t> st_ref((size_t)B3), st_run[2] = {std::vector<int32_t>((size_t)B3), std::vector<int32_t>((size_t)B3)},
        ns_run[2] = {std::vector<int32_t>((size_t)B3), std::vector<int32_t>((size_t)B3)};
    for (int j = 0; j < n; ++j)
      for (int64_t i = 0; i < B3; ++i) {
        const double u = std::fmod(0.6180339887498949 * (double)(i + 1) + 0.37 * j, 1.0);     // low-discrepancy in [0, 1)
        const double w = std::fmod(0.7548776662466927 * (double)(i + 1) + 0.19 * j, 1.0);
        q3[(size_t)j * B3 + i] = 0.31 + 0.07 * j + 0.29 * u;
        p3[(size_t)j * B3 + i] = 0.23 - 0.05 * j + 0.29 * w;
      }
    double *e_q = nullptr, *e_p = nullptr; int32_t *e_st = nullptr, *e_ns = nullptr;
    bool alloc_ok = hipMalloc((void**)&e_q, c3 * 8) == hipSuccess && hipMalloc((void**)&e_p, c3 * 8) == hipSuccess &&
                    hipMalloc((void**)&e_st, B3 * 4) == hipSuccess && hipMalloc((void**)&e_ns, B3 * 4) == hipSuccess;
    long long b3 = B3;
    auto upload = [&]() { hipMemcpy(e_q, q3.data(), c3 * 8, hipMemcpyHostToDevice); hipMemcpy(e_p, p3.data(), c3 * 8, hipMemcpyHostToDevice); };
    auto download = [&](std::vector<double>& y) { hipMemcpy(y.data(), e_q, c3 * 8, hipMemcpyDeviceToHost); hipMemcpy(y.data() + c3, e_p, c3 * 8, hipMemcpyDeviceToHost); };
    if (alloc_ok) {
      std::vector<double> ref3b(2 * c3);
      // Between repeated launches a kernel of the module fills every VGPR of every SIMD with launch-dependent
      // values: a stepping kernel that reads registers it did not write (scripts/probes/sgpr_spill_repro)
      // agrees with itself back to back and differs once something else has used the registers.
      auto scribble = [&](unsigned seed) {
        void* as[] = {&seed};
        if (hipModuleLaunchKernel(s->mod().fn[K_SCRIBBLE], 2048, 1, 1, 256, 1, 1, 0, s->cur->stream, as, nullptr) != hipSuccess) (void)hipGetLastError();
      };
      for (int r = 0; r < 2 && rc == HAMK_OK; ++r) {         // the fixed-step kernel, twice as well
        scribble(0x9e3779b9u * (unsigned)(r + 1));
        upload();
        double ddt = T / 64, no_drift = 0.0; int ns = 64;
        void* a4[] = {&e_q, &e_p, &b3, &ddt, &ns, &no_drift, &e_st};
        rc = launch(s, K_RK4, B3, a4);
        if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RK4 kernel failed");
        download(r == 0 ? ref3 : ref3b);
      }
      hipMemcpy(st_ref.data(), e_st, B3 * 4, hipMemcpyDeviceToHost);
      if (rc == HAMK_OK && std::memcmp(ref3.data(), ref3b.data(), 2 * c3 * 8) != 0) {
        *rk4_ok = false;
        g_selfcheck_detail = "two runs of the RK4 kernel on the same input DIFFER";
      }
      for (int r = 0; r < 2 && rc == HAMK_OK && has_rkf; ++r) {
        scribble(0x85ebca6bu * (unsigned)(r + 3));
        upload();
        double h0 = T / 100, ea = kRefEpsilon, er = kRefEpsilon, t0 = 0.0, t1 = T;
        int nt = 2, flags = rkf_flags(1, 1, s->gsl_api), max_sub = 4096;
        const double *cq = e_q, *cp = e_p, *cts = nullptr;
        int ncalls = 1, it_every = 0;
        void* a5[] = {&cq, &cp, &e_q, &e_p, &b3, &nt, &cts, &t0, &t1, &h0, &ea, &er, &flags, &max_sub, &e_st, &e_ns, &ncalls, &it_every};
        rc = launch(s, K_RKF45, B3, a5);
        if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RKF45 kernel failed");
        download(run[r]);
        hipMemcpy(st_run[r].data(), e_st, B3 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ns_run[r].data(), e_ns, B3 * 4, hipMemcpyDeviceToHost);
      }
      if (rc == HAMK_OK && has_rkf) {
        bool same = std::memcmp(run[0].data(), run[1].data(), 2 * c3 * 8) == 0 && ns_run[0] == ns_run[1] && st_run[0] == st_run[1];
        double worst = 0.0;
        for (int64_t i = 0; i < B3; ++i) {
          if (st_ref[(size_t)i] != 0 || st_run[0][(size_t)i] != 0) continue;       // outside the system's domain: cannot judge
          if (ns_run[0][(size_t)i] > 16) continue;       // a hard stretch: 64 RK4 steps are no yardstick there
          for (int j = 0; j < 2 * n; ++j) {
            const double a = run[0][(size_t)j * B3 + i], f = ref3[(size_t)j * B3 + i];
            const double e = std::fabs(a - f) / std::fmax(1.0, std::fabs(f));
            if (!(e <= worst)) worst = e;
          }
        }
        if (!same || !(worst <= 1e-4)) {
          *rkf_ok = false;
          char msg[160];
          std::snprintf(msg, sizeof msg, "adaptive end-to-end check: two runs %s, worst deviation from 64 RK4 steps %.3g",
                        same ? "agree" : "DIFFER", worst);
          g_selfcheck_detail = msg;
          if (std::getenv("HAMK_SELFCHECK_VERBOSE")) std::fprintf(stderr, "hamk self-check: %s\n", msg);
        }
      }
    }
    hipFree(e_q); hipFree(e_p); hipFree(e_st); hipFree(e_ns);
    (void)hipGetLastError();
  }
  if (const char* e = test_env("HAMK_SELFCHECK_FAULT")) {        // test hook: pretend the unrolled body is wrong
    if (std::strstr(e, "rk4") && !s->curv->desc.rk4_stage_loop) *rk4_ok = false;
    if (std::strstr(e, "rkf") && !s->curv->desc.rkf_stage_loop) *rkf_ok = false;
  }
  return rc;
}

static void derive_body_fields(const hamk_system* s, SystemDesc& d);

static int post_build_checks(hamk_system* s, Variant* v);
static int self_check(hamk_system* s) {
  if (!s->self_check_on) return HAMK_OK;
  Variant* v = s->curv;
  if (s->mod().self_checked) return HAMK_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool rk4_ok = true, rkf_ok = true;
    int rc = self_check_once(s, &rk4_ok, &rkf_ok);
    if (rc != HAMK_OK) return rc;
    if (rk4_ok && rkf_ok) { s->mod().self_checked = true; return HAMK_OK; }
    const bool can_retry = attempt == 0 && v->mapping == HAMK_MAP_LANE &&
                           ((!rk4_ok && !v->desc.rk4_stage_loop) || (!rkf_ok && !v->desc.rkf_stage_loop));
    if (!can_retry)
      return fail(HAMK_ERR_COMPILE, std::string("self-check failed: the fused ") + (!rk4_ok ? "RK4" : "RKF45") +
                                        " kernel disagrees with the hamEqs kernel (miscompiled module?)" +
                                        (g_selfcheck_detail.empty() ? "" : " [" + g_selfcheck_detail + "]"));
    if (!rk4_ok) v->desc.rk4_stage_loop = true;            // rebuild with the stage-loop bodies
    if (!rkf_ok) v->desc.rkf_stage_loop = true;
    derive_body_fields(s, v->desc);
    v->source = generate_source(v->desc);
    rc = build_code(v, s->cache_on, build_force(s));
    if (rc != HAMK_OK) return rc;
    rc = post_build_checks(s, v);                          // (the rebuilt module passes the same look-at-the-built-kernel rules as a first build)
    if (rc != HAMK_OK) return rc;
    v->generation++;                                       // other devices reload (and re-check) lazily
    for (DevState* d : s->devs) if (d != s->cur) d->mod[v->mapping].self_checked = false;
    rc = load_modules(s);
    if (rc != HAMK_OK) return rc;
    v->self_check_rebuilds++;
  }
  return fail(HAMK_ERR_COMPILE, "self-check failed");
}

static int load_modules(hamk_system* s) {
  DevModule* d = &s->mod();
  const Variant* v = s->curv;
  d->unload();
  HIP_TRY(hipModuleLoadData(&d->module, v->code.data()));
  if (!v->code2.empty()) HIP_TRY(hipModuleLoadData(&d->module2, v->code2.data()));
  for (int k = 0; k < K__COUNT; ++k)
    if (v->has[k]) HIP_TRY(hipModuleGetFunction(&d->fn[k], (v->use2[k] && d->module2) ? d->module2 : d->module, kKernelNames[k]));
  d->code_generation = v->generation;
  return HAMK_OK;
}

// The state of the calling thread's current device (created on first use; nothing is loaded yet).
int current_device_state(hamk_system* s) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  if (!s->cur || s->cur->device != dev) {
    s->cur = nullptr;
    for (DevState* d : s->devs) if (d->device == dev) s->cur = d;
    if (!s->cur) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, dev));
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HAMK_ERR_NODEVICE, std::string("device ") + prop.gcnArchName + " is not gfx950 (MI355X); libhamk has no other code path");
      DevState* d = new DevState();
      d->device = dev;
      s->devs.push_back(d);
      s->cur = d;
    }
  }
  return HAMK_OK;
}

int variant_for(hamk_system* s, int64_t B, int kernel, Variant** out);
std::string check_options(const hamk_options& o, int n);

// Everything a call over B trajectories needs in place: the device state, the specialisation chosen for (n, B) --
// built on first use --, its modules loaded on this device and self-checked.
int bind_device(hamk_system* s, int64_t B, int kernel) {
  TRY0(current_device_state(s));
  TRY0(variant_for(s, B, kernel, &s->curv));
  DevModule& m = s->mod();
  if (m.module && m.code_generation == s->curv->generation) return HAMK_OK;
  TRY0(load_modules(s));
  return self_check(s);
}

static int64_t trajectories_per_block(const Variant* v) {
  // lane kernels: one trajectory per thread; wave kernels: 64/NP trajectories per wavefront; quad: four lanes each
  const int n = v->desc.n;
  if (v->mapping == HAMK_MAP_WAVE) return 4 * (64 / (n <= 16 ? 16 : n <= 32 ? 32 : 64));
  if (v->mapping == HAMK_MAP_QUAD) return 64;
  return 256;
}

int launch(hamk_system* s, KernelId k, int64_t B, void** args) {
  const unsigned block = 256;
  const int64_t per_block = trajectories_per_block(s->curv);
  const int64_t grid = (B + per_block - 1) / per_block;
  if (grid > 0x7fffffffLL) return fail(HAMK_ERR_INVALID, "ensemble too large for one launch");
  HIP_TRY(hipModuleLaunchKernel(s->mod().fn[k], (unsigned)grid, 1, 1, block, 1, 1, 0, s->cur->stream, args, nullptr));
  return HAMK_OK;
}

// ---------------------------------------------------------------------------
// options -> specialisations
// ---------------------------------------------------------------------------
// Ensemble size below which a lane-kernel system (n <= 16) runs four lanes per trajectory instead (0: never).
// One trajectory per lane does the least work per trajectory but puts 64 of them in a wavefront: 65 536 trajectories are
// 1024 wavefronts -- one per SIMD -- and every halving of the ensemble idles half the chip, while the quad kernels spread
// the same trajectories over four times the wavefronts.  Measured on MI355X (profiles/r03_throughput_vs_B.jsonl,
// r03_throughput_vs_B_quad.jsonl; RK4 steps/s):
//     chain16   B = 8192: lane 2.85e8, quad 4.76e8, wave 2.28e8    16384: 5.67e8 / 9.48e8 / 2.42e8    32768: 1.13e9 / 9.5e8
//     chain8    B = 8192: lane 1.23e9, quad 1.12e9, wave 4.8e8     16384: 2.45e9 / 2.22e9             (lane throughout)
//     threeBodyPolar (n = 6): lane 2.1e9 at 8192 against quad 1.15e9 (lane throughout)
//     chain14   B = 8192: lane 3.86e8, quad 4.98e8    16384: 7.69e8 / 9.92e8    32768: 1.54e9 / 1.28e9
//     chain12   B = 8192: lane 5.33e8, quad 7.01e8    16384: 1.06e9 / 1.37e9    32768: 2.12e9 / 1.73e9
//     chain11   B = 8192: lane 6.69e8, quad 7.39e8    16384: 1.34e9 / 1.46e9    32768: 2.68e9 / 1.85e9
//     chain10   B = 8192: lane 7.99e8, quad 7.94e8    16384: 1.59e9 / 1.58e9    32768: 3.17e9 / 1.99e9   (a tie: lane)
// (profiles/r03_rules_probe.jsonl; chain13 the same picture, +15 %).  The wave-cooperative kernels never win at n <= 16
// for B >= 8192.
static int64_t quad_below(int n) {
  return n >= 11 ? 32768 : 0;
}

static bool env_flag(const char* name, bool* value) {           // "0" / "1" test overrides
  const char* e = test_env(name);
  if (!e || (e[0] != '0' && e[0] != '1')) return false;
  *value = e[0] == '1';
  return true;
}

std::string check_options(const hamk_options& o, int n) {
  auto in = [](int v, std::initializer_list<int> ok) { for (int k : ok) if (v == k) return true; return false; };
  if (!in(o.mapping, {HAMK_AUTO, HAMK_MAP_LANE, HAMK_MAP_WAVE, HAMK_MAP_QUAD})) return "mapping must be HAMK_AUTO or a HAMK_MAP_* value";
  if (o.mapping == HAMK_MAP_LANE && n > 16) return "unsupported: HAMK_MAP_LANE needs n <= 16 (one trajectory no longer fits one lane)";
  if (o.mapping == HAMK_MAP_QUAD && n > 32) return "unsupported: HAMK_MAP_QUAD needs n <= 32 (a lane holds a quarter of K in registers)";
  if (!in(o.ad_mode, {HAMK_AUTO, HAMK_AD_H, HAMK_AD_D, HAMK_AD_R})) return "ad_mode must be HAMK_AUTO or a HAMK_AD_* value";
  if (!in(o.rk4_body, {HAMK_AUTO, HAMK_BODY_UNROLLED, HAMK_BODY_STAGE_LOOP}) || !in(o.rkf_body, {HAMK_AUTO, HAMK_BODY_UNROLLED, HAMK_BODY_STAGE_LOOP}))
    return "rk4_body / rkf_body must be HAMK_AUTO or a HAMK_BODY_* value";
  if (!in(o.trig, {HAMK_AUTO, HAMK_TRIG_DIRECT, HAMK_TRIG_TABLE, HAMK_TRIG_TABLE_ROTATE})) return "trig must be HAMK_AUTO or a HAMK_TRIG_* value";
  if (!in(o.gsl_api, {HAMK_AUTO, 1, 2})) return "gsl_api must be HAMK_AUTO, 1 (gsl_odeiv) or 2 (gsl_odeiv2)";
  if (!in(o.build, {HAMK_AUTO, HAMK_BUILD_DEFAULT, HAMK_BUILD_NOLICM})) return "build must be HAMK_AUTO or a HAMK_BUILD_* value";
  for (int v : {o.self_check, o.k_reassoc, o.rk4_park, o.rkf_park, o.cache})
    if (!in(v, {HAMK_AUTO, HAMK_ON, HAMK_OFF})) return "switches must be HAMK_AUTO, HAMK_ON or HAMK_OFF";
  if (o.rk4_min_waves < 0 || o.rk4_min_waves > 8) return "rk4_min_waves must be 0 (auto) .. 8";
  if (o.max_substeps < 0) return "max_substeps must be >= 0";
  if (o.ensemble_size < 0) return "ensemble_size must be >= 0";
  return std::string();
}

// Which lanes serve a trajectory for an ensemble of B (hamk_options::mapping = HAMK_AUTO).
// The lane kernels do the least work per trajectory (compile-time sparsity of the seeds, everything in registers) but
// put 64 trajectories in a wavefront: below ~64 x 1024 SIMDs trajectories they leave SIMDs idle, and for the systems
// whose lane kernel is large the wave-cooperative kernels (4 trajectories per wavefront at n <= 16) then win.
// Thresholds measured on MI355X (scripts/sweep_batch.py -> profiles/r03_throughput_vs_B*.jsonl).
// Which kernels the quad module (hamk_quad.hpp) provides: all eight since the second half of round 3.  The per-kernel
// dispatch stays: a module that lacks a kernel leaves it to the module that serves the system's size otherwise.
static bool quad_has(int kernel) { (void)kernel; return true; }      // (the first version provided the four kernels of the hot path only)

// Can a lane run the per-trajectory first-order sweep of this system with compile-time seeds?  It keeps one register pair
// per DISTINCT entry of the Jacobian (hamk_codegen.cpp distinct_jacobian_entries): 2n for a chain, m n for a dense map.
static bool quad_eligible(hamk_system* s) {
  if (s->quad_eligible < 0) s->quad_eligible = distinct_jacobian_entries(s->base) <= 8 * s->base.n ? 1 : 0;
  return s->quad_eligible == 1;
}
// Round 6: a DENSE Jacobian whose entries are cheap.  With compile-time seeds a lane evaluates J at forward_gradient_work
// operations per sweep whatever the number of distinct entries; hamk_quad.hpp's dense path (K accumulated in passes over groups of
// row slots, one re-evaluation of J per pass) then needs ~3 x that + n^2 m / 8 per lane for SIXTEEN trajectories per wavefront,
// where the wave-cooperative kernels evaluate the whole tape in every lane for two.  Taken where a sweep costs at most 4 m n
// (x = 2 q + A sin q + B cos q: 2 m n); a map whose operations all depend on all inputs (tape x n) stays on the wave kernels.
// Measured on MI355X (profiles/r06_dense_quad_ab.jsonl): dense32 / dense24 RK4 steps/s on this mapping against the wave kernels'.
static bool quad_dense_eligible(hamk_system* s) {
  if (s->quad_dense_eligible < 0) {
    const long long mn = (long long)s->base.m * s->base.n;
    s->quad_dense_eligible = (s->base.n > 16 && s->base.n <= 32 && forward_gradient_work(s->base) <= 4 * mn) ? 1 : 0;
  }
  bool b = false;
  if (env_flag("HAMK_QUAD_DENSE", &b)) return b && s->base.n > 16 && s->base.n <= 32;      // test override (A/B against the wave kernels)
  return s->quad_dense_eligible == 1;
}

// Is K = J^T M J semi-definite by construction?  Only then may a kernel factorise it without pivoting.  The reference
// inverts EVERY K by LU with partial pivoting (hmatrix `inv`, Hamilton.hs:321, :381), so a system with a non-positive
// inertia -- K symmetric but possibly indefinite, and still invertible -- must run on kernels that pivot: the lane
// kernels (solve_spd falls back to solve_lu per trajectory) and the wave-cooperative ones (solve_pivoted).  The four-lane
// kernels do not, and are never chosen for such a system.
static bool inertia_positive(const hamk_system* s) {
  for (double w : s->base.inertia) if (!(w > 0.0)) return false;
  return true;
}

static int choose_mapping(hamk_system* s, int64_t B, int kernel) {
  const int n = s->base.n;
  const int rest = n > 16 ? HAMK_MAP_WAVE : HAMK_MAP_LANE;  // where the kernels the quad module lacks run
  if (s->opt.mapping != HAMK_AUTO) return (s->opt.mapping == HAMK_MAP_QUAD && !quad_has(kernel)) ? rest : s->opt.mapping;
  if (s->ensemble_size > 0) B = s->ensemble_size;           // the launch is a piece of a larger ensemble: one mapping for all pieces
  bool w = false;
  const bool pos = inertia_positive(s);
  if (pos && env_flag("HAMK_QUAD", &w) && w && n <= 32) return quad_has(kernel) ? HAMK_MAP_QUAD : rest;      // tests / experiments
  const bool no_quad = !pos || (env_flag("HAMK_QUAD", &w) && !w);
  if (env_flag("HAMK_WAVE", &w)) return (w || n > 16) ? HAMK_MAP_WAVE : HAMK_MAP_LANE;
  if (n > 32) return HAMK_MAP_WAVE;
  if (n > 16) return (!no_quad && quad_has(kernel) && (quad_eligible(s) || quad_dense_eligible(s))) ? HAMK_MAP_QUAD : HAMK_MAP_WAVE;
  const int64_t below = quad_below(n);                      // ensembles smaller than this leave the lane kernels
  if (B < below && !no_quad && quad_eligible(s)) return quad_has(kernel) ? HAMK_MAP_QUAD : HAMK_MAP_LANE;
  return HAMK_MAP_LANE;
}

// What follows from the choice of stepping bodies.  make_desc ends with it; the two places that switch a built variant to the
// stage-loop bodies afterwards (the > 64 KiB fallback of variant_for, the self-check's one rebuild) call it again, so that the
// rebuilt module is the configuration the stage-loop body ships in and hamk_system_get_options reports what runs.
static void derive_body_fields(const hamk_system* s, SystemDesc& d) {
  const hamk_options& o = s->opt;
  const int n = d.n, mapping = d.mapping;
  bool b = false;
  // RK4 stage loop with y / acc parked in LDS (hamk_device.hpp rk4_body): where one right-hand side alone fills the
  // register file the waiting state is what spills; chain16 300 spilled registers -> 34, none in the loop.  Measured
  // at B = 65 536 (profiles/r03_rules_probe.jsonl; RK4 steps/s parked / not): chain16 2.09e9 / 1.01e9, chain14
  // 2.77e9 / 2.63e9, chain13 3.19e9 / 3.46e9, chain12 3.74e9 / 3.90e9, chain10 5.03e9 / 5.42e9 -- it pays from n = 14
  d.rk4_park = mapping == HAMK_MAP_LANE && n >= 14;
  if (o.rk4_park != HAMK_AUTO) d.rk4_park = o.rk4_park == HAMK_ON;
  if (d.rk4_park && (!d.rk4_stage_loop || mapping != HAMK_MAP_LANE || n > 16)) d.rk4_park = false;          // 2 x 2n x 2 KiB of LDS per block: n <= 16
  // RKF45 stepper whose nine vectors (18 n doubles per trajectory) wait in LDS (y, dydt, the first k's) and in a
  // run-time-indexed private array (scratch memory, touched only between right-hand sides) instead of competing with K
  // for registers (hamk_device.hpp rkf45_body_parked, hamk_quad.hpp rkf45_body_parked).  Measured on MI355X, stepHam
  // calls/s, registers / parked.  Lane kernels at B = 65 536 (profiles/r03_lane_rkf_park.jsonl): chain16 1.57e7 / 7.88e7
  // (1516 -> 24 spilled registers), chain14 2.87e7 / 1.27e8, chain12 5.08e7 / 1.80e8, chain10 7.59e7 / 2.50e8, chain8
  // 2.03e8 / 3.77e8, chain7 3.29e8 / 4.99e8, chain6 5.00e8 / 6.12e8, threeBodyPolar 7.08e8 / 8.21e8, chain5 and chain4 ties.
  // Quad kernels at B = 16 384 (profiles/r03_quad_rkf_park.jsonl): chain32 4.14e6 / 6.63e6, chain24 1.40e7 / 1.74e7,
  // chain20 2.70e7 / 2.94e7, chain17 3.34e7 / 3.48e7
  // Lane kernels: the stage-loop body IS the parked one (round 4: the unparked stage loop tied at n = 4, 5, lost from n = 6 and
  // was removed), so rkf_park follows rkf_body there; the option decides for the quad kernels only.
  d.rkf_park = mapping == HAMK_MAP_QUAD && n >= 17;
  if (mapping == HAMK_MAP_QUAD) {
    if (o.rkf_park != HAMK_AUTO) d.rkf_park = o.rkf_park == HAMK_ON;
    else if (env_flag("HAMK_RKF_PARK", &b)) d.rkf_park = b;
  }
  if (mapping == HAMK_MAP_LANE) d.rkf_park = d.rkf_stage_loop;
  // Small systems (n <= 7): the parked stepper at TWO wavefronts per SIMD.  With every waiting vector in LDS (round 3; 144 KiB
  // per block at n = 6) a CU holds one block, and a right-hand side of a few hundred dependent instructions cannot keep a
  // SIMD busy alone (VALU issue 0.37).  Here the block takes half the LDS -- 36 doubles per lane: y, dydt and one or two
  // more rows -- the other rows are REGISTERS (the private array is indexed by literals and promoted; 256 VGPRs at n = 6,
  // two spilled), and the kernel is capped for two wavefronts.  Measured on MI355X, stepHam dt at B = 262 144, one
  // wavefront / two (profiles/r04_rkf_hybrid_ab.jsonl): chain4 1.57e9 / 2.01e9, chain5 1.27e9 / 1.62e9, chain6 1.05e9 /
  // 1.32e9, threeBodyPolar 1.15e9 / 1.41e9, chain7 8.8e8 / 9.4e8; states equal to 4e-16, sub-step counts identical.  (Without
  // the sincos table to make room: 1.34e9; two LDS rows: 1.19e9; three wavefronts: 8.5e8.  From n = 8 the right-hand side
  // alone needs more than 256 registers.)  The rule was measured on the chains and threeBodyPolar: a small-n system with a
  // heavy tape may not fit 256 registers -- variant_for looks at the built kernel and goes back to one wavefront where the
  // cap made it spill (rkf_two_waves_off); HAMK_RKF_TWO_WAVES=0|1 is the test override.
  d.rkf_two_waves = mapping == HAMK_MAP_LANE && d.rkf_stage_loop && n <= 7 && !d.rkf_two_waves_off;
  // Parked stepper, n = 8, 9 (four rows of 2n doubles in LDS: y, dydt, k2, k3 -- the systems whose attempt moves the most LDS
  // bytes per right-hand side): components stored in PAIRS, [j / 2][lane][2], so that a row is read by ds_read_b128 instead of
  // ds_read2st64_b64 -- the same 16 bytes per lane at half the LDS-array cycles, and the one-wavefront-per-SIMD rate of the
  // 16-byte read is the higher one.  Measured (profiles/r05b_pair_rows_stepham.jsonl, stepHam dt, B = 65 536, bitwise equal
  // results): chain8 +3.1 %, chain9 +5.9 %; chain10 -1.3 %, chain11..13 +1.1 / +1.5 / 0 % (three and two LDS rows: left alone);
  // the RK4 kernel that parks its state (n >= 14) loses 1-4 % to the 16-byte stores (r05a_pair_rows_ab.jsonl): not there.
  d.pair_rows = mapping == HAMK_MAP_LANE && d.rkf_stage_loop && (n == 8 || n == 9);
  if (env_flag("HAMK_PAIR_ROWS", &b)) d.pair_rows = b && mapping == HAMK_MAP_LANE;
  if (env_flag("HAMK_RKF_TWO_WAVES", &b)) d.rkf_two_waves = b && mapping == HAMK_MAP_LANE && d.rkf_stage_loop && n <= 7;
}

static SystemDesc make_desc(const hamk_system* s, int mapping, bool* forced_rk4, bool* forced_rkf) {
  const hamk_options& o = s->opt;
  SystemDesc d = s->base;
  const int n = d.n, m = d.m;
  bool b = false;
  d.mapping = mapping;
  d.wave = mapping == HAMK_MAP_WAVE;
  // four lanes per trajectory, dense Jacobian (chosen by quad_dense_eligible, or the mapping forced): K in passes (hamk_quad.hpp assemble_dense)
  d.quad_dense = mapping == HAMK_MAP_QUAD && distinct_jacobian_entries(s->base) > 8 * n;
  // second-order AD: measured on MI355X (scripts/sweep.py): H >= D up to n = 3, D ahead from n = 4; the reverse sweep
  // pays from n = 8 (chain8 +4 %, chain16 +12 %); below, the compiler already strips the structural zeros of the
  // directional jets and Jet2 is as cheap
  d.mode_h = (n <= 3);
  d.mode_r = (n >= 8);
  int ad = o.ad_mode;
  if (ad == HAMK_AUTO) if (const char* e = test_env("HAMK_AD_MODE")) ad = (e[0] == 'H' || e[0] == 'h') ? HAMK_AD_H : (e[0] == 'D' || e[0] == 'd') ? HAMK_AD_D : (e[0] == 'R' || e[0] == 'r') ? HAMK_AD_R : HAMK_AUTO;
  if (ad == HAMK_AD_H) { d.mode_h = true; d.mode_r = false; }
  if (ad == HAMK_AD_D) { d.mode_h = false; d.mode_r = false; }
  if (ad == HAMK_AD_R) { d.mode_h = false; d.mode_r = true; }
  d.rk4_stage_loop = (n >= 7);
  *forced_rk4 = true;
  if (o.rk4_body != HAMK_AUTO) d.rk4_stage_loop = o.rk4_body == HAMK_BODY_STAGE_LOOP;
  else if (env_flag("HAMK_RK4_LOOP", &b)) d.rk4_stage_loop = b;
  else *forced_rk4 = false;
  d.rkf_stage_loop = (n >= 4);
  *forced_rkf = true;
  if (o.rkf_body != HAMK_AUTO) d.rkf_stage_loop = o.rkf_body == HAMK_BODY_STAGE_LOOP;
  else if (env_flag("HAMK_RKF_LOOP", &b)) d.rkf_stage_loop = b;
  else *forced_rkf = false;
  // n > 32 (one trajectory per wavefront): the RK4 kernel capped at 256 VGPRs -- two wavefronts per SIMD, ~160
  // spilled registers -- beats one wavefront with everything in registers: chain48 1.05e7 -> 1.45e7, chain64
  // 7.6e6 -> 9.5e6 RK4 steps/s on MI355X (profiles/r02_wave_blocked.jsonl)
  d.rk4_min_waves = 1;
  if (d.wave) d.rk4_min_waves = 2;                          // (n <= 32: two trajectories per wavefront at 228 registers, hamk_wave.hpp)
  if (o.rk4_min_waves > 0) d.rk4_min_waves = o.rk4_min_waves;
  d.k_reassoc = true;
  if (o.k_reassoc != HAMK_AUTO) d.k_reassoc = o.k_reassoc == HAMK_ON;
  d.k_symbolic = true;                                      // (where it applies and pays: hamk_codegen.cpp symbolic_mass_matrix)
  if (env_flag("HAMK_K_SYMBOLIC", &b)) d.k_symbolic = b;    // test override (A/B)

  {
    // sincos in the stepping kernels (hamk_device.hpp StageTrig).  Every evaluation through the LDS table is
    // the fewest instructions, but each is a 16-byte gather at a lane-dependent address (~20-25 LDS cycles
    // per wavefront) and the CU's 16 wavefronts share one LDS unit: where a right-hand side is short and
    // trig-dense the unit saturates, and taking only the step's one full evaluation from the table (stages
    // 2-4 by rotation in registers) is faster.  Measured on MI355X (profiles/r02_sweep_trig.jsonl): rotation
    // wins for doublePendulum (2 sites per ~90-instruction RHS: 8.36 vs 8.21e10) and pendulum, the table for
    // twoBody (+8 %), threeBodyPolar (+8 %) and the chains (+24 % at n = 8); spring is a tie.  The rule
    // below reproduces those choices from an estimate of the instructions per RHS and sincos site.
    const int f_nops = (int)d.f_ops.size(), u_nops = (int)d.u_ops.size();
    std::vector<char> seen(f_nops > 0 ? f_nops : 1, 0);
    int sites = 0;
    for (int i = 0; i < f_nops; ++i)
      if ((d.f_ops[i].op == HAMK_OP_SIN || d.f_ops[i].op == HAMK_OP_COS) && !seen[d.f_ops[i].a]) { seen[d.f_ops[i].a] = 1; ++sites; }
    const double width = d.mode_h ? 1.0 + n + 0.5 * n * (n + 1) : 3.0 * n + 3.0;      // jet components carried per tape value
    const double est_rhs = (f_nops + u_nops) * width + 2.0 * m * n * n + n * n * n / 3.0;
    d.use_lut = (sites >= 1 && sites <= 4 && est_rhs / sites < 100.0) ? 2 : 1;
    // round 6: where K and dT/dq come from the symbolic mass matrix the right-hand side is a fraction of that estimate, and rotations
    // win for every system with 1-4 sites -- measured on one box (profiles/r06_trig_rule_ab.jsonl): threeBodyPolar 3.23e10 (table) /
    // 3.39e10 (one table evaluation per step + rotations), spring 6.16e10 / 6.33e10; doublePendulum and pendulum had them already
    if (mapping == HAMK_MAP_LANE && sites >= 1 && sites <= 4 && n <= 7 && d.k_symbolic && symbolic_rhs_applies(d)) d.use_lut = 2;
    // the four-lane kernels from n = 14: no table.  Their register file is K's, the table costs 8 KiB of an LDS that is nearly
    // full and a handful of registers: chain32 2.50e8 steps/s without it, 2.28e8 with (Scratch_Size 372 -> 488 B, HBM traffic
    // per launch 233 -> 788 MB; profiles/r03e_chain32_summary.json vs the first r04 pass, where this rule had been dropped
    // by mistake together with the lane-kernel rule it used to share a line with)
    if (n >= 14 && mapping == HAMK_MAP_QUAD) d.use_lut = 0;
  }
  // sincos_lut's nine fp64 literals in VECTOR registers for the mid-size lane kernels (hamk_device.hpp LutK): every RKF45 kernel is
  // taken from the build without MachineLICM, which re-materialises them -- two s_mov_b32 each -- at every evaluation: 144
  // scalar moves per right-hand side of chain8, a seventh of the issue slots of a kernel that has one wavefront per SIMD.  With
  // the literals gone the RK4 kernels of n = 10..16 also stop spilling SGPRs in the default build and are taken from it.
  // Measured on MI355X (profiles/r04_trig_const_vgpr_ab.jsonl; stepHam calls/s, RK4 steps/s): chain8 +6 % / 0, chain10 +4 % /
  // +4 %, chain12 +4 % / +5 %, chain14 +5 % / +1 %; chain16 +3 % / -2 %, chain6 and threeBodyPolar +-1 %, doublePendulum
  // -6 % / -6 % (18 more registers where occupancy pays) -- so: 8 <= n <= 14.
  d.trig_const_vgpr = mapping == HAMK_MAP_LANE && n >= 8 && n <= 14;
  if (env_flag("HAMK_TRIG_CONST_VGPR", &b)) d.trig_const_vgpr = b && mapping == HAMK_MAP_LANE;      // test override (the rule above was measured on the chains)
  if (o.trig != HAMK_AUTO) d.use_lut = o.trig == HAMK_TRIG_DIRECT ? 0 : (o.trig == HAMK_TRIG_TABLE ? 1 : 2);
  else if (const char* e = test_env("HAMK_TRIG_LUT")) { if (e[0] >= '0' && e[0] <= '2') d.use_lut = e[0] - '0'; }
  derive_body_fields(s, d);
  return d;
}

// What variant_for decides by LOOKING AT THE BUILT KERNELS: a kernel over 64 KiB goes to the stage-loop body, a two-wavefront
// stepper that spills goes back to one wavefront, a wave RK4 kernel that spills by the hundred under its two-wavefront cap is rebuilt
// for one.  One helper, applied until nothing changes any more, for a first build AND for the self-check's rebuild (which flips bodies
// and re-derives fields: the rebuilt module must pass the same rules -- ADVICE r05).
static int post_build_checks(hamk_system* s, Variant* v) {
  const int mapping = v->mapping;
  const size_t kLimit = 64 * 1024;
  const bool lane = mapping == HAMK_MAP_LANE;
  bool lane_flag_ = false;
  int rc = HAMK_OK;
  for (int pass = 0; pass < 4; ++pass) {
    const std::string before = v->source;
    // keep every kernel comfortably inside SOPP branch reach: fall back to the stage-loop bodies
    const bool big_rkf = lane && !v->forced_rkf_body && !v->desc.rkf_stage_loop && chosen_kernel_bytes(v, K_RKF45) > kLimit;
    const bool big_rk4 = lane && !v->forced_rk4_body && !v->desc.rk4_stage_loop && chosen_kernel_bytes(v, K_RK4) > kLimit;
    if (big_rkf || big_rk4) {
      if (big_rkf) v->desc.rkf_stage_loop = true;
      if (big_rk4) v->desc.rk4_stage_loop = true;
      derive_body_fields(s, v->desc);
      v->source = generate_source(v->desc);
      rc = build_code(v, s->cache_on, build_force(s));
      if (rc != HAMK_OK) return rc;
    }
    // the two-wavefront stepper caps hamk_rkf45_k at 256 registers: where THIS system's right-hand side does not fit (a heavy
    // tape at small n) the cap turns into spill code inside the attempt loop -- back to one wavefront with every row in LDS
    if (lane && v->desc.rkf_two_waves && !env_flag("HAMK_RKF_TWO_WAVES", &lane_flag_) &&
        vgpr_spill_count(v->use2[K_RKF45] ? v->code2 : v->code, kKernelNames[K_RKF45]) > 32) {
      v->desc.rkf_two_waves_off = true;
      derive_body_fields(s, v->desc);
      v->source = generate_source(v->desc);
      rc = build_code(v, s->cache_on, build_force(s));
      if (rc != HAMK_OK) return rc;
    }
    // wave-cooperative RK4 kernel: two wavefronts per SIMD (256 registers) is the measured default -- the chains spill a few hundred
    // registers under the cap and still gain (chain48 1.32e7 vs 1.06e7 steps/s, chain64 8.1e6 vs 7.6e6) -- but a system whose tape
    // is heavy (a dense coordinate map: every lane carries one-direction jets of n^2 terms) spills by the thousand and waits on
    // scratch for three quarters of its cycles: dense32 967 spilled registers, 490 GB of HBM traffic per launch, 4.4e6 steps/s at
    // two wavefronts against 1.30e7 at one with 6 spilled (profiles/r05c_wave_probe.jsonl; dense24, 241 spilled: a tie).
    if (mapping == HAMK_MAP_WAVE && s->opt.rk4_min_waves == 0 && v->desc.rk4_min_waves > 1 &&
        vgpr_spill_count(v->use2[K_RK4] ? v->code2 : v->code, kKernelNames[K_RK4]) > 512) {
      v->desc.rk4_min_waves = 1;
      v->source = generate_source(v->desc);
      rc = build_code(v, s->cache_on, build_force(s));
      if (rc != HAMK_OK) return rc;
    }
    if (v->source == before) break;
  }
  return HAMK_OK;
}

int variant_for(hamk_system* s, int64_t B, int kernel, Variant** out) {
  const int mapping = choose_mapping(s, B, kernel);
  if (s->var[mapping]) { *out = s->var[mapping]; return HAMK_OK; }
  Variant* v = new Variant();
  v->mapping = mapping;
  v->desc = make_desc(s, mapping, &v->forced_rk4_body, &v->forced_rkf_body);
  // an explicitly set option that this specialisation cannot honour is refused, not silently replaced (the lane mapping's
  // stage-loop stepper IS the parked one, its unrolled stepper keeps everything in registers: rkf_park follows rkf_body there)
  // (only where no four-lane variant of this handle could honour it: the mapping pinned to LANE, or n <= 10 -- with the mapping left to
  // the library a system of 11 <= n <= 16 also serves small ensembles on the quad kernels, where rkf_park = OFF means something)
  if (mapping == HAMK_MAP_LANE && (s->opt.mapping == HAMK_MAP_LANE || s->base.n <= 10) &&
      s->opt.rkf_park != HAMK_AUTO && (s->opt.rkf_park == HAMK_ON) != v->desc.rkf_stage_loop) {
    const bool on = s->opt.rkf_park == HAMK_ON;
    delete v;
    return fail(HAMK_ERR_UNSUPPORTED, std::string("hamk_options: rkf_park = ") + (on ? "ON" : "OFF") + " cannot be honoured on the lane mapping: its " +
                                          (on ? "unrolled adaptive body keeps every vector in registers" : "stage-loop adaptive body is the parked one") +
                                          " (set rkf_body instead, or leave rkf_park AUTO)");
  }
  v->source = generate_source(v->desc);
  int rc = build_code(v, s->cache_on, build_force(s));
  if (rc != HAMK_OK) { delete v; return rc; }
  rc = post_build_checks(s, v);
  if (rc != HAMK_OK) { delete v; return rc; }
  // dense map on the four-lane kernels (quad_dense_eligible): the premise is that a tile's accumulators, the entries of J it needs
  // and their sincos pairs fit the registers.  Where the built kernel says otherwise -- it spills by the hundred, and at one wavefront
  // per SIMD a kernel that waits for scratch loses whatever it saves in instructions (first build of dense24, its shared products
  // still materialised once: 1 499 spilled registers, 1.98e7 RK4 steps/s against the wave kernels' 3.05e7; unshared: 20 registers,
  // 6.27e7) -- the system goes back to the wave-cooperative kernels.  Not where the mapping was stated (options, HAMK_QUAD, HAMK_QUAD_DENSE).
  {
    bool forced = false;
    const bool stated = s->opt.mapping != HAMK_AUTO || env_flag("HAMK_QUAD", &forced) || env_flag("HAMK_QUAD_DENSE", &forced);
    if (mapping == HAMK_MAP_QUAD && v->desc.quad_dense && !stated &&
        vgpr_spill_count(v->use2[K_RK4] ? v->code2 : v->code, kKernelNames[K_RK4]) > 768) {
      s->quad_dense_eligible = 0;
      delete v;
      return variant_for(s, B, kernel, out);
    }
  }
  for (int k = 0; k < K__COUNT; ++k) v->has[k] = mapping != HAMK_MAP_QUAD || quad_has(k);
  s->var[mapping] = v;
  *out = v;
  return HAMK_OK;
}

}  // namespace hamk_host

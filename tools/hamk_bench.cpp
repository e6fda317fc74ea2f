// hamk_bench -- the BASELINE.json metric through the C ABI alone, from a compiled host.
//
// What a Haskell (or any non-Python) host gets from libhamk.so: double pendulum (System 4 2,
// /root/reference app/Examples.hs:75-94), an ensemble of B seeded initial conditions drawn in HBM from the
// global index (hamk_sample_batch), toPhase on the device, `launches` x hamk_rk4_steps(nsteps) timed with
// the host clock between hamk_synchronize calls.  No torch, no HIP headers: include/hamk.h and
// include/hamilton.hpp only.  bench.py is the driver's harness (one process per GPU, RCCL); this
// one is the single-process form: --gpus G puts one shard on each of the first G devices, one
// System per device, and gathers the final state with hamk_gather_batch.
//
//   g++ -std=c++17 -O2 -Iinclude tools/hamk_bench.cpp -o hamk_bench -Lhamilton_amd -lhamk -Wl,-rpath,$PWD/hamilton_amd -Wl,-rpath,/opt/rocm/lib
//   ./hamk_bench [--batch 1048576] [--nsteps 100] [--launches 50] [--warmup 5] [--gpus 1] [--dump-first K]
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "hamilton.hpp"

namespace hm = hamilton;

static hm::System double_pendulum() {                     // Examples.hs:75-94 with m1 = m2 = 1
  return hm::mkSystemP({1, 1, 1, 1}, 2,
      [](const std::vector<hm::Var>& q) {
        return std::vector<hm::Var>{hm::sin(q[0]), 1.0 - hm::cos(q[0]),
                                    hm::sin(q[0]) + hm::sin(q[1]) / 2.0, 1.0 - hm::cos(q[0]) - hm::cos(q[1]) / 2.0};
      },
      [](const std::vector<hm::Var>& x) { return 5.0 * (1.0 * x[1] + 1.0 * x[3]); });
}

int main(int argc, char** argv) {
  int64_t B = 1 << 20; int nsteps = 100, launches = 50, warmup = 5, gpus = 1; int64_t dump = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--batch")) B = std::atoll(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--nsteps")) nsteps = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--launches")) launches = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--warmup")) warmup = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--dump-first")) dump = std::atoll(argv[i + 1]);
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  const double dt = 0.01;
  const double PI = 3.14159265358979323846;
  try {
    if (hamk_device_count() < gpus) { std::fprintf(stderr, "need %d HIP device(s), see %d\n", gpus, hamk_device_count()); return 3; }
    std::vector<std::unique_ptr<hm::System>> sys;
    std::vector<hm::DevicePhase> state((size_t)gpus);
    const hm::Box box{{-PI, -PI}, {PI, PI}, {-1.0, -1.0}, {1.0, 1.0}};      // q_box (-pi, pi), qd_box (-1, 1): SURVEY.md 8d C2
    for (int g = 0; g < gpus; ++g) {                        // weak scaling: every device owns B trajectories
      hm::check(hamk_set_device(g));
      sys.emplace_back(new hm::System(double_pendulum()));
      // one ensemble of gpus x B members: every shard on the mapping chosen for the whole, its initial conditions drawn on
      // ITS device from the global trajectory index (per-index splitmix64, seed 20241008 = hamilton_amd/examples.py)
      hm::setEnsembleSize(*sys.back(), (int64_t)gpus * B);
      state[(size_t)g] = hm::samplePhaseDevice(*sys.back(), box, (int64_t)g * B, B, 20241008ull);
    }
    auto step_all = [&]() {
      for (int g = 0; g < gpus; ++g) { hm::check(hamk_set_device(g)); hm::rk4Steps(dt, nsteps, *sys[(size_t)g], state[(size_t)g]); }
    };
    auto sync_all = [&]() {
      for (int g = 0; g < gpus; ++g) { hm::check(hamk_set_device(g)); hm::synchronize(*sys[(size_t)g]); }
    };
    std::vector<double> h0;
    if (dump > 0) { hm::check(hamk_set_device(0)); h0 = hm::hamiltonian(*sys[0], state[0]); }
    for (int w = 0; w < warmup; ++w) step_all();
    sync_all();
    const auto t0 = std::chrono::steady_clock::now();
    for (int l = 0; l < launches; ++l) step_all();
    sync_all();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto g0 = std::chrono::steady_clock::now();
    std::vector<const hm::DevicePhase*> parts;
    for (auto& d : state) parts.push_back(&d);
    hm::check(hamk_set_device(0));
    hm::Phase all = hm::gather(parts);
    const double gather_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count() * 1e3;
    int64_t flagged = 0;
    for (auto& d : state) for (int32_t st : d.download_status()) flagged += (st != 0);
    const double units = (double)gpus * (double)B * nsteps * launches;
    std::printf("{\"metric\": \"RK4 phase-space steps/sec (ensemble)\", \"host\": \"C++ over the C ABI (tools/hamk_bench.cpp)\", "
                "\"value\": %.6e, \"unit\": \"trajectory-steps/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
                "\"ms_per_step\": %.6f, \"trajectories_per_gpu\": %" PRId64 ", \"rk4_steps_per_launch\": %d, "
                "\"hbm_frac_survey_8d\": %.4f, \"gather_ms_to_host\": %.3f, \"status_flagged\": %" PRId64 "}\n",
                units / el, gpus, launches, warmup, el / launches * 1e3, B, nsteps,
                units / el / gpus * 64.0 / 8.0e12, gather_ms, flagged);
    for (int64_t i = 0; i < dump && i < all.B; ++i)
      std::printf("traj %" PRId64 " q = %.17g %.17g p = %.17g %.17g H0 = %.17g\n", i, all.positions[(size_t)i], all.positions[(size_t)all.B + i],
                  all.momenta[(size_t)i], all.momenta[(size_t)all.B + i], h0.empty() ? 0.0 : h0[(size_t)i]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "hamk_bench: %s\n", e.what());
    return 1;
  }
  return 0;
}

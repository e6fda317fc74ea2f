// hamk_bench -- the BASELINE.json metric through the C ABI alone, from a compiled host.
//
// What a Haskell (or any non-Python) host gets from libhamk.so: double pendulum (System 4 2,
// /root/reference app/Examples.hs:75-94), an ensemble of B seeded initial conditions drawn in HBM from the
// global index (hamk_sample_batch), toPhase on the device, `launches` x hamk_rk4_steps(nsteps) timed with
// the host clock between hamk_synchronize calls.  No torch, no HIP headers: include/hamk.h and
// include/hamilton.hpp only.  bench.py is the driver's harness (one process per GPU, RCCL); this
// one is the single-process form: --gpus G puts one shard on each of the first G devices, one
// System per device, and gathers the final state with hamk_gather_batch.  --world W --rank R --id-file F
// is the PROCESS-PER-GPU form of a native host (start the binary W times): rank R drives device R (or
// --device D), the shards meet in ONE all-gather over RCCL at the end (hamk_comm_*: SURVEY.md 8(e)), the
// communicator's id travels through the file F (rank 0 writes it aside and renames it, the others wait
// for it), the timing is the maximum over the ranks, rank 0 prints the line.
//
//   g++ -std=c++17 -O2 -Iinclude tools/hamk_bench.cpp -o hamk_bench -Lhamilton_amd -lhamk -Wl,-rpath,$PWD/hamilton_amd -Wl,-rpath,/opt/rocm/lib
//   ./hamk_bench [--batch 1048576] [--nsteps 100] [--launches 50] [--warmup 5] [--gpus 1] [--dump-first K]
//   for r in 0 1 ... ; do ./hamk_bench --world W --rank $r --id-file /tmp/hamk.id [...] & done; wait
#include <unistd.h>

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

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

static void print_line(double units, double el, int n_gpus, int launches, int warmup, int64_t B, int nsteps, double gather_ms,
                       const char* gather_key, int64_t flagged, const char* how) {
  std::printf("{\"metric\": \"RK4 phase-space steps/sec (ensemble)\", \"host\": \"C++ over the C ABI (tools/hamk_bench.cpp), %s\", "
              "\"value\": %.6e, \"unit\": \"trajectory-steps/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
              "\"ms_per_step\": %.6f, \"trajectories_per_gpu\": %" PRId64 ", \"rk4_steps_per_launch\": %d, "
              "\"hbm_frac_survey_8d\": %.4f, \"%s\": %.3f, \"status_flagged\": %" PRId64 "}\n",
              how, units / el, n_gpus, launches, warmup, el / launches * 1e3, B, nsteps,
              units / el / n_gpus * 64.0 / 8.0e12, gather_key, gather_ms, flagged);
}

// one process per GPU: this process is rank `rank` of `world`
static int run_rank(int world, int rank, int device, const std::string& id_file, int64_t B, int nsteps, int launches, int warmup, int64_t dump) {
  namespace hm = hamilton;
  const double dt = 0.01, PI = 3.14159265358979323846;
  if (device < 0) device = rank;
  if (hamk_device_count() <= device) { std::fprintf(stderr, "rank %d: need HIP device %d, see %d device(s)\n", rank, device, hamk_device_count()); return 3; }
  hm::check(hamk_set_device(device));
  hm::Comm::Id id;
  if (rank == 0) {
    id = hm::Comm::uniqueId();
    const std::string tmp = id_file + ".tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id.data(), 1, id.size(), f) != id.size() || std::fclose(f) != 0 || std::rename(tmp.c_str(), id_file.c_str()) != 0) {
      std::fprintf(stderr, "rank 0: cannot write %s\n", id_file.c_str());
      return 4;
    }
  } else {
    std::FILE* f = nullptr;
    for (int tries = 0; tries < 6000 && !(f = std::fopen(id_file.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    if (!f || std::fread(id.data(), 1, id.size(), f) != id.size()) { std::fprintf(stderr, "rank %d: no id in %s\n", rank, id_file.c_str()); return 4; }
    std::fclose(f);
  }
  hm::Comm comm(id, world, rank);
  if (rank == 0) ::unlink(id_file.c_str());                 // (every rank has read it: hamk_comm_create returned)
  hm::System sys = double_pendulum();
  hm::setEnsembleSize(sys, (int64_t)world * B);
  const hm::Box box{{-PI, -PI}, {PI, PI}, {-1.0, -1.0}, {1.0, 1.0}};
  hm::DevicePhase state = hm::samplePhaseDevice(sys, box, (int64_t)rank * B, B, 20241008ull);
  for (int w = 0; w < warmup; ++w) hm::rk4Steps(dt, nsteps, sys, state);
  hm::synchronize(sys);
  comm.allGatherScalar(0.0);                                // barrier: the timed regions start together
  const auto t0 = std::chrono::steady_clock::now();
  for (int l = 0; l < launches; ++l) hm::rk4Steps(dt, nsteps, sys, state);
  hm::synchronize(sys);
  const double el_me = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double el = 0.0;
  for (double e : comm.allGatherScalar(el_me)) el = e > el ? e : el;     // the job's time is its slowest rank's
  int64_t flagged_me = 0;
  for (int32_t st : state.download_status()) flagged_me += (st != 0);
  int64_t flagged = 0;
  for (double f : comm.allGatherScalar((double)flagged_me)) flagged += (int64_t)f;
  const auto g0 = std::chrono::steady_clock::now();
  const hm::DevicePhase all = comm.allGather(state, std::vector<int64_t>((size_t)world, B));      // the path's one collective
  const double gather_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count() * 1e3;
  if (rank == 0) {
    print_line((double)world * (double)B * nsteps * launches, el, world, launches, warmup, B, nsteps, gather_ms, "allgather_ms_rccl", flagged,
               "one process per GPU, RCCL all-gather through hamk_comm_*");
    if (dump > 0) {
      const hm::Phase h = all.download();
      for (int64_t i = 0; i < dump && i < h.B; ++i)
        std::printf("traj %" PRId64 " q = %.17g %.17g p = %.17g %.17g H0 = 0\n", i, h.positions[(size_t)i], h.positions[(size_t)h.B + i],
                    h.momenta[(size_t)i], h.momenta[(size_t)h.B + i]);
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  int64_t B = 1 << 20; int nsteps = 100, launches = 50, warmup = 5, gpus = 1; int64_t dump = 0;
  int world = 0, rank = 0, device = -1; std::string id_file;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--batch")) B = std::atoll(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--nsteps")) nsteps = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--launches")) launches = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--warmup")) warmup = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--dump-first")) dump = std::atoll(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--world")) world = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--rank")) rank = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--id-file")) id_file = argv[i + 1];
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  const double dt = 0.01;
  const double PI = 3.14159265358979323846;
  try {
    if (world > 0) {
      if (rank < 0 || rank >= world || id_file.empty()) { std::fprintf(stderr, "--world W needs --rank R (0 <= R < W) and --id-file F\n"); return 2; }
      return run_rank(world, rank, device, id_file, B, nsteps, launches, warmup, dump);
    }
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
    print_line(units, el, gpus, launches, warmup, B, nsteps, gather_ms, "gather_ms_to_host", flagged, "one process, peer copies");
    for (int64_t i = 0; i < dump && i < all.B; ++i)
      std::printf("traj %" PRId64 " q = %.17g %.17g p = %.17g %.17g H0 = %.17g\n", i, all.positions[(size_t)i], all.positions[(size_t)all.B + i],
                  all.momenta[(size_t)i], all.momenta[(size_t)all.B + i], h0.empty() ? 0.0 : h0[(size_t)i]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "hamk_bench: %s\n", e.what());
    return 1;
  }
  return 0;
}

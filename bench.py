#!/usr/bin/env python3
"""bench.py -- RK4 phase-space steps/sec over an ensemble (BASELINE.json metric).

One bench "step" = one launch of the hot path (`hamk_rk4_steps`) advancing this rank's
whole ensemble shard by --rk4-per-step classic RK4 steps; inputs are resident in HBM
before the timed region.  Workload at N=1 = BASELINE.json configs[1]: double pendulum
(System 4 2, Examples.hs:75-94), 1,048,576 random Phase-2 initial conditions, fp64.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python bench.py --gpus N --steps K --warmup W          (starts its N ranks itself: one process per GPU over RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (the driver's launch; --gpus must equal WORLD_SIZE)

Multi-GPU: the ensemble shards by contiguous global index range (weak scaling: every rank
owns --batch trajectories); there is no data-path collective; one RCCL all_gather of the
final state runs after the timed region (reported as gather_ms).

Prints ONE JSON line on rank 0.

`roofline` has two readings, and says which one binds:
  * the north-star yardstick: `achieved` = 32*n algorithmic bytes per
    trajectory-step -- charged per RK4 step although --rk4-per-step steps are fused per launch and
    the state stays in registers in between -- divided by the kernel's average duration measured
    with HIP events on the launch stream.  It is NOTIONAL bandwidth: `roofline.hbm_physical` is what
    a launch really moves (state in + state out + status), its GB/s, and the PMC figure it matches
    (`traffic`, static from profiles/, labelled as such);
  * `roofline.fp64`: what physically bounds the kernel.  fp64 flops and VALU
    instructions per RK4 step are COUNTED from the code object of this system (llvm-objdump of its
    stepping loop, scripts/isa_stats.py), giving achieved TFLOP/s against the 78.6 TFLOP/s vector
    fp64 peak and the fraction of VALU issue slots used (every VALU instruction of a wavefront
    occupies its SIMD's 16 lanes for 4 cycles; 1024 SIMDs x 2.4 GHz nominal).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hamilton_amd import api, ensemble, examples  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6        # vector fp64 = 1/2 of the 157.3 TF fp32 vector peak
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
NOMINAL_HZ = 2.4e9
# fused steps per launch = the config's nsteps (C2-C4: 1000, C5: 200)
DEFAULT_RK4_PER_STEP = {"chain8": 200, "chain16": 200, "chain32": 200}
BASELINE_CONFIG = {"doublePendulum": ("configs[1]", 1 << 20), "twoBody": ("configs[2]", 1 << 20), "spring": ("configs[2]", 1 << 20),
                   "threeBodyPolar": ("configs[3]", 1 << 18), "chain8": ("configs[4]", 1 << 16), "chain16": ("configs[4]", 1 << 16),
                   "chain32": ("configs[4]", 1 << 16)}


class ClockSampler:
    """Shader clock of one GPU during the timed region, from the driver's own table (sysfs hwmon freq1_input, else pp_dpm_sclk's `*` level, of the card whose PCI address is the CUDA device's),
    sampled from a host thread every few milliseconds.  The chip clocks to its power budget (fp64-dense kernels sustain ~1.9-2.0
    of the nominal 2.4 GHz), so issue fractions priced at the nominal clock understate what the SIMDs really did."""
    def __init__(self, index: int):
        import glob
        self.path = None
        try:                                                     # the card whose PCI address is cuda:index's
            bdf = None
            try:
                pr = torch.cuda.get_device_properties(index)
                bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            except Exception:                                    # noqa: BLE001 -- a torch without the pci_* fields: ask the HIP runtime
                import ctypes
                hip = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]      # the runtime ALREADY mapped (never a second one)
                lib = ctypes.CDLL(hip[0])
                buf = ctypes.create_string_buffer(64)
                if lib.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
                    bdf = buf.value.decode().lower()
            for card in sorted(glob.glob("/sys/class/drm/card*/device")) if bdf else []:
                if os.path.realpath(card).endswith(bdf):
                    hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input")))      # current gfx clock, Hz
                    self.path = hw[0] if hw else os.path.join(card, "pp_dpm_sclk")
                    break
        except Exception:                                        # noqa: BLE001 -- no sysfs, older torch: no clock figure
            self.path = None
        self.samples, self._stop, self._t = [], False, None

    def _read(self):
        try:
            txt = open(self.path).read()
            if self.path.endswith("freq1_input"):
                return float(txt.strip()) / 1e6
            for ln in txt.splitlines():
                if ln.strip().endswith("*"):
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:                                        # noqa: BLE001
            return None
        return None

    def _run(self):
        while not self._stop:
            v = self._read()
            if v:
                self.samples.append(v)
            time.sleep(0.002)

    def start(self):
        if self.path:
            import threading
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()

    def stop(self):
        self._stop = True
        if self._t:
            self._t.join(timeout=1.0)
        return (sum(self.samples) / len(self.samples)) if self.samples else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node.  Under torch.distributed.run it must equal WORLD_SIZE; without a launcher "
                         "(no WORLD_SIZE in the environment) N > 1 makes bench.py start N ranks ITSELF (torch.distributed.run on 127.0.0.1)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--system", default="doublePendulum")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default: the BASELINE.json size of --system)")
    ap.add_argument("--rk4-per-step", type=int, default=None,
                    help="RK4 steps fused into one launch (default: SURVEY 8d's nsteps of the config: 1000, chains 200)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every rank owns --batch trajectories; "
                         "strong: --batch is the whole ensemble, sharded contiguously over the ranks")
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--integrator", choices=["rk4", "stepham"], default="rk4",
                    help="rk4: the BASELINE metric (hamk_rk4_steps).  stepham: the reference's own stepper "
                         "(adaptive RKF45, Hamilton.hs:390-402), one stepHam(dt) per launch -- secondary figure")
    ap.add_argument("--calls-per-launch", type=int, default=1,
                    help="--integrator stepham: consecutive stepHam(dt) calls fused into one launch (hamk_step_ham_iterate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the all-cores baseline leg")
    ap.add_argument("--no-isa", action="store_true", help="skip the instruction count of the stepping loop (roofline.fp64)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the multi-GPU code path "
                         "(barrier, max-over-ranks, final all_gather) on a 1-GPU box")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of a multi-rank run: nccl (= RCCL over xGMI; the driver's 1/2/4/8-GPU runs) or gloo -- "
                         "collectives on host copies, ranks may SHARE a GPU (rank r uses cuda:(r mod device count)): the whole "
                         "WORLD_SIZE > 1 path on a 1-GPU box (tests/test_gpu_configs.py)")
    ap.add_argument("--dump-state", default=None, help="rank 0 writes the final state of the WHOLE ensemble (after the gather) to this .npz")
    ap.add_argument("--drift-tol", type=float, default=1e-3,
                    help="per-launch energy check of the timed launches (HAMK_ST_DRIFT); 0 = plain hamk_rk4_steps")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = BASELINE_CONFIG.get(a.system, (None, 1 << 20))[1]
    if a.rk4_per_step is None:
        a.rk4_per_step = DEFAULT_RK4_PER_STEP.get(a.system, 1000)
    return a


def cpu_baseline_leg(spec, s, dt, target_seconds, own_length=None):
    """Oracle (C restatement of the reference algorithm) on this box's host cores, on a bounded
    sample of the same workload -- all cores (`value`) and one thread (`single_thread`), SURVEY.md
    section 8d; also yields the metric's `max |dphase| vs CPU ref`."""
    from oracle import oracle
    o = oracle.OracleSystem(spec)
    cores = oracle.max_threads()
    nsteps = 100
    probe_B = 8
    q, qd = examples.sample_config(spec, 0, probe_B)
    p = o.to_phase_batch(q, qd)
    t0 = time.perf_counter()
    o.rk4_steps_batch(q, p, dt, 5, threads=1)
    rate1 = probe_B * 5 / (time.perf_counter() - t0)            # trajectory-steps/s of one thread
    S1 = int(min(1 << 16, max(8, rate1 * min(6.0, target_seconds / 2) / nsteps)))
    q, qd = examples.sample_config(spec, 0, S1)
    p = o.to_phase_batch(q, qd)
    t0 = time.perf_counter()
    o.rk4_steps_batch(q, p, dt, nsteps, threads=1)
    el1 = time.perf_counter() - t0
    probe_S = 16 * cores                                          # all cores: measure the rate, do not assume the scaling;
    while True:                                                   # grow the probe until thread start-up no longer dominates it
        q, qd = examples.sample_config(spec, 0, probe_S)
        p = o.to_phase_batch(q, qd)
        t0 = time.perf_counter()
        o.rk4_steps_batch(q, p, dt, 5)
        el_probe = time.perf_counter() - t0
        rate_all = probe_S * 5 / el_probe
        if el_probe > 0.4 or probe_S >= (1 << 19):
            break
        probe_S *= 4
    S = int(min(1 << 20, max(32, rate_all * target_seconds / nsteps)))     # (a 64-link chain on a few host cores: 32, not more)
    S -= S % 32
    q, qd = examples.sample_config(spec, 0, S)
    p = o.to_phase_batch(q, qd)
    t0 = time.perf_counter()
    oq, op = o.rk4_steps_batch(q, p, dt, nsteps)
    el = time.perf_counter() - t0
    # same sample through the HIP path (outside any timed region)
    tq, tp = torch.from_numpy(q).cuda(), torch.from_numpy(p).cuda()
    one = api.rk4Steps(dt, 1, s, api.Phase(tq, tp))
    o1q, o1p = o.rk4_steps_batch(q, p, dt, 1)
    ph = api.rk4Steps(dt, nsteps, s, api.Phase(tq, tp), drift_tol=1e-6)
    st = s.last_status.cpu().numpy()
    torch.cuda.synchronize()
    d1 = max(np.max(np.abs(one.positions.cpu().numpy() - o1q)), np.max(np.abs(one.momenta.cpu().numpy() - o1p)))

    def lane_report(gq, gp, rq, rp, status):
        """max |dphase| per lane over ALL lanes (the metric's second half, unfiltered), where it sits, and -- beside it,
        never instead of it -- the figure over the lanes the launch did not flag at drift 1e-6."""
        dall = np.maximum(np.abs(gq - rq).max(0), np.abs(gp - rp).max(0))
        dall = np.where(np.isfinite(dall), dall, np.inf)
        worst = int(np.argmax(dall))
        calm = status == 0
        return dall, {"max_all_lanes": float(dall.max()), "worst_lane": worst, "worst_lane_status": int(status[worst]),
                      "median_all_lanes": float(np.median(dall)), "p99_all_lanes": float(np.quantile(dall, 0.99)),
                      "max_unflagged_lanes": float(dall[calm].max()) if calm.any() else None,
                      "unflagged_lanes": int(calm.sum()), "lanes": int(dall.size)}

    gq, gp = ph.positions.cpu().numpy(), ph.momenta.cpu().numpy()
    dall, rep = lane_report(gq, gp, oq, op, st)
    # energy drift of the worst lane over these steps, from the hamiltonian outside the kernel: says whether its
    # |dphase| is roundoff amplified by an under-resolved / close-encounter member or a disagreement worth chasing
    w = rep["worst_lane"]
    hw0 = o.observe_batch(q[:, w:w + 1], p[:, w:w + 1])[2][0]
    hw1 = o.observe_batch(oq[:, w:w + 1], op[:, w:w + 1])[2][0]
    rep["worst_lane_rel_energy_drift_cpu"] = float(abs(hw1 - hw0) / max(1.0, abs(hw0)))
    base = {"value": S * nsteps / el, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
            "sample": f"{S} trajectories x {nsteps} RK4 steps of the same seeded ensemble, "
                      f"oracle/libhamk_oracle.so (OpenMP, {cores} threads), {el:.1f} s",
            "single_thread": {"value": S1 * nsteps / el1, "cores": 1,
                              "sample": f"{S1} trajectories x {nsteps} RK4 steps, one thread, {el1:.1f} s"}}
    parity = {"max_abs_dphase_1_step": float(d1),
              f"max_abs_dphase_{nsteps}_steps_all_lanes": rep["max_all_lanes"],
              f"max_abs_dphase_{nsteps}_steps": rep["max_unflagged_lanes"],
              f"median_abs_dphase_{nsteps}_steps_all_lanes": rep["median_all_lanes"],
              f"p99_abs_dphase_{nsteps}_steps_all_lanes": rep["p99_all_lanes"],
              "worst_lane": {"index": rep["worst_lane"], "status": rep["worst_lane_status"],
                             "rel_energy_drift_cpu": rep["worst_lane_rel_energy_drift_cpu"]},
              "trajectories": S, f"trajectories_in_max_at_{nsteps}_steps": rep["unflagged_lanes"],
              "note": "the all-lanes figure is the metric; the second max is over the lanes the launch did not flag "
                      "(HAMK_ST_DRIFT at 1e-6 over these steps: under-resolved fast members / close encounters, where a fixed "
                      "step amplifies roundoff without bound)",
              "reference": "oracle (CPU restatement; reference Haskell toolchain absent)"}
    # Where the configured dt leaves most of the ensemble unresolved (BASELINE config 5 at SURVEY's dt = 0.005: energy
    # error O(1) on every member) the 100-step figure above compares two chaotic amplifications of roundoff.  A second
    # sample at a RESOLVED step -- dt halved until >= 90 % of the lanes keep their energy to 1e-6 over the 100 steps --
    # gives the configuration a parity number that means something.
    if rep["unflagged_lanes"] < 0.9 * S:
        S2 = max(32, S // 4)
        q2, p2 = q[:, :S2].copy(), p[:, :S2].copy()
        t2q, t2p = torch.from_numpy(q2).cuda(), torch.from_numpy(p2).cuda()
        dt2, frac = dt, 0.0
        for _ in range(12):
            dt2 *= 0.5
            api.rk4Steps(dt2, nsteps, s, api.Phase(t2q, t2p), drift_tol=1e-6)
            frac = float((s.last_status == 0).double().mean())
            if frac >= 0.9:
                break
        ph2 = api.rk4Steps(dt2, nsteps, s, api.Phase(t2q, t2p), drift_tol=1e-6)
        st2 = s.last_status.cpu().numpy()
        r2q, r2p = o.rk4_steps_batch(q2, p2, dt2, nsteps)
        _, rep2 = lane_report(ph2.positions.cpu().numpy(), ph2.momenta.cpu().numpy(), r2q, r2p, st2)
        parity["resolved_step"] = {"dt": dt2, "dt_ratio": dt / dt2, "steps": nsteps, "trajectories": S2,
                                   "unflagged_frac_at_1e-6": frac,
                                   f"max_abs_dphase_{nsteps}_steps_all_lanes": rep2["max_all_lanes"],
                                   f"max_abs_dphase_{nsteps}_steps": rep2["max_unflagged_lanes"],
                                   f"median_abs_dphase_{nsteps}_steps_all_lanes": rep2["median_all_lanes"]}
    # ... and at the launch's OWN length (SURVEY 8d nsteps: 1000 for C2-C4, 200 for C5 -- what one timed launch does), on a
    # small sample: all-lanes figure, the chaotic growth of rounding over the config's whole time span included
    if own_length and own_length != nsteps:
        S3 = int(min(S, max(32, S * nsteps // (4 * own_length))))
        S3 -= S3 % 32
        q3, p3 = q[:, :S3].copy(), p[:, :S3].copy()
        t0 = time.perf_counter()
        r3q, r3p = o.rk4_steps_batch(q3, p3, dt, own_length)
        el3 = time.perf_counter() - t0
        ph3 = api.rk4Steps(dt, own_length, s, api.Phase(torch.from_numpy(q3).cuda(), torch.from_numpy(p3).cuda()), drift_tol=1e-6)
        st3 = s.last_status.cpu().numpy()
        _, rep3 = lane_report(ph3.positions.cpu().numpy(), ph3.momenta.cpu().numpy(), r3q, r3p, st3)
        parity["own_length"] = {"steps": own_length, "trajectories": S3, "cpu_seconds": el3,
                                f"max_abs_dphase_{own_length}_steps_all_lanes": rep3["max_all_lanes"],
                                f"median_abs_dphase_{own_length}_steps_all_lanes": rep3["median_all_lanes"],
                                f"p99_abs_dphase_{own_length}_steps_all_lanes": rep3["p99_all_lanes"],
                                f"max_abs_dphase_{own_length}_steps": rep3["max_unflagged_lanes"],
                                "unflagged_lanes": rep3["unflagged_lanes"],
                                "note": "one timed launch's worth of steps; all lanes, then the lanes not flagged at drift 1e-6"}
    return base, parity


def probe_reference_toolchain():
    """SURVEY.md section 8d, CPU-baseline step (1): is the reference's own toolchain (GHC + cabal + GSL, for `ad`,
    hmatrix and hmatrix-gsl) on this box?  Probed, not assumed; the real ad+hmatrix path is timed only if it is."""
    import shutil
    import subprocess
    probed = []
    for tool, args in (("ghc", ["--version"]), ("cabal", ["--version"]), ("stack", ["--version"]), ("gsl-config", ["--version"])):
        path = shutil.which(tool)
        ver = None
        if path:
            try:
                ver = subprocess.run([path] + args, capture_output=True, text=True, timeout=20).stdout.strip().splitlines()[0]
            except Exception as e:                               # noqa: BLE001
                ver = f"present, not runnable: {e!r}"
        probed.append({"tool": tool, "path": path, "version": ver})
    have = {r["tool"]: r["path"] is not None for r in probed}
    available = have["ghc"] and (have["cabal"] or have["stack"]) and have["gsl-config"]
    out = {"available": bool(available), "probed": probed}
    if not available:
        out["note"] = "reference_haskell: unavailable (toolchain absent) -- cpu_baseline is the C restatement (kind: port)"
    else:
        out.update(time_reference_haskell())
    return out


def time_reference_haskell(timeout_s: float = 900.0):
    """SURVEY.md section 8d, CPU-baseline step (1), for a box that HAS the reference's toolchain: build
    bindings/haskell/bench (C1.hs: the reference's double pendulum through the reference's own `stepHam` / `hamEqs`, i.e. ad +
    hmatrix + hmatrix-gsl) against a checkout of mstksg/hamilton (HAMILTON_SRC, default /root/reference) with
    `cabal build --offline` in a scratch copy, run it, and return its JSON line as `measured`.  Mechanical on purpose: any
    failure (no Hackage packages offline, no checkout) is reported with the tail of cabal's output, never raised.  This image
    has no GHC: the function has never run here."""
    import shutil
    import subprocess
    import tempfile
    src = os.environ.get("HAMILTON_SRC", "/root/reference")
    bench_dir = os.path.join(ROOT, "bindings", "haskell", "bench")
    if not os.path.exists(os.path.join(src, "hamilton.cabal")):
        return {"note": f"toolchain present, but no checkout of mstksg/hamilton at {src} (set HAMILTON_SRC)"}
    work = tempfile.mkdtemp(prefix="hamk_c1_")
    try:
        for f in ("C1.hs", "hamilton-bench.cabal", "cabal.project"):
            shutil.copy(os.path.join(bench_dir, f), work)
        with open(os.path.join(work, "cabal.project.local"), "w") as fh:
            fh.write(f"packages: {src}\n")
        t0 = time.perf_counter()
        b = subprocess.run(["cabal", "build", "--offline", "c1"], cwd=work, capture_output=True, text=True, timeout=timeout_s)
        if b.returncode != 0:
            return {"note": "toolchain present; `cabal build --offline c1` failed (Hackage dependencies of hamilton not available offline?)",
                    "cabal_tail": (b.stdout + b.stderr)[-1500:]}
        exe = subprocess.run(["cabal", "list-bin", "c1"], cwd=work, capture_output=True, text=True, timeout=120).stdout.strip()
        r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"note": "c1 built but did not produce its line", "stderr_tail": r.stderr[-800:]}
        return {"note": "measured: the reference's own ad + hmatrix + hmatrix-gsl path on this box (bindings/haskell/bench/C1.hs)",
                "measured": json.loads(line[-1]), "build_seconds": time.perf_counter() - t0}
    except Exception as e:                                       # noqa: BLE001 -- reported, not raised
        return {"note": f"toolchain present; building / running the Haskell bench failed: {e!r}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def c1_leg(spec, s):
    """BASELINE config 1 ("plumbing"): ONE trajectory, 1000 x stepHam 0.01 from the reference's initial
    state (Examples.hs:250-267), through the host-pointer path of the C ABI and through the CPU oracle."""
    from oracle import oracle
    o = oracle.OracleSystem(spec)
    q, p = np.array(spec.q0, dtype=np.float64), np.zeros(spec.n)
    ph = api.toPhase(s, api.Config(q, np.array(spec.qd0, dtype=np.float64)))
    q, p = np.asarray(ph.positions, dtype=np.float64), np.asarray(ph.momenta, dtype=np.float64)
    oq, op = q.copy(), p.copy()
    for _ in range(20):                                           # warm: pinned arena, module, self-check
        api.stepHam(0.01, s, api.Phase(q, p))
    t0 = time.perf_counter()
    gq, gp = q, p
    for _ in range(1000):
        ph = api.stepHam(0.01, s, api.Phase(gq, gp))
        gq, gp = ph.positions, ph.momenta
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(1000):
        oq, op = o.step_ham(0.01, oq, op)
    t_cpu = time.perf_counter() - t0
    # the same 1000 calls as ONE launch (hamk_step_ham_iterate: `iterate (stepHam dt)`, README.md:150): bit-identical
    api.iterateStepHam(0.01, 10, s, api.Phase(q, p))
    t0 = time.perf_counter()
    it = api.iterateStepHam(0.01, 1000, s, api.Phase(q, p))
    t_it = time.perf_counter() - t0
    same = bool(np.array_equal(it.positions, gq) and np.array_equal(it.momenta, gp))
    return {"workload": "doublePendulum, 1 trajectory, 1000 x stepHam 0.01 (BASELINE.json configs[0])",
            "gpu_us_per_call": t_gpu * 1e3, "gpu_us_per_call_fused": t_it * 1e3, "cpu_oracle_us_per_call": t_cpu * 1e3,
            "one_launch_bit_identical_to_1000_calls": same,
            "max_abs_dphase_after_1000_calls": float(max(np.max(np.abs(gq - oq)), np.max(np.abs(gp - op)))),
            "note": "gpu_us_per_call: one launch + one synchronisation per call through the Python mirror of the C ABI (what BASELINE "
                    "config 1 measures; comparable round over round); _fused: the same 1000 calls as ONE launch of hamk_step_ham_iterate"}


def stepham_bench(a, s, spec, dt, state, dist, dev, rank, world):
    """Secondary: the reference's OWN stepper over the ensemble -- stepHam(dt) calls/s (GSL-semantics adaptive RKF45 per
    lane, Hamilton.hs:390-402, :443-448).  One bench step = one launch = one stepHam(dt) of every trajectory."""
    ph = state
    s.describe_batch(ph.positions.shape[1])                  # the specialisation a launch over THIS shard uses
    K = max(1, a.calls_per_launch)
    step = (lambda x: api.stepHam(dt, s, x, inplace=True)) if K == 1 else (lambda x: api.iterateStepHam(dt, K, s, x, inplace=True))
    for _ in range(a.warmup):
        ph = step(ph)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nsub_sum = torch.zeros(ph.positions.shape[1], dtype=torch.int64, device=dev)
    wave_max_sum = 0.0
    lanes = s.lanes_per_trajectory
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        ph = step(ph)
    ev1.record()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K                      # per stepHam call
    kernel_s = ev0.elapsed_time(ev1) * 1e-3 / max(1, a.steps) / K
    # divergence: a wavefront runs until its slowest member is done (outside the timed region: one more call, counted)
    probe = api.stepHam(dt, s, api.Phase(ph.positions.clone(), ph.momenta.clone()))
    del probe
    nsub = s.last_nsub.to(torch.float64)
    per_wave = 64 // lanes
    Bw = (nsub.numel() // per_wave) * per_wave
    wmax = nsub[:Bw].reshape(-1, per_wave).amax(1)
    if rank == 0:
        B = a.batch
        n = spec.n
        calls_per_s = world * B * a.steps / el
        attempts_per_s = calls_per_s * float(nsub.mean())
        out = {"metric": "stepHam calls/sec (ensemble, adaptive RKF45 GSL semantics)",
               "value": calls_per_s, "unit": "trajectory-stepHam/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
               "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
               "data": "synthetic (per-index splitmix64 initial conditions, seed 20241008)",
               "config": {"workload": f"{a.system} (System {spec.m} {spec.n}) ensemble, stepHam dt={dt}", "trajectories_per_gpu": B,
                          "stepham_calls_per_launch": K,
                          "kernel_path": ("four lanes per trajectory" if lanes == 4 else "wave-cooperative") if lanes > 1 else "one trajectory per lane", "gsl_api": s.gsl_api},
               "mean_substeps": float(nsub.mean()), "max_substeps": float(nsub.max()),
               "rhs_evals_per_s": attempts_per_s * 6,
               "divergence": {"mean_substeps_per_lane": float(nsub.mean()), "mean_of_wave_max_substeps": float(wmax.mean()),
                              "lane_utilisation": float(nsub[:Bw].mean() / wmax.mean()),
                              "note": "a wavefront executes max-over-its-lanes attempts; utilisation = mean / mean-of-wave-max"}}
        alg = 32.0 * n * B                                   # a launch reads and writes one Phase per trajectory
        fp64 = {"peak_tflops": FP64_PEAK_TFLOPS, "lanes_per_trajectory": lanes}
        if not a.no_isa:
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import isa_stats
                isa = isa_stats.rkf45_attempt_stats(spec, s)
            except Exception as e:                           # noqa: BLE001
                isa, fp64["isa_error"] = None, repr(e)
            if isa:
                wave_attempts_per_s = float(wmax.sum()) * a.steps / (kernel_s * a.steps)      # per GPU
                fp64.update(isa)
                fp64["wave_attempts_per_s"] = wave_attempts_per_s
                fp64["valu_issue_frac"] = wave_attempts_per_s * isa["valu_per_wave_attempt"] * 4.0 / (N_SIMD * NOMINAL_HZ)
                fp64["achieved_tflops_useful"] = lanes * B * float(nsub.mean()) / kernel_s * isa["valu_f64_per_wave_attempt"] * 2 / 1e12
                fp64["note"] = ("valu_issue_frac counts what the wavefronts EXECUTE (wave-max attempts); achieved_tflops_useful counts "
                                "fp64 instructions x 2 on the attempts the lanes needed (an upper bound: not every fp64 instruction is an FMA)")
        out["roofline"] = {"bound": "hbm", "achieved": alg / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg / kernel_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "hamk_rkf45_k", "kernel_ms": kernel_s * 1e3,
                           "algorithmic_bytes_per_launch": alg, "physically_binding": "fp64-valu (divergent trip counts)", "fp64": fp64}
        if world == 1 and not a.no_cpu_baseline:
            from oracle import oracle
            o = oracle.OracleSystem(spec)
            o.gsl_api = s.gsl_api
            cores = oracle.max_threads()
            S = 64 * cores
            q, qd = examples.sample_config(spec, 0, S)
            p = o.to_phase_batch(q, qd)
            t0 = time.perf_counter()
            o.step_ham_batch(q, p, dt)
            probe_t = time.perf_counter() - t0
            S = int(min(1 << 20, max(64, S * min(64.0, a.cpu_seconds / max(probe_t, 1e-6)))))
            q, qd = examples.sample_config(spec, 0, S)
            p = o.to_phase_batch(q, qd)
            t0 = time.perf_counter()
            oq, op, ons = o.step_ham_batch(q, p, dt)
            elc = time.perf_counter() - t0
            g = api.stepHam(dt, s, api.Phase(torch.from_numpy(q).cuda(), torch.from_numpy(p).cuda()))
            gns = s.last_nsub.cpu().numpy()
            same = gns == ons
            dq = np.maximum(np.abs(g.positions.cpu().numpy() - oq).max(0), np.abs(g.momenta.cpu().numpy() - op).max(0))
            out["cpu_baseline"] = {"value": S / elc, "unit": "trajectory-stepHam/s", "cores": cores, "kind": "port",
                                   "sample": f"{S} trajectories x one stepHam({dt}) of the same seeded ensemble, oracle (OpenMP, {cores} threads), {elc:.1f} s",
                                   "reference_haskell": probe_reference_toolchain()}
            out["parity"] = {"identical_substep_counts_frac": float(same.mean()), "max_abs_dphase_all_lanes": float(dq.max()),
                             "max_abs_dphase_lanes_with_identical_counts": float(dq[same].max()) if same.any() else None,
                             "trajectories": S, "reference": "oracle (CPU restatement of GSL rkf45 + standard controller; reference toolchain absent)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(gpus: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here -- the same command the driver
    uses (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1, a free port) -- and hand its exit code back.
    Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus or 1) > 1:
        raise SystemExit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus is not None and a.gpus != world:           # a line that says n_gpus = N must have run N ranks
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}, or drop the launcher "
                         f"and let --gpus start the ranks")
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.dist_backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())      # ranks may share a GPU; collectives run on host copies
            dist.init_process_group(backend="gloo")
        else:
            if world > torch.cuda.device_count():
                raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {torch.cuda.device_count()} "
                                 f"(--dist-backend gloo lets ranks share a GPU)")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if (dist is not None and a.dist_backend == "gloo") else dev        # where collective buffers live

    spec = examples.get(a.system)
    dt = a.dt if a.dt is not None else spec.dt
    s = api.system_from_spec(spec)                       # tape -> hiprtc gfx950 module (outside timed region)
    n = spec.n
    # this rank's shard of the global ensemble (contiguous global indices, per-index RNG)
    if a.scaling == "strong":
        if a.batch % world:
            raise SystemExit("--scaling strong needs --batch divisible by the number of ranks (equal shards for the gather)")
        lo, hi = ensemble.shard_bounds(a.batch, world, rank)
    else:
        lo, hi = ensemble.weak_bounds(a.batch, rank)
    B = hi - lo
    if a.scaling == "strong":
        # every shard on the mapping the library picks for the WHOLE ensemble: any G reproduces the 1-GPU bits
        # (hamk_options::ensemble_size; with the choice left per launch a small shard of a mid-size system changes kernels)
        ensemble.pin_for_ensemble(s, a.batch)
    # initial Configs drawn ON THE DEVICE from the global trajectory index (hamk_sample_batch): no host array,
    # no scatter; the same bits as examples.sample_config(spec, lo, B) on any shard layout
    cfg0 = api.sampleConfig(s, spec.q_box, spec.qd_box, lo, B, examples.SEED, dev)
    q, qd = cfg0.positions, cfg0.velocities
    ph = api.toPhase(s, api.Config(q, qd))               # momenta on device (Hamilton.hs:279-284)
    q, p = ph.positions.clone(), ph.momenta.clone()
    h0 = api.hamiltonian(s, api.Phase(q, p)).clone()
    state = api.Phase(q, p)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.integrator == "stepham":
        return stepham_bench(a, s, spec, dt, state, dist, dev, rank, world)
    # every launch checks its own energy invariant (two extra hamiltonian evaluations per LAUNCH of
    # rk4-per-step steps: HAMK_ST_DRIFT, SURVEY 8d C4 "flag close encounters via status"); the status
    # words of the timed launches are OR-ed on the device (one 4 B/trajectory elementwise op per launch)
    DRIFT_TOL = a.drift_tol
    status_or = torch.zeros(B, dtype=torch.int32, device=dev)
    status_or |= torch.zeros_like(status_or)              # torch loads an op's code object at its first use (measured:
    for _ in range(a.warmup):                             # 15 ms for this bitwise-or -- inside the timed region otherwise)
        api.rk4Steps(dt, a.rk4_per_step, s, state, inplace=True, drift_tol=DRIFT_TOL)
        status_or |= s.last_status
    status_or.zero_()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clk = ClockSampler(local_rank)
    clk.start()
    t0 = time.perf_counter()
    ev0.record()                                          # kernels launch on torch's current stream
    for _ in range(a.steps):
        api.rk4Steps(dt, a.rk4_per_step, s, state, inplace=True, drift_tol=DRIFT_TOL)
        status_or |= s.last_status
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sclk_mhz = clk.stop()
    barrier()
    kernel_s = ev0.elapsed_time(ev1) * 1e-3 / max(1, a.steps)     # avg launch duration, HIP events
    if dist is not None:
        t = torch.tensor([elapsed, kernel_s], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_s = float(t[0]), float(t[1])

    bad = int(torch.count_nonzero(status_or))
    bad_drift = int(torch.count_nonzero(status_or & 16))
    h1 = api.hamiltonian(s, state)
    rel = (h1 - h0).abs() / h0.abs().clamp(min=1.0)
    drift = float(rel.max())
    # fixed-step RK4 through a near-singularity (close encounter of the gravitational systems)
    # shows up as lost energy conservation: counted here, from the hamiltonian before and after
    drift_flagged = int((rel > 1e-3).sum())

    gather_ms = None
    if dist is not None:                                  # the path's only collective: final gather over xGMI
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        gq, gp = ensemble.gather_state(state.positions.to(cdev), state.momenta.to(cdev), dist, world)
        assert gq.shape == (n, world * B)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        t = torch.tensor([bad, drift_flagged, bad_drift], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        bad, drift_flagged, bad_drift = int(t[0]), int(t[1]), int(t[2])
        t = torch.tensor([drift], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        drift = float(t[0])
    else:
        gq, gp = state.positions, state.momenta
    if rank == 0 and a.dump_state:
        np.savez(a.dump_state, q=gq.cpu().numpy(), p=gp.cpu().numpy())

    if rank == 0:
        total = a.batch if a.scaling == "strong" else world * a.batch
        units = total * a.rk4_per_step * a.steps         # trajectory-steps in the timed region
        value = units / elapsed
        per_gpu_rate = B * a.rk4_per_step / kernel_s
        alg_bytes = 32.0 * n                             # read + write one Phase n per step
        achieved = per_gpu_rate * alg_bytes / 1e9
        moved = alg_bytes * B + 4 * B                    # what one launch really reads and writes: state in/out + status
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.system}.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("trajectories") == B and rec.get("rk4_steps_per_launch") == a.rk4_per_step:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = f"STATIC: profiles/pmc_traffic_{a.system}.json ({rec.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')}), not measured by this run"
            except Exception:
                traffic = None
        s.describe_batch(B)                              # the specialisation a launch over this shard uses
        lanes_per_traj = s.lanes_per_trajectory          # 1, or the cooperative group size the module was built with
        wave = lanes_per_traj >= 16
        fp64 = {"per_gpu_steps_per_s": per_gpu_rate, "peak_tflops": FP64_PEAK_TFLOPS, "lanes_per_trajectory": lanes_per_traj}
        if not a.no_isa:
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import isa_stats
                isa = isa_stats.rk4_step_stats(spec, s)
            except Exception as e:                       # a missing llvm-objdump must not cost the bench line
                isa = None
                fp64["isa_error"] = repr(e)
            if isa:
                flops_step = isa["fp64_flops_per_lane_step"] * lanes_per_traj          # per trajectory-step
                wave_steps_per_s = per_gpu_rate * lanes_per_traj / 64.0
                fp64.update({
                    "flops_per_step": flops_step, "achieved_tflops": per_gpu_rate * flops_step / 1e12,
                    # a kernel that also issues v_mfma_f64 has two fp64 pipes to fill: vector 78.6 + matrix 78.6 TFLOP/s
                    "peak_tflops": FP64_PEAK_TFLOPS * (2.0 if isa["mfma_per_wave_step"] else 1.0),
                    "frac_of_peak": per_gpu_rate * flops_step / 1e12 / (FP64_PEAK_TFLOPS * (2.0 if isa["mfma_per_wave_step"] else 1.0)),
                    "valu_insts_per_wave_step": isa["valu_per_wave_step"], "valu_f64_insts_per_wave_step": isa["valu_f64_per_wave_step"],
                    "mfma_insts_per_wave_step": isa["mfma_per_wave_step"], "lds_insts_per_wave_step": isa["lds_per_wave_step"],
                    "scratch_insts_per_wave_step": isa["scratch_per_wave_step"],
                    "valu_issue_frac": wave_steps_per_s * isa["valu_per_wave_step"] * 4.0 / (N_SIMD * NOMINAL_HZ),
                    "valu_issue_frac_at_measured_clock": (wave_steps_per_s * isa["valu_per_wave_step"] * 4.0 / (N_SIMD * sclk_mhz * 1e6)) if sclk_mhz else None,
                    "sclk_mhz_during_timed_region": sclk_mhz, "sclk_source": clk.path,
                    "valu_issue_frac_note": "VALU wave-instructions/s x 4 cycles / (1024 SIMDs x clock): at the 2.4 GHz nominal clock, and at the shader clock the "
                                            "driver reported while the timed launches ran (sysfs hwmon freq1_input of the device's card, sampled every 2 ms; null where absent)",
                    "count_source": isa["source"], "loop": isa["loop_is"]})
        cfg_id, cfg_B = BASELINE_CONFIG.get(a.system, (None, None))
        out = {
            "metric": "RK4 phase-space steps/sec (ensemble)", "value": value, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (per-index splitmix64 initial conditions, seed 20241008)",
            "config": {"workload": f"{a.system} (System {spec.m} {spec.n}) ensemble"
                                   + (f", BASELINE.json {cfg_id}" if cfg_id and a.batch == cfg_B else ""),
                       "trajectories_per_gpu": B, "rk4_steps_per_launch": a.rk4_per_step, "dt": dt,
                       "kernel_path": "wave-cooperative" if wave else ("four lanes per trajectory" if lanes_per_traj == 4 else "one trajectory per lane"),
                       "parallelism": f"ensemble-shard x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the fraction that BINDS, next to the yardstick's (details in roofline.fp64)
                         "fp64_frac_of_peak": fp64.get("frac_of_peak"), "fp64_valu_issue_frac_at_measured_clock": fp64.get("valu_issue_frac_at_measured_clock"),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "hamk_rk4_steps_k", "kernel_ms": kernel_s * 1e3,
                         "algorithmic_bytes_per_trajectory_step": alg_bytes,
                         "algorithmic_bytes_per_launch": alg_bytes * B * a.rk4_per_step,
                         "physically_binding": "fp64-valu",
                         "note": "achieved/frac are the north-star yardstick of SURVEY 8d: 32n B CHARGED per fused RK4 step -- notional, "
                                 "not bandwidth; real HBM traffic is hbm_physical; the binding roofline is fp64",
                         "hbm_physical": {"bytes_moved_per_launch": moved, "GBps": moved / kernel_s / 1e9,
                                          "frac_of_peak": moved / kernel_s / 1e9 / HBM_PEAK_GBS},
                         "fp64": fp64},
            "status_flagged": bad, "status_flagged_drift": bad_drift, "drift_tol_per_launch": DRIFT_TOL,
            "max_rel_energy_drift": drift, "energy_drift_over_1e-3": drift_flagged,
        }
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            out["rccl"] = {"world": dist.get_world_size(), "backend": dist.get_backend(),
                           "collectives": "barrier + max-over-ranks timing + ONE all_gather of the final state after the timed region; none on the data path"}
        if world == 1 and not a.no_cpu_baseline:
            base, parity = cpu_baseline_leg(spec, s, dt, a.cpu_seconds, a.rk4_per_step)
            base["reference_haskell"] = probe_reference_toolchain()
            out["cpu_baseline"] = base
            out["parity"] = parity
            if a.system == "doublePendulum":
                out["config1"] = c1_leg(spec, s)
        def finite(x):                                      # strict JSON: a non-finite figure is reported as null
            if isinstance(x, float) and not np.isfinite(x):
                return None
            if isinstance(x, dict):
                return {k: finite(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [finite(v) for v in x]
            return x
        print(json.dumps(finite(out), allow_nan=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Turn gpurun_out/prof_<tag>_<system>/ (scripts/profile.sh) into the committed summaries under profiles/:
<tag>_<system>_kernel_stats.csv, <tag>_<system>_summary.json and pmc_traffic_<system>.json (the file
bench.py quotes, labelled static, as roofline.traffic when trajectories and steps per launch match)."""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
system = sys.argv[2] if len(sys.argv) > 2 else "doublePendulum"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{system}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "hamk_rk4_steps_k"

suffix = "" if (KERNEL == "hamk_rk4_steps_k" or system.endswith("_stepham")) else "_stepham"
shutil.copy(os.path.join(src, "stats", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_{system}{suffix}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "stats", f"{tag}_kernel_stats.csv")))}
summary = {"system": system, "kernel": KERNEL,
           "rocprofv3_kernel_stats": {k: stats[KERNEL][k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}}
bench = None
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj) and os.path.getsize(bj):
    bench = json.loads(open(bj).read())
    summary["bench_line_under_profiler"] = bench
    summary["hip_event_kernel_ms_same_run"] = bench["roofline"]["kernel_ms"]
    summary["rocprof_vs_hip_events"] = float(stats[KERNEL]["AverageNs"]) * 1e-6 / bench["roofline"]["kernel_ms"]

pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_mfma", "pmc_vmem", "pmc_ifetch", "pmc_wait"):
    path = os.path.join(src, name, f"{tag}_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"] == KERNEL:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
    if meta:
        summary["dispatch"] = {k: meta[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
summary["pmc_mean_per_launch"] = pmc
B = bench["config"]["trajectories_per_gpu"] if bench else None
K = bench["config"].get("rk4_steps_per_launch") if bench else None
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and bench:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B
    # (MI355X_MICROARCH.md "HBM"): doubled.  The kernel reads 16 n B and writes 16 n B + 4 B (status)
    # per trajectory per launch.
    n = int(bench["roofline"].get("algorithmic_bytes_per_trajectory_step", bench["roofline"].get("algorithmic_bytes_per_launch", 0) / max(1, B)) / 32)
    fetch = pmc["FETCH_SIZE"] * 1024 * 2
    write = pmc["WRITE_SIZE"] * 1024
    summary["hbm"] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                      "expected_read_bytes": 16 * n * B, "expected_write_bytes": 16 * n * B + 4 * B}
    if KERNEL == "hamk_rk4_steps_k":
      json.dump({"system": system, "trajectories": B, "rk4_steps_per_launch": K,
               "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
               "source": f"profiles/{tag}_{system}_summary.json: rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) / WRITE_SIZE, separate passes"},
              open(os.path.join(dst, f"pmc_traffic_{system}.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc and K:
    summary["valu_insts_per_wave_per_rk4_step"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"] / K
    # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count QUAD-cycles summed over every wavefront (or SIMD) of the chip,
    # GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs (MI355X_MICROARCH.md): a fraction needs the same unit on both
    # sides.  (Round 2 divided SQ_ACTIVE_INST_VALU by SQ_BUSY_CYCLES and called 7.8 a fraction.)
    if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
        summary["valu_active_frac_of_wave_cycles"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        summary["valu_pipe_busy_frac_of_1024_simds"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_WAIT_INST_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc:
        summary["wait_frac_of_wave_cycles"] = pmc["SQ_WAIT_INST_ANY"] / pmc["SQ_WAVE_CYCLES"]
for k_wait in ("SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"):
    if k_wait in pmc and "SQ_WAVE_CYCLES" in pmc:
        summary[k_wait.lower() + "_frac_of_wave_cycles"] = pmc[k_wait] / pmc["SQ_WAVE_CYCLES"]
if "SQ_LDS_IDX_ACTIVE" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    # LDS-array cycles summed over the 256 CUs against the kernel's cycles x 256 LDS units
    summary["lds_busy_frac_of_256_units"] = pmc["SQ_LDS_IDX_ACTIVE"] / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
    if "SQ_LDS_BANK_CONFLICT" in pmc:
        summary["lds_bank_conflict_frac_of_lds_cycles"] = pmc["SQ_LDS_BANK_CONFLICT"] / max(1.0, pmc["SQ_LDS_IDX_ACTIVE"])
if "SQ_INSTS_LDS" in pmc and "SQ_WAVES" in pmc and K:
    summary["lds_insts_per_wave_per_rk4_step"] = pmc["SQ_INSTS_LDS"] / pmc["SQ_WAVES"] / K
if KERNEL != "hamk_rk4_steps_k" and "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
    summary["valu_insts_per_wave_per_launch"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
json.dump(summary, open(os.path.join(dst, f"{tag}_{system}{suffix}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

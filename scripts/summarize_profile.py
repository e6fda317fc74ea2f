"""Turn gpurun_out/prof_<tag>/ (scripts/profile.sh) into the committed summaries under profiles/."""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "hamk_rk4_steps_k"

shutil.copy(os.path.join(src, "stats", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "stats", f"{tag}_kernel_stats.csv")))}
summary = {"kernel": KERNEL, "rocprofv3_kernel_stats": {k: stats[KERNEL][k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}}

pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
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
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B
    # (MI355X_MICROARCH.md "HBM"): doubled.  Calibration in our own access pattern: this kernel
    # reads exactly 4*8*B bytes and writes 4*8*B + 4*B bytes per launch (B = 2^20).
    fetch = pmc["FETCH_SIZE"] * 1024 * 2
    write = pmc["WRITE_SIZE"] * 1024
    B = 1 << 20
    summary["hbm"] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                      "expected_read_bytes": 32 * B, "expected_write_bytes": 32 * B + 4 * B,
                      "algorithmic_state_bytes_per_launch": 64 * B}
    json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
               "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
              open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
    summary["valu_insts_per_wave_per_rk4_step"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"] / 100.0
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj) and os.path.getsize(bj):
    summary["bench_line_under_profiler"] = json.loads(open(bj).read())
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

#!/usr/bin/env bash
# The 1/2/4/8-GPU curve of SURVEY.md section 8e with one command, for whoever has the 8-GPU node (the builder's boxes
# have one GPU; the driver's SCALE run measures the weak C2 line only):
#   weak   C2  doublePendulum, 2^20 trajectories PER GPU           (the driver's contract)
#   strong C4  threeBodyPolar, 2^18 trajectories over all GPUs     (BASELINE configs[3]: "1 -> 8 MI355X shard over xGMI")
#   strong C5  chain8 / chain16 / chain32, 2^16 trajectories       (BASELINE configs[4]: "8 x MI355X")
# One rank per GPU over RCCL (torch.distributed.run, rendezvous on 127.0.0.1); one JSON line per run, appended to
# $OUT (default gpurun_out/scale_curve.jsonl).  Strong runs state the whole ensemble's size to the library
# (hamk_options::ensemble_size), so every G reproduces the 1-GPU bits.  Expected shape: DESIGN.md "Multi-GPU".
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${OUT:-gpurun_out/scale_curve.jsonl}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
GPUS=${GPUS:-"1 2 4 8"}
mkdir -p "$(dirname "$OUT")"
run() {   # N, then bench.py arguments
  local n=$1; shift
  # bench.py starts its own ranks for --gpus N > 1 (torch.distributed.run, one process per GPU, 127.0.0.1)
  python bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline "$@" | tail -1 >> "$OUT"
}
for n in $GPUS; do
  run "$n" --system doublePendulum --scaling weak
  run "$n" --system threeBodyPolar --scaling strong --batch 262144
  for c in chain8 chain16 chain32; do
    run "$n" --system "$c" --scaling strong --batch 65536
  done
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = {}
print(f"{'workload':34s} {'scaling':7s} {'G':>2s} {'steps/s':>11s} {'eff':>5s} {'kernel ms':>9s} {'8d frac':>7s} {'gather ms':>9s}")
for r in rows:
    key = (r["config"]["workload"].split(" (")[0], r["scaling"])
    base.setdefault(key, r["value"] / r["n_gpus"] if r["scaling"] == "weak" else r["value"])
    ideal = base[key] * r["n_gpus"]
    print(f"{key[0]:34s} {key[1]:7s} {r['n_gpus']:2d} {r['value']:11.3e} {r['value'] / ideal:5.2f} {r['roofline']['kernel_ms']:9.3f} "
          f"{r['roofline']['frac']:7.3f} {r.get('gather_ms') or 0:9.2f}")
PY

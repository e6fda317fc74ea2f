mkdir -p gpurun_out
export HAMK_TEST_OVERRIDES=1
timeout 120 python scripts/host_buffer_rate.py --systems doublePendulum:262144:0.01,doublePendulum:524288:0.01 --steps 100,1000 > gpurun_out/hp_dev.jsonl 2>/dev/null
for P in 131072 262144 524288; do
  HAMK_HOST_PIECE=$P timeout 120 python scripts/host_buffer_rate.py --systems doublePendulum:1048576:0.01 --steps 32,100,1000 > gpurun_out/hp_$P.jsonl 2>/dev/null
done
python - <<EOF
import json,glob
for f in sorted(glob.glob("gpurun_out/hp_*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("/")[-1], d["B"], d["steps_per_call"], "ms dev %.3f host %.3f eq %s"%(d["ms_device"], d["ms_host"], d["bitwise_equal"]))
EOF

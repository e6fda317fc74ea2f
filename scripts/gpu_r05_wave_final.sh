export HAMK_CACHE_DIR=$PWD/.hamk_cache HAMK_TEST_OVERRIDES=1; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_wave.py -m gpu -q -x 2>&1 | tail -3
rm -f $O/r05_bench_wave.jsonl
for sys in dense24 dense32 chain48 chain64; do
  timeout 300 python bench.py --system $sys --batch 16384 --rk4-per-step 20 --steps 10 --warmup 2 --cpu-seconds 3 2> $O/bench_r05_${sys}.err | tail -1 >> $O/r05_bench_wave.jsonl
  tail -1 $O/r05_bench_wave.jsonl | head -c 140; echo
done
timeout 200 python scripts/wave_probe.py --systems=dense24,dense32,chain32 > $O/r05k_wave_probe.jsonl 2> $O/r05k.err
for sys in dense32 chain64; do
  HAMK_PROF_PASSES="stats fetch write sq lds mfma wait" timeout 400 bash scripts/profile.sh r05 $sys --batch 16384 --rk4-per-step 20 > /dev/null 2>&1
done

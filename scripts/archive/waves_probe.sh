#!/bin/bash
# GPU: the n > 32 RK4 kernel at one vs two wavefronts per SIMD (HAMK_RK4_WAVES), parity first.
export HAMK_CACHE_DIR=$PWD/.hamk_cache
mkdir -p gpurun_out
HAMK_RK4_WAVES=2 timeout 70 python -m pytest tests/test_gpu_wave.py -q -x -m gpu -k "vs_oracle and (chain33 or chain48 or chain64)" 2>&1 | tail -2
for s in chain48 chain64; do
  for w in 1 2; do
    HAMK_RK4_WAVES=$w timeout 40 python bench.py --system $s --batch 16384 --rk4-per-step 20 --steps 3 --warmup 1 --no-cpu-baseline --no-isa 2>/dev/null | tail -1 > gpurun_out/waves${w}_$s.json
    python -c "import json;d=json.load(open('gpurun_out/waves${w}_$s.json'));print('$s waves=$w', d['value'])"
  done
done

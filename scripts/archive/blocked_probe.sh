#!/bin/bash
# GPU: the panel factorisation of the wave kernels (HAMK_WAVE_BLOCKED=1) -- parity tests and throughput beside the flat one.
export HAMK_CACHE_DIR=$PWD/.hamk_cache
mkdir -p gpurun_out
HAMK_WAVE_BLOCKED=1 timeout 110 python -m pytest tests/test_gpu_wave.py -q -x -m gpu -k "chain32 or chain33 or chain20 or chain40" 2>&1 | tail -3
for v in 0 1; do
  HAMK_WAVE_BLOCKED=$v timeout 40 python bench.py --system chain32 --steps 3 --warmup 1 --no-cpu-baseline --no-isa 2>/dev/null | tail -1 > gpurun_out/blk${v}_chain32.json
  python -c "import json;d=json.load(open('gpurun_out/blk${v}_chain32.json'));print('chain32 blocked=$v', d['value'])"
done
for v in 0 1; do
  HAMK_WAVE_BLOCKED=$v timeout 40 python bench.py --system chain64 --batch 16384 --rk4-per-step 20 --steps 3 --warmup 1 --no-cpu-baseline --no-isa 2>/dev/null | tail -1 > gpurun_out/blk${v}_chain64.json
  python -c "import json;d=json.load(open('gpurun_out/blk${v}_chain64.json'));print('chain64 blocked=$v', d['value'])"
done

#!/bin/bash
# GPU box, round 4, pass M (the round's last GPU seconds): the tree after the A/B switches were removed -- smoke(), the four-lane
# kernels against the oracle on the GPU, one chain32 bench line.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04m_smoke.log
timeout 60 python -m pytest tests/test_gpu_wave.py -m gpu -q -k "test_quad_path_vs_oracle" 2>&1 | tail -2 | tee gpurun_out/r04m_quad_tests.log
timeout 40 python bench.py --system chain32 --steps 10 --warmup 3 --no-cpu-baseline --no-isa 2>/dev/null | tail -1 | tee gpurun_out/r04m_chain32.json | head -c 200; echo

cd ${GRAFT_REPO_ROOT:-$(pwd)}
for sys in threeBodyPolar chain8 chain16; do
  for k in 1 8; do
    python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 --no-cpu-baseline --no-isa --calls-per-launch $k 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sys', 'K=$k', '%.3e'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
  done
done

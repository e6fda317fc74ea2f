#!/bin/bash
# Builds the reproducer twice -- with the options libhamk gives hiprtc, and with MachineLICM disabled --
# prints each build's spill counts for hamk_rkf45_k and runs both (needs an MI355X for the runs).
cd "$(dirname "$0")"
# the frozen round-1 device library travels compressed (it is a reproducer input, not a component)
[ -f hamk_device_r01.hpp ] || xz -dk hamk_device_r01.hpp.xz
OPTS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-honor-nans -fno-signed-zeros"
for v in default nolicm; do
  EXTRA=""; [ $v = nolicm ] && EXTRA="-mllvm -disable-machine-licm"
  /opt/rocm/bin/hipcc $OPTS $EXTRA -Rpass-analysis=kernel-resource-usage repro.hip -o repro_$v 2> build_$v.log || { tail -5 build_$v.log; exit 1; }
  echo "== $v: $(grep -A12 'Function Name: hamk_rkf45_k' build_$v.log | grep -E 'SGPRs Spill|VGPRs:|ScratchSize' | tr '\n' ' ')"
done
if [ "$1" != "build" ]; then for v in default nolicm; do echo "== run $v"; ./repro_$v; done; fi

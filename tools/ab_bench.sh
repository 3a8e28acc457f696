#!/bin/bash
# A/B of library variants on the GPU box: every gpurun_prof/ab_*.so is run through the parity smoke once and
# a short bench 3 times, interleaved; prints env-steps/s.  Usage: bash tools/ab_bench.sh [extra bench args]
[ -z "$NOSMOKE" ] && for L in gpurun_prof/ab_*.so; do echo "== $L"; MJLAB_AMD_LIB=$L timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "qacc|qvel|smoke ok|Error"; done
for rep in 1 2 3; do
  for L in gpurun_prof/ab_*.so; do
    MJLAB_AMD_LIB=$L timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$L rep $rep: %.0f env-steps/s  %.4f ms/step  dominant kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done

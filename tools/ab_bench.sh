#!/bin/bash
# A/B of library variants on the GPU box: every gpurun_prof/ab_*.so is run through the parity
# smoke and a short bench; prints env-steps/s and per-stage ms.
for L in gpurun_prof/ab_*.so; do
  echo "== $L"
  [ -z "$NOSMOKE" ] && MJLAB_AMD_LIB=$L timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  MJLAB_AMD_LIB=$L timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   value %.0f env-steps/s  ms/step %.3f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['stage_ms'].items()}))"
done

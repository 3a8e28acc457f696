#!/bin/bash
# Wave-priority classes from score quantiles: bench of two scenes for several quantile triples (one library, env-driven),
# 2 interleaved repetitions.  MJLAB_NO_PRIORITY_REFRESH = the built-in row-count thresholds.
run() { timeout 300 python bench.py --steps 150 --warmup 40 --no-cpu-baseline --scene $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-18s %-22s %9.0f env-steps/s  %.4f ms' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for S in g1_velocity_flat go1_velocity_flat g1_velocity_rough; do
    MJLAB_NO_PRIORITY_REFRESH=1 run $S static
    for Q in 0.45,0.85,0.95 0.36,0.83,0.94 0.55,0.90,0.97 0.30,0.70,0.90; do MJLAB_PRIO_Q=$Q run $S $Q; done
  done
done

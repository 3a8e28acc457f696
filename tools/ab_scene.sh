#!/bin/bash
# Per-stage times of bench.py for the scenes given as arguments (default: flat and rough G1).
for S in ${@:-g1_velocity_flat g1_velocity_rough}; do
  python bench.py --scene $S --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$S: value %.0f env-steps/s  ms/step %.3f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['stage_ms'].items()}))"
done

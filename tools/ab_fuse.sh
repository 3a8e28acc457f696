#!/bin/bash
# A/B of the launch structures inside one gpurun call: per-stage kernels vs fused pre-solve vs one kernel per substep.
# Each variant 3 times, interleaved, so that box-to-box and run-to-run variation cancels.  Usage: bash tools/ab_fuse.sh <tag> [extra bench args]
TAG=${1:-ab_fuse}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2 3; do
  for F in "stage" "step" "step --substeps-per-call 4"; do
    timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --fuse $F "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$F rep $rep: value %.0f env-steps/s  ms/step %.4f' % (d['value'], d['ms_per_step']))" | tee -a $OUT/ab.txt
  done
done

#!/bin/bash
# A/B of the launch structures inside one gpurun call, each variant 3 times, interleaved, so that run-to-run
# variation shows.  Usage: bash tools/ab_fuse.sh <tag> [extra bench args]
TAG=${1:-ab_fuse}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2 3; do
  for F in "--fuse stage --substeps-per-call 1" "--fuse step --substeps-per-call 1 --no-control-kernel" "--fuse step --substeps-per-call 4 --no-control-kernel" "--fuse step"; do
    timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline $F "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$F rep $rep: value %.0f env-steps/s  ms/step %.4f' % (d['value'], d['ms_per_step']))" | tee -a $OUT/ab.txt
  done
done

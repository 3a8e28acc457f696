#!/bin/bash
# Bench line of every scene / variant quoted in DESIGN.md section 5 (one box, one build).  Usage: bash tools/bench_scenes.sh <tag>
OUT=gpurun_out/${1:-scenes}; mkdir -p $OUT
run() { local name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/$name.json
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print('%-28s %9.0f env-steps/s  %.4f ms' % (sys.argv[2], d['value'], d['ms_per_step']))" $OUT/$name.json $name; }
run g1_velocity_flat
for S in g1_tracking_flat go1_velocity_flat g1_velocity_rough go1_velocity_rough; do run $S --scene $S; done
run g1_flat_masked_forward --masked-forward
run g1_flat_8192 --envs-per-gpu 8192
run g1_flat_16384 --envs-per-gpu 16384
run g1_flat_readback --readback

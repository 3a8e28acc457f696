#!/bin/bash
# round 3 profile of the build the docs describe: tests, smoke, bench (+ full env), rocprofv3 stats, PMC traffic, SQ passes, phases, tail,
# the exchange over RCCL with one rank, half-size launch (what a half batch of the ping-pong exchange costs)
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
SQ=1 bash tools/gpu_round.sh r03_v7
T=gpurun_out/r03_v7
for E in 2048 8192; do
  timeout 200 python bench.py --envs-per-gpu $E --steps 100 --warmup 20 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_envs$E.json
  python -c "
import json; d=json.load(open('$T/bench_envs$E.json')); print('$E envs: %.0f env-steps/s %.4f ms' % (d['value'], d['ms_per_step']))" | tee -a $T/scenes.txt
done
for SC in g1_tracking_flat go1_velocity_flat g1_velocity_rough go1_velocity_rough; do
  timeout 300 python bench.py --scene $SC --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_$SC.json
  python -c "
import json; d=json.load(open('$T/bench_$SC.json')); print('$SC: %.0f env-steps/s %.4f ms kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $T/scenes.txt
done
MJLAB_LS_PARALLEL=0 timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_exact_ls.json
python -c "
import json; d=json.load(open('$T/bench_exact_ls.json')); print('g1_velocity_flat, exact line search (ls_parallel off): %.0f env-steps/s %.4f ms' % (d['value'], d['ms_per_step']))" | tee -a $T/scenes.txt
cp gpurun_out/exchange_one_rank_rccl.json $T/ 2>/dev/null
MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 300 python tools/tail_profile.py > $T/tail_profile.log 2>&1; tail -4 $T/tail_profile.log
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null

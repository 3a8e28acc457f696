#!/bin/bash
# One short GPU-box pass that verifies a build: sanity gate, GPU test suite, smoke(), bench.py (+ the scenes table with SCENES=1).
# Usage (dev container):  tools/stage_reference.sh 600 'bash tools/gpu_verify.sh <tag>'     (or plain gpurun, without the reference's env tests)
# A box whose GPU faults at the first launch must not eat the budget in timeouts (two calls of round 3 landed on one): the gate.
TAG=${1:-verify}; T=gpurun_out/$TAG; mkdir -p $T
[ -d gpurun_ref/src ] && export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || { echo "GPU SANITY FAILED" | tee $T/status.txt; exit 9; }
rm -f gpurun_out/parity_gate.txt
timeout 600 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -3 $T/gputests.log
grep -n "^E  .*Error\|^___" $T/gputests.log | cut -c1-800 | head -12
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt gpurun_out/exchange_one_rank_rccl.json $T/ 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $T/status.txt; tail -1 $T/smoke.log
timeout 400 python bench.py > $T/bench.log 2>&1; echo "bench rc=$?" | tee -a $T/status.txt; tail -1 $T/bench.log > $T/bench.json
python -c "
import json; d=json.load(open('$T/bench.json')); print({k: d[k] for k in ('value','value_with_gather','value_full_env','ms_per_step','std_over_5')})"
line() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print('%s: %.0f env-steps/s %.4f ms kernel %.4f' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$1" "$2" | tee -a $T/scenes.txt; }
if [ -n "$SCENES" ]; then
  for SC in g1_tracking_flat go1_velocity_flat g1_velocity_rough go1_velocity_rough; do
    timeout 200 python bench.py --scene $SC --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_$SC.json; line $T/bench_$SC.json $SC
  done
  timeout 200 python bench.py --envs-per-gpu 2048 --steps 100 --warmup 20 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_envs2048.json; line $T/bench_envs2048.json "g1_velocity_flat, 2048 envs"
  timeout 200 python bench.py --exact-ls --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_exact_ls.json; line $T/bench_exact_ls.json "g1_velocity_flat, exact line search (ls_parallel off)"
fi
if ls gpurun_prof/ab_*.so > /dev/null 2>&1; then NOSMOKE=1 bash tools/ab_bench.sh --no-full-env --no-latency-bound 2>&1 | tee $T/ab.txt; fi

#!/bin/bash
# round 3, call 3: A/B of the lane-parallel grid line search, GPU tests, full-env bench leg
T=gpurun_out/r03_v2; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
NOSMOKE=1 bash tools/ab_bench.sh --no-full-env 2>&1 | tee $T/ab_lsp.txt
MJLAB_LS_PARALLEL=0 MJLAB_AMD_LIB=gpurun_prof/ab_1u4.so python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_exact.json
python -c "
import json; d=json.load(open('$T/bench_exact.json')); print('exact search: %.0f env-steps/s %.4f ms kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $T/ab_lsp.txt
timeout 900 python -m pytest tests -m gpu -x -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -5 $T/gputests.log
timeout 600 python bench.py --steps 100 --warmup 20 > $T/bench.log 2>&1; echo "bench rc=$?" | tee -a $T/status.txt
tail -1 $T/bench.log > $T/bench.json
python -c "
import json; d=json.load(open('$T/bench.json')); print({k: d[k] for k in ('value','value_with_gather','value_full_env','value_full_env_note','ms_per_step')}); print(d['roofline'])"

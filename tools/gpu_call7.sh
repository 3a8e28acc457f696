#!/bin/bash
T=gpurun_out/r03_v6; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
rm -f gpurun_out/parity_gate.txt
timeout 1500 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -4 $T/gputests.log
grep -n "^E  .*Error\|^___" $T/gputests.log | cut -c1-1200 | head -30
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null
for SC in g1_tracking_flat g1_velocity_flat; do
  timeout 300 python bench.py --scene $SC --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>$T/bench_$SC.err | tail -1 > $T/bench_$SC.json
  python -c "
import json; d=json.load(open('$T/bench_$SC.json')); print('$SC: %.0f env-steps/s %.4f ms kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || tail -5 $T/bench_$SC.err
done
timeout 300 python tools/export_rollout_states.py > $T/export.log 2>&1; echo "export rc=$?" | tee -a $T/status.txt; tail -3 $T/export.log

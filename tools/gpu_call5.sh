#!/bin/bash
T=gpurun_out/r03_v4; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
rm -f gpurun_out/parity_gate.txt
timeout 1500 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -8 $T/gputests.log
grep -n "^E  .*AssertionError\|^___" $T/gputests.log | cut -c1-1500 | head -40
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null
for rep in 1 2; do for LF in 1 0; do
  MJLAB_LOCAL_FRAME=$LF timeout 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 > $T/bench_lf${LF}_$rep.json
  python -c "
import json; d=json.load(open('$T/bench_lf${LF}_$rep.json')); print('LOCAL_FRAME=$LF rep $rep: %.0f env-steps/s %.4f ms kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $T/ab_local_frame.txt
done; done

#!/bin/bash
T=gpurun_out/r03_v10; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || { echo "GPU SANITY FAILED" | tee $T/status.txt; exit 9; }
rm -f gpurun_out/parity_gate.txt
timeout 600 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -3 $T/gputests.log
grep -n "^E  .*Error\|^___" $T/gputests.log | cut -c1-800 | head -12
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $T/status.txt; tail -1 $T/smoke.log
timeout 300 python bench.py --no-full-env > $T/bench.log 2>&1; echo "bench rc=$?" | tee -a $T/status.txt; tail -1 $T/bench.log > $T/bench.json
python -c "
import json; d=json.load(open('$T/bench.json')); print({k: d[k] for k in ('value','value_with_gather','ms_per_step','std_over_5')}); print(d['roofline']['kernel_ms'], d['roofline']['flops'], d['roofline']['traffic_source'])"

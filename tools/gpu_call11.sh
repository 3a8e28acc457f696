#!/bin/bash
T=gpurun_out/r03_v9; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || { echo "GPU SANITY FAILED" | tee $T/status.txt; exit 9; }
rm -f gpurun_out/parity_gate.txt
timeout 600 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -3 $T/gputests.log
grep -n "^E  .*Error\|^___" $T/gputests.log | cut -c1-800 | head -12
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $T/status.txt; grep -v amdgpu $T/smoke.log | tail -12
for rep in 1 2; do
  for L in gpurun_prof/ab_*.so; do
    MJLAB_AMD_LIB=$L timeout 120 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-full-env 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$L rep $rep: %.0f env-steps/s  %.4f ms/step  dominant kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $T/ab_hskip.txt
  done
done

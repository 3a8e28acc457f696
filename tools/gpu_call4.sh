#!/bin/bash
# round 3, call 4: local-frame kinematics: A/B, the new far-from-origin test, full GPU suite (gate included)
T=gpurun_out/r03_v3; mkdir -p $T
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
rm -f gpurun_out/parity_gate.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "local_frame or ls_parallel" > $T/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $T/status.txt
grep -E "median|passed|failed|Error" $T/new_tests.log | head -30
NOSMOKE=1 bash tools/ab_bench.sh --no-full-env 2>&1 | tee $T/ab_rel.txt
timeout 1200 python -m pytest tests -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $T/status.txt
tail -15 $T/gputests.log
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt $T/ 2>/dev/null
grep -E "^==|^   qacc  |qacc off|^   qacc_smooth" $T/parity_gate.txt

#!/bin/bash
# second pass of the fault bisect: the stand-alone scratch check (tools/scratch_repro.hip) under the runtime's scratch settings
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/fault2
mkdir -p $OUT
B=gpurun_aux/scratch_repro
( timeout 120 $B ) > $OUT/default.log 2>&1; echo "default rc=$?" | tee -a $OUT/summary.txt
( HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 120 $B ) > $OUT/noreclaim.log 2>&1; echo "noreclaim rc=$?" | tee -a $OUT/summary.txt
( timeout 120 $B 2000000 ) > $OUT/longspin.log 2>&1; echo "longspin rc=$?" | tee -a $OUT/summary.txt
( AMD_LOG_LEVEL=4 timeout 120 $B 2>&1 | grep -i -E "scratch|private_seg_size=[1-9]|mismatch|grid " | cut -c1-400 | head -300 ) > $OUT/amdlog.log 2>&1
tail -n 40 $OUT/default.log $OUT/noreclaim.log

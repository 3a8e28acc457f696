#!/bin/bash
# Instruction-cache counter passes of the bench's dominant kernel (GPU box): is a 364 KB kernel whose waves stream ~165 KB of straight-line
# code per pass limited by instruction fetch?   gpurun --timeout 600 -- 'bash tools/icache_counters.sh <tag>'   (MJLAB_AMD_LIB=... for a variant)
TAG=${1:-icache}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
BCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-env --no-latency-bound --no-big-batch"
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -B4 -A2 "SQ_IFETCH_LEVEL" | head -40) > $OUT/ifetch_metric.txt
i=0
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQC_ICACHE_BUSY_CYCLES SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READYB"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/pmc_IC$i -o pmc -- $BCMD > $R/$OUT/pmc_IC$i.log 2>&1); echo "pmc IC$i rc=$?" | tee -a $OUT/status.txt
done
for f in $OUT/pmc_IC*/pmc_counter_collection.csv; do [ -f $f ] && python tools/reduce_pmc.py $f; done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +8M -delete
ls $OUT $OUT/pmc_IC1 2>/dev/null | head -30

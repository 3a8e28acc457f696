#!/bin/bash
# Memory-latency counter passes of the bench's dominant kernel (GPU box): mean latency of vector / scalar / LDS / instruction-fetch requests
# (rocprofv3's accumulate() metrics, one per pass) and the scalar data cache's hit rate.   gpurun --timeout 900 -- 'bash tools/latency_counters.sh <tag>'
TAG=${1:-latency}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
BCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-env --no-latency-bound --no-big-batch"
i=0
for P in "VmemLatency" "SmemLatency" "LdsLatency" "InstrFetchLatency" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/pmc_L$i -o pmc -- $BCMD > $R/$OUT/pmc_L$i.log 2>&1); echo "pmc L$i ($P) rc=$?" | tee -a $OUT/status.txt
done
for f in $OUT/pmc_L*/pmc_counter_collection.csv; do [ -f $f ] && python tools/reduce_pmc.py $f && grep "k_control_step" $f | cut -d, -f5- ; done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +8M -delete

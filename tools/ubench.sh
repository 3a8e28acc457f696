#!/bin/bash
# Calibration run on the GPU box: gpurun_prof/ubench plain (timings, s_memtime cycles) and under the SAME rocprofv3
# counter passes tools/gpu_round.sh / tools/sq_counters.sh use for the product kernels.
# Usage: gpurun --timeout 900 -- 'bash tools/ubench.sh <tag>'   ->  gpurun_out/<tag>/{ubench.jsonl, pmc_*.csv}
TAG=${1:-ubench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 90 $R/gpurun_prof/ubench > $OUT/ubench.jsonl 2> $OUT/ubench.err; echo "ubench rc=$?" | tee -a $OUT/status.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 120 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o pmc -- $R/gpurun_prof/ubench quick > $R/$OUT/pmc_$C.log 2>&1); echo "pmc $C rc=$?" | tee -a $OUT/status.txt
done
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/pmc_SQ$i -o pmc -- $R/gpurun_prof/ubench quick > $R/$OUT/pmc_SQ$i.log 2>&1); echo "pmc SQ$i rc=$?" | tee -a $OUT/status.txt
done
find $OUT -name "*.db" -delete
for f in $OUT/pmc_*/pmc_counter_collection.csv; do python tools/reduce_pmc.py $f; done
python tools/ubench_report.py $TAG | tee $OUT/report.txt

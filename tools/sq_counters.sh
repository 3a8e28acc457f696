#!/bin/bash
# SQ counter passes for the stage kernels (GPU box). Usage: bash tools/sq_counters.sh <tag>
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
BCMD="python $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA"
P3="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/p$i -o pmc -- $BCMD > $R/$OUT/p$i.log 2>&1); echo "pass $i rc=$?"
done
find $OUT -name "*.db" -delete
python tools/sq_summary.py $TAG | tee $OUT/summary.txt

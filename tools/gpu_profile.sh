#!/bin/bash
# The counter passes of a build with hard per-pass limits (tools/gpu_round.sh's passes; condensed by tools/collect_profile.py <tag>):
# rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE, two SQ passes, phase and tail profiles of the profiling library
# (python -m mjlab_amd.native --out gpurun_prof/libmjlab_amd_prof.so -DMJLAB_PROFILE).  Usage: gpurun --timeout 480 -- 'bash tools/gpu_profile.sh <tag>'
TAG=${1:-profile}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || { echo "GPU SANITY FAILED" | tee -a $OUT/status.txt; exit 9; }
BCMD="python $R/bench.py --scene ${SCENE:-g1_velocity_flat} --steps 40 --warmup 10 --no-cpu-baseline --no-full-env --no-latency-bound --no-big-batch"  # (the quarter-size launches of roofline.latency would mix into the kernel's average)
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- $BCMD > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/status.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 120 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o pmc -- $BCMD > $R/$OUT/pmc_$C.log 2>&1); echo "pmc $C rc=$?" | tee -a $OUT/status.txt
done
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/pmc_SQ$i -o pmc -- $BCMD > $R/$OUT/pmc_SQ$i.log 2>&1); echo "pmc SQ$i rc=$?" | tee -a $OUT/status.txt
done
for f in $OUT/pmc_*/pmc_counter_collection.csv; do [ -f $f ] && python tools/reduce_pmc.py $f; done
if [ -f gpurun_prof/libmjlab_amd_prof.so ]; then
  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 90 python tools/profile_phases.py > $OUT/phases.log 2>&1; echo "phases rc=$?" | tee -a $OUT/status.txt
  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 90 python tools/tail_profile.py > $OUT/tail_profile.log 2>&1; echo "tail rc=$?" | tee -a $OUT/status.txt
fi
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +8M -delete

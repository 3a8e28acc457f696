#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprofv3 kernel stats, HBM PMC passes, phase profile.
# Usage (dev container): gpurun --timeout 1500 -- '[SCENE=<scene>] [NOTESTS=1] bash tools/gpu_round.sh <tag> [quick]'
TAG=${1:-run}
QUICK=${2:-}
SCENE=${SCENE:-g1_velocity_flat}   # SCENE=g1_velocity_rough profiles the box-terrain scene
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
if [ -z "$NOTESTS" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/status.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/status.txt
fi
timeout 600 python bench.py --scene $SCENE > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -1 $OUT/bench.log > $OUT/bench.json
if [ -z "$QUICK" ]; then
  BCMD="python $R/bench.py --scene $SCENE --steps 40 --warmup 10 --no-cpu-baseline --no-full-env --no-latency-bound"  # (the quarter-size launches of roofline.latency would mix into the kernel's average)  # (the full-env leg is ~25 k small torch launches: minutes under --pmc)
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- $BCMD > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/status.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 150 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o pmc -- $BCMD > $R/$OUT/pmc_$C.log 2>&1); echo "pmc $C rc=$?" | tee -a $OUT/status.txt
  done
  if [ -n "$SQ" ]; then
    P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
    P2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS"
    i=0
    for P in "$P1" "$P2"; do
      i=$((i+1))
      (cd /tmp && timeout 150 rocprofv3 --pmc $P --output-format csv -d $R/$OUT/pmc_SQ$i -o pmc -- $BCMD > $R/$OUT/pmc_SQ$i.log 2>&1); echo "pmc SQ$i rc=$?" | tee -a $OUT/status.txt
      python tools/reduce_pmc.py $OUT/pmc_SQ$i/pmc_counter_collection.csv
    done
  fi
  if [ -f gpurun_prof/libmjlab_amd_prof.so ]; then
    MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 300 python tools/profile_phases.py > $OUT/phases.log 2>&1; echo "phases rc=$?" | tee -a $OUT/status.txt
  fi
  find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +8M -delete
fi
tail -3 $OUT/gputests.log 2>/dev/null; tail -2 $OUT/smoke.log 2>/dev/null; cat $OUT/bench.json; cat $OUT/phases.log 2>/dev/null

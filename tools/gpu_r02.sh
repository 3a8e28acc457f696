#!/bin/bash
# Round-2 GPU pass: full GPU test suite (no -x, output kept), smoke, bench (task events on / off),
# rocprofv3 kernel stats + HBM + SQ passes, phase profile.  Usage: gpurun --timeout 1800 -- 'bash tools/gpu_r02.sh <tag>'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
if [ -z "$NOTESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -q -s -rA > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/status.txt
grep -E "passed|failed" $OUT/gputests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/status.txt
cp gpurun_out/parity_margins.txt gpurun_out/parity_gate.txt $OUT/ 2>/dev/null
fi
timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -1 $OUT/bench.log > $OUT/bench.json
timeout 600 python bench.py --no-task-events --no-cpu-baseline > $OUT/bench_noevents.log 2>&1; tail -1 $OUT/bench_noevents.log > $OUT/bench_noevents.json
# A/B of the launch structure on this box (same build): one kernel per stage, reference call pattern
timeout 600 python bench.py --fuse stage --substeps-per-call 1 --no-cpu-baseline > $OUT/bench_stage.log 2>&1; tail -1 $OUT/bench_stage.log > $OUT/bench_stage.json
timeout 600 python bench.py --fuse step --substeps-per-call 1 --no-control-kernel --no-cpu-baseline > $OUT/bench_step1.log 2>&1; tail -1 $OUT/bench_step1.log > $OUT/bench_step1.json
timeout 600 python bench.py --fuse step --substeps-per-call 4 --no-control-kernel --no-cpu-baseline > $OUT/bench_step4.log 2>&1; tail -1 $OUT/bench_step4.log > $OUT/bench_step4.json
BCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- $BCMD > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/status.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o pmc -- $BCMD > $R/$OUT/pmc_$C.log 2>&1); echo "pmc $C rc=$?" | tee -a $OUT/status.txt
done
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/$OUT/pmc_SQ -o pmc -- $BCMD > $R/$OUT/pmc_SQ.log 2>&1); echo "pmc SQ rc=$?" | tee -a $OUT/status.txt
python tools/reduce_pmc.py $OUT/pmc_SQ/pmc_counter_collection.csv
if [ -f gpurun_prof/libmjlab_amd_prof.so ]; then
  MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 300 python tools/tail_profile.py > $OUT/tail_profile.log 2>&1; MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 300 python tools/profile_phases.py > $OUT/phases.log 2>&1; FUSE=step MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 300 python tools/profile_phases.py > $OUT/phases_fused.log 2>&1; echo "phases rc=$?" | tee -a $OUT/status.txt
fi
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +8M -delete
cat $OUT/status.txt; for f in bench bench_noevents bench_stage bench_step1 bench_step4; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])" $OUT/$f.json; done

#!/bin/bash
# rocprofv3 kernel census of the captured environments (GPU box, reference staged): launches per control step by kernel.
#   tools/stage_reference.sh 900 'bash tools/graph_census.sh <tag>'   ->  gpurun_out/<tag>/kernel_stats_<task>.csv
TAG=${1:-graph_census}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
R=$(pwd); STEPS=200
for T in Mjlab-Velocity-Flat-Unitree-G1 Mjlab-Tracking-Flat-Unitree-G1; do
  rm -rf $OUT/prof_$T
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$T -o trace -- python $R/tools/graphed_env_profile.py 4096 $STEPS $T random > $R/$OUT/profile_$T.log 2>&1)
  f=$(find $OUT/prof_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/kernel_stats_$T.csv" "$T" $STEPS <<'PY'
import csv, sys
src, dst, task, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rows = list(csv.DictReader(open(src)))
nstep = steps + 13  # graphed_env_profile.py: 2 warm-up steps + the capture + 10 untimed + the timed steps
total = sum(int(r["Calls"]) for r in rows)
with open(dst, "w") as f:
  f.write(f"# {task}: GraphedRlEnv at 4096 envs, {nstep} control steps (2 warm-up + capture + 10 + {steps} timed, random policy) under rocprofv3 --kernel-trace --stats: {total} kernel launches = {total / nstep:.0f} per step incl. set-up\n")
  f.write("name,calls,calls_per_step,avg_us,pct\n")
  for r in rows[:60]:
    f.write(f"{r['Name'].replace(',', ';')[:90]},{r['Calls']},{int(r['Calls']) / nstep:.2f},{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}\n")
print(open(dst).readline().strip())
PY
  grep GRAPHED $OUT/profile_$T.log
done
find $OUT -name "*.db" -delete; find $OUT -name "*trace*.csv" -size +2M -delete

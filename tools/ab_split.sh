for rep in 1 2 3; do
  for F in "--fuse presolve --substeps-per-call 1 --no-control-kernel" "--fuse step --substeps-per-call 1 --no-control-kernel" "--fuse step"; do
    timeout 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-latency-bound $F 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$F rep $rep: value %.0f env-steps/s  ms/step %.4f' % (d['value'], d['ms_per_step']))"
  done
done

for rep in 1 2 3; do
  for F in "--balance-every 0" "--balance-every 8" "--balance-every 4" "--balance-every 16"; do
    timeout 300 python bench.py --steps 160 --warmup 32 --no-cpu-baseline $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$F rep $rep: value %.0f env-steps/s  ms/step %.4f  kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done

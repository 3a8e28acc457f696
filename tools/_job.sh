T=gpurun_out/r04_v13; mkdir -p $T
[ -d gpurun_ref/src ] && export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
rm -f gpurun_out/parity_gate.txt
timeout 900 python -m pytest tests/test_gpu_reference_env.py "tests/test_gpu_parity_gate.py::test_rollout_state_parity_with_the_grid_line_search" -m gpu -q -rs -s > $T/gputests.log 2>&1; echo "gputests rc=$?"
tail -4 $T/gputests.log; grep -n "graphed.*env vs\|^E  " $T/gputests.log | cut -c1-400 | head
timeout 300 python bench.py --scene go1_velocity_flat --steps 100 --no-cpu-baseline > $T/bench_go1.log 2>&1; tail -1 $T/bench_go1.log > $T/bench_go1.json
python -c "
import json; d=json.load(open('$T/bench_go1.json')); print({k: d[k] for k in ('value','value_full_env','value_full_env_graphed','value_full_env_graphed_note','ms_per_step')})"

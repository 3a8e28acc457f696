bash tools/gpu_verify.sh r04_v6
cat gpurun_out/env_golden_margins.txt
python -c "
import json; d=json.load(open('gpurun_out/r04_v6/exchange_one_rank_rccl.json')); p=d['pipelined']; print({k: p.get(k) for k in ('ms_per_step','ms_per_step_same_halves_sequential_exchange','ms_per_step_halves_compute_only','ms_per_step_halves_concurrent_compute_only','exchange_overlap_frac','error')}, d['ms_per_step'], d['exchange_ms_per_step'])"
MJLAB_DIST_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-full-env --no-latency-bound 2>/dev/null | tail -1 > gpurun_out/r04_v6/exchange_4096_one_rank_rccl.json
python -c "
import json; d=json.load(open('gpurun_out/r04_v6/exchange_4096_one_rank_rccl.json')); p=d['pipelined']; print('4096:', {k: p.get(k) for k in ('ms_per_step','ms_per_step_same_halves_sequential_exchange','ms_per_step_halves_compute_only','ms_per_step_halves_concurrent_compute_only','exchange_overlap_frac','error')}, d['ms_per_step'], d['exchange_ms_per_step'])"

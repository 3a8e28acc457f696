T=gpurun_out/r04_v19; mkdir -p $T
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
bash tools/ab_bench.sh --no-full-env --no-latency-bound 2>&1 | grep -v "smoke forward\|smoke step" | tee $T/ab_sched_strategy.txt

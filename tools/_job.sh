T=gpurun_out/r04_v18; mkdir -p $T
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
bash tools/ab_bench.sh --no-full-env --no-latency-bound --scene g1_tracking_flat 2>&1 | tee $T/ab_inoise_tracking.txt
NOSMOKE=1 bash tools/ab_bench.sh --no-full-env --no-latency-bound 2>&1 | tee $T/ab_inoise_velocity.txt

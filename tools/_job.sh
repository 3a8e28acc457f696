T=gpurun_out/r04_v5; mkdir -p $T
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
timeout 300 python -m pytest tests/test_env_golden.py -m gpu -q -rs > $T/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $T/gputests.log; grep -n "^E  " $T/gputests.log | cut -c1-300 | head -5
cat gpurun_out/env_golden_margins.txt
bash tools/ab_bench.sh --no-full-env --no-latency-bound 2>&1 | tee $T/ab_chol_rl.txt
for NW in 1024 4096; do
  NWORLD=$NW MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof_rl.so timeout 120 python tools/profile_phases.py > $T/phases_rl_$NW.txt 2>&1
done
grep -n "chol_factor\|chol_solve\|mean cycles per world-step in k_solve" $T/phases_rl_*.txt

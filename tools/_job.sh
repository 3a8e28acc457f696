T=gpurun_out/r04_v11; mkdir -p $T
[ -d gpurun_ref/src ] && export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
rm -f gpurun_out/parity_gate.txt
timeout 900 python -m pytest tests/test_gpu_reference_env.py tests/test_env_golden.py tests/test_golden.py tests/test_gpu_parity_gate.py -m gpu -q -rs -s > $T/gputests.log 2>&1; echo "gputests rc=$?"
tail -4 $T/gputests.log; grep -n "graphed.*env vs\|^E  " $T/gputests.log | cut -c1-400 | head
cat gpurun_out/env_golden_margins.txt | grep "post_\|rows"
cp gpurun_out/parity_gate.txt $T/ 2>/dev/null
grep -n "deeper than 5 mm" $T/parity_gate.txt | cut -c1-300

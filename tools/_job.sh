T=gpurun_out/r04_v14; mkdir -p $T
[ -d gpurun_ref/src ] && export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 90 python -c "import torch; x = torch.ones(1024, device='cuda'); print('gpu ok', float((x * 2).sum()))" || exit 9
timeout 900 python -m pytest tests/test_gpu_reference_env.py -m gpu -q -rs -s > $T/gputests.log 2>&1; echo "gputests rc=$?"
tail -4 $T/gputests.log; grep -n "graphed.*env vs\|^E  " $T/gputests.log | cut -c1-400 | head

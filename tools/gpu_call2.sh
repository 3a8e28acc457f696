bash tools/ubench.sh r03_ubench
mkdir -p gpurun_out/r03_v1
export MJLAB_REFERENCE_SRC=$PWD/gpurun_ref/src
timeout 900 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r03_v1/gputests.log 2>&1; echo "gputests rc=$?" | tee -a gpurun_out/r03_v1/status.txt
tail -5 gpurun_out/r03_v1/gputests.log
timeout 300 python -m pytest tests/test_gpu_reference_env.py tests/test_gpu_parity.py -m gpu -q -s -k "reference_g1 or registered or ls_parallel" > gpurun_out/r03_v1/n1_lsp.log 2>&1; echo "n1/lsp rc=$?" | tee -a gpurun_out/r03_v1/status.txt
grep -E "reference G1|ls_parallel,|passed|failed" gpurun_out/r03_v1/n1_lsp.log
for rep in 1 2; do for LS in 0 1; do
  MJLAB_LS_PARALLEL=$LS timeout 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_v1/bench_ls${LS}_$rep.json
  python -c "
import json; d=json.load(open('gpurun_out/r03_v1/bench_ls${LS}_$rep.json')); print('LS_PARALLEL=$LS rep $rep: %.0f env-steps/s %.4f ms kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
cp gpurun_out/parity_gate.txt gpurun_out/parity_margins.txt gpurun_out/r03_v1/ 2>/dev/null

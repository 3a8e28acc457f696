bash tools/gpu_verify.sh r04_v12
cat gpurun_out/env_golden_margins.txt | grep "post_"
grep -n "deeper than 5 mm" gpurun_out/r04_v12/parity_gate.txt | cut -c1-330
NWORLD=4096 MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 120 python tools/profile_phases.py 2>&1 | grep "switch zone"

bash tools/gpu_verify.sh r04_v3
T=gpurun_out/r04_v3
for NW in 1024 4096; do
  NWORLD=$NW MJLAB_AMD_LIB=gpurun_prof/libmjlab_amd_prof.so timeout 120 python tools/profile_phases.py > $T/phases_$NW.txt 2>&1
done
head -20 $T/phases_1024.txt

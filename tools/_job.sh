bash tools/gpu_verify.sh r04_v15
bash tools/gpu_profile.sh r04_v15p

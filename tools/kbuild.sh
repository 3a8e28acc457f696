#!/bin/bash
# Build a library variant and print the register / scratch footprint of the G1 solve kernel.
# Usage: tools/kbuild.sh <out.so> [-DFOO ...]
OUT=$(realpath -m $1); shift
T=/tmp/kb_$(basename $OUT .so); mkdir -p $T
( cd $T && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 -shared -fPIC "$@" --save-temps=obj /root/repo/mjlab_amd/csrc/mjlab_amd.hip -o $T/lib.so 2>&1 | grep -E "error" )
cp $T/lib.so $OUT
S=$(ls $T/*gfx950*.s | head -1)
for K in _Z17k_solve_integrateILi36 _Z10k_position _Z12k_constraint _Z10k_velocity _Z11k_collision; do
  echo "$K: $(grep -A14 "\.name: *$K" $S | grep -E "vgpr_count|vgpr_spill|sgpr_spill|private_seg" | tr -s ' ' | tr '\n' ' ')"
done

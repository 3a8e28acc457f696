#!/bin/bash
# Build a library variant and print the register / scratch footprint of the G1-sized (NVP = 36) kernels.
# Usage: tools/kbuild.sh <out.so> [-DFOO ...]
OUT=$(realpath -m $1); shift
python -m mjlab_amd.native --out $OUT "$@" > /dev/null || exit 1
T=/tmp/kb_$(basename $OUT .so); rm -rf $T; mkdir -p $T
for PART in 0 1; do
  mkdir -p $T/p$PART
  ( cd $T/p$PART && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 -fPIC "$@" -DMJLAB_NVP=36 -DMJLAB_NVP_PART=$PART --save-temps=obj \
      -c /root/repo/mjlab_amd/csrc/nvp_inst.hip -o $T/p$PART/nvp36.o 2>&1 | grep -E "error" ) &
done
mkdir -p $T/abi
( cd $T/abi && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 -fPIC "$@" --save-temps=obj -c /root/repo/mjlab_amd/csrc/mjlab_amd.hip -o $T/abi/abi.o 2>&1 | grep -E "error" ) &
wait
for S in $T/*/*gfx950*.s; do
  grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $S | paste - - - - - | sed 's/  */ /g' | grep -E "k_solve_integrate|k_substep|k_control_step|k_position|k_collision|k_velocity|k_constraint|k_presolve"
done

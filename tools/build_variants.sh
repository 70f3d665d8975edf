#!/bin/bash
# dev: build variant libraries of the SAME sources with extra -D flags into tmp_libs/ (git-ignored, travels with gpurun)
# usage: tools/build_variants.sh name1 "-DFLAG=1" name2 "-DFLAG=2 -DCGIC_PHASE_CLOCKS" ...
set -e
cd "$(dirname "$0")/../control-gic_amd/csrc"
mkdir -p ../../tmp_libs
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-parameter"
SRC="cgic_table.hip cgic_vq.hip cgic_vq_bwd.hip cgic_entropy.hip cgic_router.hip cgic_coder.hip cgic_decode.hip cgic_decode_ss.hip cgic_merge.hip cgic_launch.hip"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $extra -fgpu-rdc -shared -o ../../tmp_libs/lib_$name.so $SRC && echo built $name ) &
done
wait

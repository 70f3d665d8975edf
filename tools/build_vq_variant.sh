#!/bin/bash
# dev: variant libraries that differ only in cgic_vq.hip's -D flags: that unit is recompiled, the others are linked from the objects
# `make` left in csrc/ (run make first).  usage: tools/build_vq_variant.sh name1 "-DFLAG=1" name2 "" ...   -> tmp_libs/lib_<name>.so
set -e
cd "$(dirname "$0")/../control-gic_amd/csrc"
mkdir -p ../../tmp_libs
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-parameter"
OTHERS="cgic_table.o cgic_vq_bwd.o cgic_entropy.o cgic_router.o cgic_coder.o cgic_decode.o cgic_decode_ss.o cgic_merge.o cgic_launch.o"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c cgic_vq.hip -o /tmp/cgic_vq_$name.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tmp_libs/lib_$name.so /tmp/cgic_vq_$name.o $OTHERS && echo built $name ) &
done
wait

#!/bin/bash
# dev: register / spill counts of the filter-path kernels of the current cgic_vq.hip (no rdc: codegen happens at compile time)
cd "$(dirname "$0")/../control-gic_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-parameter "$@" -c -o /tmp/vq_regs.o cgic_vq.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|VGPR Spill|ScratchSize" | grep -A3 "vq_filter_router_kernelILb1ELb0\|vq_filter_kernelILb1ELb0" | sed 's/.*remark: //' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'

#!/bin/bash
# Registers, spills, LDS and occupancy of the kernels of one translation unit (compiler remarks; no GPU needed):
#   tools/kernel_resources.sh block|table|main [name filter]
cd "$(dirname "$0")/../simd-gaussian-ray-tracing_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize --cuda-device-only -Rpass-analysis=kernel-resource-usage"
case "$1" in
  block) SRC=vrt_block_kernel.hip; EXTRA="-DVRT_RENDER_ECMAX=6 -DVRT_RENDER_WPE=3 -mllvm -amdgpu-sched-strategy=max-ilp" ;;
  table) SRC=vrt_table_kernel.hip; EXTRA="" ;;
  *) SRC=vrt_kernels.hip; EXTRA="" ;;
esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o /dev/null 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/^.*remark: //' | paste - - - - - - - - - - | grep -E "${2:-.}" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-400

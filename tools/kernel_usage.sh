#!/bin/bash
# Register / spill / scratch figures of the kernels of one source file under extra -D flags (device-only compile):
#   tools/kernel_usage.sh pbs_fft_wave.hip [-DX=1 ...] | grep -A12 "ILi1ELi23ELi0ELb0ELi0"
cd "$(dirname "$0")/.."
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage "$@" tfhe_rs_amd/csrc/$src -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|SGPRs:|Occupancy|LDS Size" | sed -e 's/.*remark: [^ ]* //'

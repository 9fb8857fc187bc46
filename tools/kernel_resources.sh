#!/usr/bin/env bash
# registers / spills / LDS of the observation kernels as the compiler reports them (device-only compile with the build's flags):
#   tools/kernel_resources.sh [unit ...]      (default: fl_obs_m2 fl_obs_m0 fl_obs_m1)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
units=("$@"); [ ${#units[@]} -eq 0 ] && units=(fl_obs_m2 fl_obs_m0 fl_obs_m1)
for u in "${units[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -disable-lsr \
    -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result ${EXTRA_HIPCC_FLAGS:-} -Rpass-analysis=kernel-resource-usage \
    -c "$ROOT/flatland_marl_amd/csrc/$u.hip" -o /dev/null 2>&1 |
    grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//' | paste - - - - - - - - - | sed 's/  */ /g' &
done
wait

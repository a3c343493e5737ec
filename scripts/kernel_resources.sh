#!/bin/bash
# VGPR / AGPR / scratch / occupancy of every kernel in libmemvul_hip.so as hipcc reports them (no GPU needed).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/kr && cd /tmp/kr
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$ROOT/memvul_amd/csrc/engine.hip" -o /tmp/kr/x.so 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - - |
  grep -E "${1:-.}"

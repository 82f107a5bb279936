#!/bin/bash
# Build first (abort on failure), then run the given command on the GPU box.
set -e
make -C /root/repo/smith-waterman_amd all 2>&1 | grep -E "error|Error" && { echo "BUILD FAILED"; exit 1; }
make -C /root/repo/smith-waterman_amd all >/dev/null
T=${GPU_TIMEOUT:-900}
/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1 | grep -v "amdgpu.ids"

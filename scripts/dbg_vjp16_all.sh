#!/bin/bash
run() { echo "=== $FLOWRL_HIP_LIB $*"; timeout -k 10 120 python -u scripts/dbg_vjp16.py "$@" 2>&1 | grep -v "amdgpu.ids" | grep -E "===|dL/dxT|nan [1-9]|Error|error|bad|sentinel|   wt|lane" || exit 1; }
run cfg5_noclip_b512 B=256
run cfg5_noclip_b512 B=2048

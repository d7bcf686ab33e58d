#!/bin/bash
# development aid: build libflowrl_hip with one translation unit recompiled under extra defines
# usage: scripts/build_variant.sh <tag> <source.hip> <defines...>   ->  flow-rl_amd/flowrl_hip/exp_<tag>.so
set -e
cd "$(dirname "$0")/../flow-rl_amd/csrc"
tag=$1; src=$2; shift 2
stem=$(basename "$src" .hip)
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-value -I../../include -I. -fno-slp-vectorize "$@" -c "$src" -o "_obj/exp_${tag}.o"
objs=$(ls _obj/*.o | grep -v "/exp_" | grep -v "/${stem}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs "_obj/exp_${tag}.o" -o "../flowrl_hip/exp_${tag}.so"
echo "built exp_${tag}.so"

#!/bin/bash
# A/B of the f32 sampler's activation code generation: interleaved rounds on ONE device, kernel only.
#   base        libflowrl_hip.so (packed-f32 act_fn2, SLP vectorizer on)
#   noslp       -fno-slp-vectorize
#   scalar      two scalar activations per register pair + -fno-slp-vectorize
#   scalar_slp  two scalar activations, SLP vectorizer on
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab_mish
for round in 1 2 3; do
  for v in base noslp scalar scalar_slp; do
    lib=flow-rl_amd/flowrl_hip/exp_rs_$v.so; [ $v = base ] && lib=flow-rl_amd/flowrl_hip/libflowrl_hip.so
    FLOWRL_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --kernel-only --steps 200 2>/dev/null | sed "s/^/$v r$round /" | tee -a gpurun_out/ab_mish/ab.log || exit 2
  done
done

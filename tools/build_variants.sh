#!/bin/bash
# Variant builds of libultra_hip.so for A/B measurements on the GPU box (they travel with gpurun; build/ is git-ignored):
#   build/stamps.so                 -DUH_MIXFFT_STAMPS (tools/mix_fft_stalls.py)
#   build/v_<name>.so               one per "name=flags" argument, e.g.  bash tools/build_variants.sh occ4="-DUH_MIX2_WAVES=4"
cd "$(dirname "$0")/../projectultra_amd/csrc" || exit 1
mkdir -p ../../build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -ldl"
/opt/rocm/bin/hipcc $FL -DUH_MIXFFT_STAMPS $STAMP_FLAGS -o ../../build/stamps.so ultra_hip.hip 2>&1 | grep -E "error"
for a in "$@"; do
  name=${a%%=*}; flags=${a#*=}
  /opt/rocm/bin/hipcc $FL $flags -o ../../build/v_$name.so ultra_hip.hip 2>&1 | grep -E "error"
  echo "built build/v_$name.so ($flags)"
done

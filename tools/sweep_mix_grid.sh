#!/bin/bash
# Sweep of the transform's grid size (workgroups per CU) on the GPU box: needs build/v_ab.so (bash tools/build_variants.sh ab="-DUH_AB_SWITCHES").
for n in 1048576 131072; do
for g in 0 24 48 64 96 128 192 256; do
  if [ $g = 0 ]; then envs=""; else envs="ULTRA_HIP_MIX_WG_PER_CU=$g"; fi
  env ULTRA_HIP_LIB=build/v_ab.so $envs python3 bench.py --config cfg3 --frames $n --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('frames $n grid $g: step %.3f ms  mix_fft %.4f x%g  (total %.3f)' % (d['ms_per_step'], k['mix_fft_kernel']['avg_launch_ms'], k['mix_fft_kernel']['launches_per_step'], k['mix_fft_kernel']['ms_per_step']))"
done; done

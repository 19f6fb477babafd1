#!/bin/bash
# Development helper: time tools/acquire_bench.py (or "$@") against every build/v_*.so (variants of libultra_hip.so
# built from modified sources), restoring the in-tree library afterwards.  Run on the GPU box.
cp projectultra_amd/libultra_hip.so /tmp/libultra_hip.keep
for v in build/v_*.so; do
  cp $v projectultra_amd/libultra_hip.so
  echo "== $v"
  ${@:-python3 tools/acquire_bench.py 4096} 2>&1 | grep -v "amdgpu.ids"
done
cp /tmp/libultra_hip.keep projectultra_amd/libultra_hip.so

#!/bin/bash
# ISA of one kernel: bash tools/kernel_isa.sh <mangled-name regex> [out.s]   (device-only -S of ultra_hip.hip, cached in /tmp/ultra.s)
cd "$(dirname "$0")/../projectultra_amd/csrc" || exit 1
if [ ! -f /tmp/ultra.s ] || [ -n "$(find . ../../include -newer /tmp/ultra.s -name '*.h*' 2>/dev/null)" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
     --cuda-device-only -S -o /tmp/ultra.s ultra_hip.hip 2>/dev/null
fi
python3 - "$1" "${2:-/tmp/kernel.s}" <<'PY'
import re,sys
pat=re.compile(sys.argv[1]); out=[]; on=False
for l in open('/tmp/ultra.s'):
    m=re.match(r'^(_Z\S+|\w+):\s*;? ?@',l)
    if m: on=bool(pat.search(m.group(1)))
    if on:
        out.append(l)
        if 's_endpgm' in l: on=False
open(sys.argv[2],'w').writelines(out); print(len(out),'lines ->',sys.argv[2])
PY

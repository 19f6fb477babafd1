#!/bin/bash
# Register / LDS / occupancy figures of the kernels whose (demangled) name matches $1 (default: all), from hipcc's
# -Rpass-analysis=kernel-resource-usage.  usage: bash tools/kernel_resources.sh [regex] [extra hipcc flags]
cd "$(dirname "$0")/../projectultra_amd/csrc" || exit 1
PAT=${1:-.}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage "$@" -o /dev/null ultra_hip.hip 2>&1 |
python3 -c "
import re,sys,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark: +Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark: +(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',l)
    if m and cur: rows[cur][m.group(1).split()[0]]=int(m.group(2))
names=list(rows)
dem=subprocess.run(['c++filt']+names,capture_output=True,text=True).stdout.split('\n')
pat=re.compile(sys.argv[1])
for n,d in zip(names,dem):
    short=re.sub(r'\(.*','',d).replace('ultra_hip::dev::','').replace('void ','')
    if pat.search(short):
        r=rows[n]; print(f\"{short:58s} VGPR {r.get('VGPRs',0):3d} AGPR {r.get('AGPRs',0):3d} SGPR {r.get('SGPRs',0):3d} scratch {r.get('ScratchSize',0):4d} occ {r.get('Occupancy',0)} LDS {r.get('LDS',0)}\")
" "$PAT"

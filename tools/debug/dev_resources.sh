#!/bin/bash
# Quick look at a kernel's registers and spills without the full build: device-only compile of the reduced dispatch
# (-DLCF_DEV_BUILD), resource remarks of the kernels matching $1 (a regex on the mangled name).  Extra flags: $2...
cd "$(dirname "$0")/../../lightcurve_fitting_amd/csrc"
PAT=${1:-k_solo_run}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -ffp-contract=on \
  -mllvm -disable-machine-licm -DLCF_DEV_BUILD "$@" -Rpass-analysis=kernel-resource-usage -S --cuda-device-only -o /tmp/lcf_dev.s lcf_hip.hip 2> /tmp/lcf_dev.res
python3 - "$PAT" <<'PY'
import re, sys
text = open('/tmp/lcf_dev.res').read()
if 'error' in text: print(text[:3000])
for m in re.finditer(r'Function Name: (\S+)(.*?)(?=Function Name:|\Z)', text, re.S):
    f = dict(re.findall(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)', m.group(2)))
    if re.search(sys.argv[1], m.group(1)):
        print(m.group(1)[14:62], {k: f.get(k) for k in ('VGPRs', 'SGPRs Spill', 'VGPRs Spill', 'ScratchSize', 'Occupancy')})
PY

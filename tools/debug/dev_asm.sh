#!/bin/bash
# Quick look at the code of the ND = 5 instantiations only: device-only assembly + resource remarks of a copy of
# lcf_hip.hip whose dimension switches keep `case 5` alone (20 s instead of 2 min).
#   bash tools/debug/dev_asm.sh [extra hipcc flags]   ->  build_variants/dev.s, build_variants/dev.resources.txt
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_variants
sed -E '/^\s*case [2346789]: LCF_[A-Z_0-9]+\([0-9]\); break;/d; s/^(\s*)default: (LCF_[A-Z_0-9]+)\(0\); break;/\1default: break;/' \
  $R/lightcurve_fitting_amd/csrc/lcf_hip.hip > $R/build_variants/dev.hip
cp $R/lightcurve_fitting_amd/csrc/lcf_device.h $R/build_variants/
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$R/include -I$R/lightcurve_fitting_amd/csrc -Wno-unused-value -ffp-contract=on -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o $R/build_variants/dev.s $R/build_variants/dev.hip 2> $R/build_variants/dev.resources.txt
grep -A12 "Function Name: .*k_runILi5ELi1ELb1ELi2E" $R/build_variants/dev.resources.txt | grep -E "VGPRs:|Spill|ScratchSize" | tr '\n' ' '; echo

#!/bin/bash
# build_variants/liblcf_<name>.so with extra -D flags:  bash tools/debug/build_variant.sh name -DLCF_X=1 ...
NAME=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wno-unused-value -ffp-contract=on -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage -shared -o $R/build_variants/liblcf_$NAME.so \
  $R/lightcurve_fitting_amd/csrc/lcf_hip.hip $R/lightcurve_fitting_amd/csrc/lcf_sed.hip 2> $R/build_variants/$NAME.resources.txt
grep -A12 "k_soloILi5ELi1ELb1ELi2E" $R/build_variants/$NAME.resources.txt | grep -E "VGPRs:|Spill|Scratch" | tr '\n' ' '; echo " <- $NAME"

#!/bin/bash
# A/B of k_solo_run build variants (build_variants/liblcf_<name>.so) x board memory type, 1000 steps of configs[1]
for rep in 1 2; do
for v in "$@"; do
  for mem in cached uncached; do
    if [ $mem = uncached ]; then export LCF_RUN_BOARD_UNCACHED=1; else unset LCF_RUN_BOARD_UNCACHED; fi
    echo "== $v $mem: $(KERNELS=auto LCF_PEER_WAIT_S=1 LCF_HIP_LIB=build_variants/liblcf_$v.so timeout -k 10 120 python tools/debug/run_kernel_check.py 1024 1000 2>&1 | grep '1000 steps')"
  done
done
done

for v in A B C D E F G; do for i in 1 2; do LCF_HIP_LIB=build_variants/liblcf_$v.so python bench.py --steps 2000 --variant 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_ab_${v}_$i.json; done; done

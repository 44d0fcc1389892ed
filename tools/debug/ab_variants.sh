# A/B of build_variants/liblcf_<v>.so on one box: configs[1] twice and companion once per variant
for v in "$@"; do
  for i in 1 2; do LCF_HIP_LIB=$PWD/build_variants/liblcf_$v.so python bench.py --steps 2000 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_${v}_mcmc$i.json; done
  LCF_HIP_LIB=$PWD/build_variants/liblcf_$v.so python bench.py --workload companion --no-cpu-baseline 2>/dev/null > gpurun_out/ab_${v}_companion.json
  LCF_HIP_LIB=$PWD/build_variants/liblcf_$v.so python bench.py --workload population --no-cpu-baseline 2>/dev/null > gpurun_out/ab_${v}_population.json
done
python tools/debug/show_bench.py gpurun_out/ab_*.json

# A/B of library variants: configs[1] per half-step over a 1000-step run, the bench lines of mcmc (1000 and 20 steps) and population
export LCF_BENCH_NO_E2E=1
bash tools/debug/ab_kernel_time.sh "$@"
for v in "$@"; do
  export LCF_HIP_LIB=$PWD/build_variants/liblcf_$v.so
  for cfg in "mcmc 1000" "mcmc 20" "population 300"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --workload $1 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']; print('$v', '$cfg', '%.4e' % d['value'], 'hs/launch', r.get('half_steps_per_launch'), 'us/hs', 1e3*r.get('kernel_ms_per_half_step',0), flush=True)
"
  done
done

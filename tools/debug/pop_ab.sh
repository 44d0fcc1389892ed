# A/B of library variants on the population line (configs[4])
export LCF_BENCH_NO_E2E=1
for v in "$@"; do
  lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="A=1"
  env $envs LCF_HIP_LIB=$PWD/build_variants/liblcf_$lib.so timeout -k 10 300 python bench.py --workload population --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']; print('$v', '%.4e walker-steps/s' % d['value'], 'us per half-step %.2f' % (1e3*r['kernel_ms_per_half_step']), 'half-steps per launch %.0f' % r['half_steps_per_launch'], flush=True)
"
done

# A/B of library variants (or of environment knobs: "variant:ENV=1") on the driver's 20-step line and on 1000 steps
export LCF_BENCH_NO_E2E=1
for v in "$@"; do
  lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="A=1"
  for steps in 20 20 1000; do
    env $envs LCF_HIP_LIB=$PWD/build_variants/liblcf_$lib.so timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$v', $steps, 'steps: %.4e walker-steps/s' % d['value'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'device %.2f' % (1e3*d['device_ms_per_step']), [round(x*1e6/$steps,2) for x in d['timed_region']['seconds']] if 'seconds' in d.get('timed_region',{}) else '', flush=True)
"
  done
done

R=$GRAFT_REPO_ROOT
for parts in 1 2; do
  for hot in 0 1; do
    export LCF_PARTS=$parts LCF_EXPERIMENT_HOT=$hot
    python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bv.log 2>&1
    python3 -c "
import json;d=json.loads(open('$R/gpurun_out/bv.log').read().strip().splitlines()[-1]);print('parts $parts hot $hot', round(d['value']/1e6,2), round(d['device_ms_per_step']*1e3,1), round(d['roofline']['kernel_ms']*1e3,1))"
  done
done

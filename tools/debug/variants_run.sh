R=$GRAFT_REPO_ROOT
for lib in $R/build_variants/*.so; do
  for tag in fused; do
    if [ $tag = nofused ]; then export LCF_NO_FUSED=1; else unset LCF_NO_FUSED; fi
    LCF_HIP_LIB=$lib python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bv.log 2>&1
    python3 -c "
import json;d=json.loads(open('$R/gpurun_out/bv.log').read().strip().splitlines()[-1]);print('$(basename $lib)', '$tag', round(d['value']/1e6,2), round(d['device_ms_per_step']*1e3,1), round(d['roofline']['kernel_ms']*1e3,1))"
  done
done

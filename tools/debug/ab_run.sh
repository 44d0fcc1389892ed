set -e
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $R/gpurun_out/t.log 2>&1 || { tail -30 $R/gpurun_out/t.log; exit 1; }
tail -2 $R/gpurun_out/t.log
for tag in fused nofused; do
  if [ $tag = nofused ]; then export LCF_NO_FUSED=1; fi
  python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/b_$tag.log 2>&1
  python3 -c "
import json;d=json.loads(open('$R/gpurun_out/b_$tag.log').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['device_ms_per_step'])"
done

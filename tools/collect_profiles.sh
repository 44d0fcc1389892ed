#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun).
#   bash tools/collect_profiles.sh r01
# kernel-trace/stats and every --pmc pass are separate runs, as the profiling guide prescribes.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for V in 2 1; do
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/v${V}_pmc$i -- python3 $R/tools/prof_kernel.py 3 $V > $OUT/v${V}_pmc$i.log 2>&1
done
python3 - "$OUT" "$V" <<'PY'
import csv, glob, collections, json, sys
out, variant = sys.argv[1], sys.argv[2]
res = {}
for p in sorted(glob.glob(out + f'/v{variant}_pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'k_fused' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = {'mean_per_launch': sum(v) / len(v), 'launches': len(v)}
json.dump(res, open(out + f'/pmc_k_fused_v{variant}.json', 'w'), indent=1)
print(variant, json.dumps(res))
PY
done
for f in $OUT/pmc_k_fused_v*.json; do cp $f $R/profiles/${TAG}_$(basename $f); done
python3 $R/bench.py --steps 1000 --warmup 20 --full-tables-reference > $OUT/bench_unprofiled.json 2> /dev/null  # -> profiles/<tag>_bench_line_full_tables.json
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv

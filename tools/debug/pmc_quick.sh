#!/bin/bash
# Counters of one kernel, summed over the launches of a short profiled run (one rocprofv3 --pmc pass per quoted group):
#   bash tools/debug/pmc_quick.sh population 144 k_pop_run "SQ_WAVES SQ_INSTS_VALU ..." "SQ_INSTS_SALU ..."
W=$1; STEPS=$2; KERNEL=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp LCF_BENCH_NO_E2E=1
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/tools/prof_kernel.py $W $STEPS > $OUT/p$i.log 2>&1 || echo "!! pass $i failed"
done
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, collections, sys
out, kern = sys.argv[1:3]
for p in sorted(glob.glob(out + '/p*/*/*_counter_collection.csv')):
    agg, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
    for k in agg:
        print(f'{k:32s} {agg[k]:16.0f}  over {n[k]} launches')
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmcq_$i -- python3 $R/tools/prof_kernel.py mcmc 5 > $R/gpurun_out/pmcq_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
res={}
for p in sorted(glob.glob(R+'/gpurun_out/pmcq_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'k_solo' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): res[k]=sum(v)/len(v)
for k,v in sorted(res.items()): print(f'{k:32s} {v:14.1f}')
PY
tail -3 $R/gpurun_out/pmcq_3.log
rm -rf $R/gpurun_out/pmcq_[0-9]

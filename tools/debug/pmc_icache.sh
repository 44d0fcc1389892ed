# Instruction-cache / fetch counters of the half-step kernel (one pass each; run through gpurun).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-mcmc}
K=${2:-k_solo}
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmci_$i -- python3 $R/tools/prof_kernel.py $W 5 > $R/gpurun_out/pmci_$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, collections, os, sys
R=os.environ['GRAFT_REPO_ROOT']
res={}
for p in sorted(glob.glob(R+'/gpurun_out/pmci_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if sys.argv[1] in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): res[k]=sum(v)/len(v)
for k,v in sorted(res.items()): print(f'{k:28s} {v:14.1f}')
PY
rm -rf $R/gpurun_out/pmci_[0-9]

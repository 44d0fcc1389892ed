cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
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
for k,v in sorted(res.items()): print(f'{k:28s} {v:14.1f}')
open(R+'/gpurun_out/pmcq.txt','w').write('\n'.join(f'{k} {v}' for k,v in sorted(res.items())))
PY
rm -rf $R/gpurun_out/pmcq_[0-9]

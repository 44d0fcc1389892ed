#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun, LAST in a round:
# the PMC summaries carry the hash of the kernel sources they were collected from, and bench.py uses them only while
# that hash matches the tree).
#   bash tools/collect_profiles.sh r04 [commit]
# kernel-trace/stats and every --pmc pass are separate runs, as the profiling guide prescribes.
TAG=${1:-r04}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
FINAL=$OUT/final   # everything that is to be committed under profiles/ (gpurun merges gpurun_out/ only)
# nothing of an earlier (possibly aborted) collection may be summarised as this tree's evidence
rm -rf $OUT/trace_* $OUT/pmc_* $FINAL
mkdir -p $OUT $FINAL
cd /tmp && export TMPDIR=/tmp
FAILED=0
declare -A STEPS=( [mcmc]=1000 [companion]=30 [population]=1000 [sed]=200 )
declare -A PSTEPS=( [mcmc]=64 [companion]=2 [population]=144 [sed]=1 )   # (mcmc: two launches of 64 half-steps; population: 32 + 4 x 64)
declare -A KERNEL=( [mcmc]=k_solo_run [companion]=k_solo [population]=k_pop_run [sed]=k_sed_interp )
declare -A PTAG=( [mcmc]=k_solo_run_mcmc [companion]=k_solo_companion [population]=population [sed]=k_sed )
# (the profiled bench runs must stay ONE process each: the end-to-end block of the headline line starts a child)
export LCF_BENCH_NO_E2E=1
for W in mcmc companion population sed; do
  echo "== $W: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 $R/bench.py --workload $W --steps ${STEPS[$W]} --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof_$W.json 2> $OUT/trace_$W.log \
    || { echo "!! $W: the kernel-trace run failed (see $OUT/trace_$W.log)"; FAILED=1; }
  cp $OUT/trace_$W/*/*_kernel_stats.csv $FINAL/${TAG}_kernel_stats_$W.csv 2>/dev/null || { echo "!! $W: no kernel statistics"; FAILED=1; }
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    echo "== $W: pmc pass $i"
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$i -- python3 $R/tools/prof_kernel.py $W ${PSTEPS[$W]} > $OUT/pmc_${W}_$i.log 2>&1 \
      || { echo "!! $W: pmc pass $i failed (see $OUT/pmc_${W}_$i.log)"; FAILED=1; }
  done
  python3 - "$OUT" "$W" "${KERNEL[$W]}" "$FINAL/${TAG}_pmc_${PTAG[$W]}.json" "$COMMIT" "$R" <<'PY'
import csv, glob, collections, json, re, sys
out, w, kern, dst, commit, root = sys.argv[1:7]
sys.path.insert(0, root)
import bench
res, name, timed, hs = {}, None, None, None
# (a profiled run that prints a bench line with resident launches: only the launches of its TIMED steps are averaged --
# the last 2 steps / half_steps_per_launch of them -- and the summary says how many half-steps such a launch covers)
for log in sorted(glob.glob(out + f'/pmc_{w}_*.log')):
    for ln in open(log, errors='replace'):
        if ln.startswith('{') and '"half_steps_per_launch"' in ln:
            d = json.loads(ln)
            hs = d['roofline']['half_steps_per_launch']
            timed = round(2 * d['steps'] / hs)
for p in sorted(glob.glob(out + f'/pmc_{w}_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(p)), key=lambda r: int(r['Dispatch_Id'])):
        if kern in r['Kernel_Name']:
            m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
            name = m.group(1) if m else r['Kernel_Name'][:60]
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        v = v[-timed:] if timed else v
        res[k] = {'mean_per_launch': sum(v) / len(v), 'launches': len(v)}
if not res:
    raise SystemExit(f'!! {w}: no counter rows for kernel {kern}: no summary written')
json.dump({'workload': w, 'kernel': name, 'collected_at_commit': commit, 'kernel_source_sha256': bench.kernel_source_sha(),
           'half_steps_per_launch': hs, 'counters': res}, open(dst, 'w'), indent=1)
print(w, name, {k: round(v['mean_per_launch'], 1) for k, v in res.items()})
PY
  [ $? -eq 0 ] || FAILED=1
done
echo "== unprofiled bench lines"
unset LCF_BENCH_NO_E2E
cd $R
mkdir -p profiles && cp $FINAL/${TAG}_pmc_*.json profiles/ 2>/dev/null
for W in mcmc companion population sed; do
  python3 bench.py --workload $W --steps ${STEPS[$W]} --warmup 5 > $FINAL/${TAG}_bench_$W.json 2> /dev/null
done
python3 bench.py --steps 1000 --warmup 5 --variant 2 --no-cpu-baseline > $FINAL/${TAG}_bench_mcmc_compressed_tables.json 2> /dev/null
python3 bench.py --steps 1000 --warmup 5 --variant 1 --no-cpu-baseline > $FINAL/${TAG}_bench_mcmc_full_tables.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $FINAL/${TAG}_bench_mcmc_20_steps.json 2> /dev/null
rm -rf $OUT/trace_* $OUT/pmc_*_[0-9]
ls -la $FINAL
[ $FAILED -eq 0 ] || { echo "!! the collection is incomplete"; exit 1; }

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 3 2; do
python bench.py --workload population --steps 2000 --warmup 5 --variant $v --no-cpu-baseline > gpurun_out/r2_popv$v.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_poptrace$v -- python3 bench.py --workload population --steps 300 --warmup 5 --variant $v --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/r2_poptrace$v/*/*kernel_stats.csv gpurun_out/r2_pop_stats_v$v.csv; rm -rf gpurun_out/r2_poptrace$v
done

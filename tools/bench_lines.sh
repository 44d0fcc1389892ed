#!/bin/bash
# The unprofiled bench lines of profiles/<tag>_bench_*.json alone (after a change of bench.py that leaves the kernels,
# and with them the PMC summaries of tools/collect_profiles.sh, as they are).   bash tools/bench_lines.sh r03
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lines_$TAG
mkdir -p $OUT
cd $R
declare -A STEPS=( [mcmc]=1000 [companion]=30 [population]=300 [sed]=200 )
for W in mcmc companion population sed; do
  python3 bench.py --workload $W --steps ${STEPS[$W]} --warmup 5 > $OUT/${TAG}_bench_$W.json 2> /dev/null
done
python3 bench.py --steps 1000 --warmup 5 --variant 2 --no-cpu-baseline > $OUT/${TAG}_bench_mcmc_compressed_tables.json 2> /dev/null
python3 bench.py --steps 1000 --warmup 5 --variant 1 --no-cpu-baseline > $OUT/${TAG}_bench_mcmc_full_tables.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_mcmc_20_steps.json 2> /dev/null
python3 tools/debug/show_bench.py $OUT/*.json

set -e
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $R/gpurun_out/t.log 2>&1 || { tail -30 $R/gpurun_out/t.log; exit 1; }
tail -2 $R/gpurun_out/t.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/emu8 -o emu -- python3 $R/tools/emulate_ranks.py 8 50 > $R/gpurun_out/emu8.log 2>&1
grep emulated $R/gpurun_out/emu8.log
python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/b.log 2>&1; tail -1 $R/gpurun_out/b.log | cut -c1-330
LCF_BENCH_FORCE_SHARDED=1 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bs.log 2>&1; tail -1 $R/gpurun_out/bs.log | cut -c1-330

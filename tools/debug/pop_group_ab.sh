export LCF_HIP_LIB=$PWD/build_variants/liblcf_dev.so LCF_BENCH_NO_E2E=1
for g in 4 8 10 12; do
  for it in 1 0; do
    LCF_POP_GROUP=$g LCF_POP_ITAB_LDS=$it timeout -k 10 200 python bench.py --workload population --no-cpu-baseline > gpurun_out/pb_${g}_${it}.log 2>&1
    python - <<PY
import json
for ln in open("gpurun_out/pb_${g}_${it}.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print("group $g itab_lds $it", "%.3e" % d["value"], "%.2f us/half-step" % (1e3*d["roofline"]["kernel_ms_per_half_step"]), d["roofline"]["kernel"][:12], flush=True)
PY
  done
  LCF_POP_GROUP=$g timeout -k 10 120 python tools/debug/pop_run_mismatch.py 5 70 6 60 2>&1 | grep -c "same | acceptance same: True"
done

# round-5 final measurements (writes gpurun_out/r05_*; the ones to keep are copied into profiles/ afterwards).  Usage: RH_COMMIT=<hash> bash tools/r05_final.sh [human|pmc|others|all]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; MODE=${1:-all}
show() { for f in "$@"; do python - <<PY
import json
try:
    d=json.loads(open("$O/r05_$f.json").read().strip().splitlines()[-1]); cb=d.get("cpu_baseline") or {}
    print("$f", d["value"], d["ms_per_step"], "h2d", d.get("value_h2d_included"), d.get("value_h2d_full_upload"), "cpu", cb.get("value"), cb.get("threads"), "stock", cb.get("value_stock_flags"), "paf", d.get("paf_sample_identical"), (d.get("roofline") or {}).get("frac"), (d.get("path") or {}).get("frac_of_hbm_peak"))
except Exception as e: print("$f", "ERR", e)
PY
done; }
if [ "$MODE" = "human" ] || [ "$MODE" = "all" ]; then
  cd $R
  timeout 1500 python bench.py --steps 5 --warmup 2 2>$O/r05_human_bench.err | tail -1 > $O/r05_human_bench.json
  RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r05_human_bench_1stream.json
  cd /tmp; rm -rf /tmp/prof3 /tmp/prof1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d >/dev/null 2>&1
  cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/r05_human_kernel_stats_3streams.csv
  RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d >/dev/null 2>&1
  cp $(find /tmp/prof1 -name "*kernel_stats.csv" | head -1) $O/r05_human_kernel_stats_1stream.csv
  timeout 900 python $R/profiles/collect_stage_kernels.py $O/r05_human_stage_kernels_1stream.json 2>&1 | head -3
  show human_bench human_bench_1stream
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
  cd /tmp
  timeout 1500 python $R/profiles/collect_pmc.py > $O/r05_pmc_summary.log 2>&1; cp $R/profiles/pmc_traffic.json $O/r05_pmc_traffic.json; tail -32 $O/r05_pmc_summary.log
fi
if [ "$MODE" = "others" ] || [ "$MODE" = "all" ]; then
  cd $R
  timeout 900 python bench.py --workload ecoli --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r05_ecoli_bench.json
  timeout 900 python bench.py --workload dmel --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/r05_dmel_bench.json
  timeout 900 python bench.py --workload ava --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r05_ava_bench.json
  show ecoli_bench dmel_bench ava_bench
fi

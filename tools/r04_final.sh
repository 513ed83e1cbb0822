# round-4 final measurements (writes gpurun_out/r04_*; the ones to keep are copied into profiles/ afterwards).  Usage: bash tools/r04_final.sh [pmc|all]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; MODE=${1:-all}
cd /tmp
timeout 1500 python $R/profiles/collect_pmc.py > $O/r04_pmc.log 2>&1; cp $R/profiles/pmc_traffic.json $O/r04_pmc_traffic.json; tail -32 $O/r04_pmc.log
if [ "$MODE" = "pmc" ]; then exit 0; fi
cd $R
timeout 1500 python bench.py --steps 5 --warmup 2 2>$O/r04_human_bench.err | tail -1 > $O/r04_human_bench.json
RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r04_human_bench_1stream.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d >/dev/null 2>&1
cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/r04_human_kernel_stats_3streams.csv
RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d >/dev/null 2>&1
cp $(find /tmp/prof1 -name "*kernel_stats.csv" | head -1) $O/r04_human_kernel_stats_1stream.csv
cd $R
timeout 900 python bench.py --workload ecoli --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r04_ecoli_bench.json
timeout 900 python bench.py --workload dmel --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/r04_dmel_bench.json
timeout 1200 python bench.py --workload dmel --reads 1000000 --steps 1 --warmup 0 --cpu-sample 4000 --no-h2d 2>/dev/null | tail -1 > $O/r04_dmel_1M_bench.json
timeout 900 python bench.py --workload ava --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r04_ava_bench.json
for f in human_bench human_bench_1stream ecoli_bench dmel_bench dmel_1M_bench ava_bench; do python - <<PY
import json
try:
    d=json.load(open("$O/r04_$f.json")); cb=d.get("cpu_baseline") or {}
    print("$f", d["value"], d["ms_per_step"], "h2d", d.get("value_h2d_included"), d.get("value_h2d_full_upload"), "cpu", cb.get("value"), cb.get("threads"), "paf", d.get("paf_sample_identical"), (d.get("roofline") or {}).get("frac"))
except Exception as e: print("$f", "ERR", e)
PY
done

#!/bin/bash
# One script for the measurements behind profiles/: bench lines, per-kernel / per-stage tables, PMC passes.  Runs on the GPU box through gpurun from the repo root:
#   tools/gpu_retry.sh 1500 /tmp/log bash tools/measure.sh <mode> <tag> [args]
# Everything lands in gpurun_out/<tag>_*; what is worth keeping is copied to profiles/ by hand.  Modes:
#   one    [filter] [pytest -k expr]  parity subset on the GPU, the 3-stream bench line, the 1-stream rocprofv3 kernel table (rows matching `filter` printed)
#   stage  [bench args]               3-stream bench line + device time per kernel AND stage of one step on one stream (profiles/collect_stage_kernels.py)
#   check                              the round-end checks as the driver runs them: pytest -m gpu, smoke(), default bench line
#   final  human|others|pmc|all        the round's artefacts: bench lines of every configuration, kernel tables on 1 / 3 streams, stage table, PMC traffic
#   mix                                instruction mix + LDS conflict counters per kernel (two --pmc passes, one stream)
#   icache                             instruction-cache requests / misses per kernel
#   trace                              per-level trace of the multi-workgroup sorter (needs a -DRH_DEV build: RH_HIPCC_EXTRA=-DRH_DEV python -m rawhash_amd.build --force)
#   kprof                              phases of the LDS block sorter on the anchor sort (needs a -DRH_KPROF build)
# WORKLOAD=ecoli|dmel|ava (default: human) selects the bench workload for one / stage / mix.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; MODE=${1:-one}; TAG=${2:-m}; shift 2 2>/dev/null; mkdir -p $O
WL=${WORKLOAD:+--workload $WORKLOAD}; [ -n "$WORKLOAD" ] && export RH_PMC_WORKLOAD=$WORKLOAD
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); cb = d.get("cpu_baseline") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "h2d", d.get("value_h2d_included"), "cpu", cb.get("value"), cb.get("cores"), "paf", d.get("paf_sample_identical"), "frac", (d.get("roofline") or {}).get("frac"), "path", (d.get("path") or {}).get("frac_of_hbm_peak"))
    except Exception as e:
        print(f, "ERR", e)
PY
}
ktable() {  # csv filter
python - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); pat = sys.argv[2]
tot = sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("k_ix", "k_synth", "void k_ix", "void k_synth"))) / 1e6
print("1-stream kernel total", round(tot))
for r in rows:
    if pat in r["Name"] and float(r["TotalDurationNs"]) > 2e6: print("  ", r["Name"].replace("void ", "")[:64], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 1))
PY
}
prof1() {  # out.csv, bench args...: rocprofv3 kernel stats of one step on one stream
  out=$1; shift; rm -rf /tmp/pf_$TAG
  RH_SUB_BATCHES=1 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$TAG -o p -- python $R/bench.py $WL --steps 1 --warmup 0 --cpu-sample 0 --no-h2d "$@" > $O/${TAG}_1s.json 2>/dev/null
  cp $(find /tmp/pf_$TAG -name "*kernel_stats.csv" | head -1) $out
}
pmc() {  # name counters...: one --pmc pass, per-kernel sums of the counters, one stream
  name=$1; shift; rm -rf /tmp/pm_$name
  RH_SUB_BATCHES=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pm_$name -o p -- python $R/bench.py $WL --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>$O/${TAG}_$name.err
  f=$(find /tmp/pm_$name -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$name: no counters collected"; tail -3 $O/${TAG}_$name.err; return; }
  python - "$f" "$O/${TAG}_pmc_$name.txt" "$@" <<'PY'
import csv, collections, sys
f, out, names = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get(names[0], 0))[:16]
with open(out, "w") as o:
    hdr = "%-56s" % "kernel" + "".join("%22s" % n for n in names)
    print(hdr); o.write(hdr + "\n")
    for k, v in rows:
        l = "%-56s" % k[:56] + "".join("%22.0f" % v.get(n, 0) for n in names)
        print(l); o.write(l + "\n")
PY
}
case $MODE in
one)
  PAT=${1:-k_sort}; KEXPR=${2:-"sort or golden or config2 or repeat or regions or human"}
  cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "$KEXPR" 2>&1 | tail -3
  timeout 600 python bench.py $WL --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_3s.json
  cd /tmp; prof1 $O/${TAG}_kernel_stats_1stream.csv
  line $O/${TAG}_3s.json $O/${TAG}_1s.json; ktable $O/${TAG}_kernel_stats_1stream.csv "$PAT" ;;
stage)
  cd $R; timeout 600 python bench.py $WL --steps 3 --warmup 1 --cpu-sample 0 --no-h2d "$@" 2>/dev/null | tail -1 > $O/${TAG}_3s.json; line $O/${TAG}_3s.json
  cd /tmp; timeout 900 python $R/profiles/collect_stage_kernels.py $O/${TAG}_stage_kernels_1stream.json 2>&1 | tail -60 ;;
check)
  cd $R
  timeout -k 10 1800 python -m pytest tests -q -m gpu --timeout 900 > $O/${TAG}_pytest.log 2>&1; grep -E "passed|failed|error" $O/${TAG}_pytest.log | tail -3
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout -k 10 900 python bench.py > $O/${TAG}_bench.out 2>$O/${TAG}_bench.err; tail -1 $O/${TAG}_bench.out > $O/${TAG}_bench.json; line $O/${TAG}_bench.json ;;
final)
  WHAT=${1:-all}
  if [ $WHAT = human ] || [ $WHAT = all ]; then
    cd $R
    timeout 1500 python bench.py --steps 5 --warmup 2 2>$O/${TAG}_human_bench.err | tail -1 > $O/${TAG}_human_bench.json
    RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_human_bench_1stream.json
    cd /tmp; rm -rf /tmp/prof3
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d >/dev/null 2>&1
    cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/${TAG}_human_kernel_stats_3streams.csv
    prof1 $O/${TAG}_human_kernel_stats_1stream.csv
    timeout 900 python $R/profiles/collect_stage_kernels.py $O/${TAG}_human_stage_kernels_1stream.json 2>&1 | head -3
    line $O/${TAG}_human_bench.json $O/${TAG}_human_bench_1stream.json
  fi
  if [ $WHAT = pmc ] || [ $WHAT = all ]; then
    cd /tmp; timeout 1800 python $R/profiles/collect_pmc.py > $O/${TAG}_pmc_summary.log 2>&1; cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json; tail -34 $O/${TAG}_pmc_summary.log
  fi
  if [ $WHAT = others ] || [ $WHAT = all ]; then
    cd $R
    timeout 900 python bench.py --workload ecoli --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/${TAG}_ecoli_bench.json
    timeout 900 python bench.py --workload dmel --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_dmel_bench.json
    timeout 900 python bench.py --workload ava --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/${TAG}_ava_bench.json
    line $O/${TAG}_ecoli_bench.json $O/${TAG}_dmel_bench.json $O/${TAG}_ava_bench.json
  fi ;;
mix)
  pmc insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc lds SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU ;;
icache)
  pmc icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVES SQ_BUSY_CYCLES ;;
trace)
  cd $R; RH_SUB_BATCHES=1 RH_BS_TRACE=1 timeout 900 python bench.py $WL --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > $O/${TAG}_line.json 2> $O/${TAG}_bs_trace.log; grep -c "BS level" $O/${TAG}_bs_trace.log ;;
kprof)
  timeout 800 python $R/tools/kprof_human.py "$@" 2>&1 | grep slot > $O/${TAG}_kprof_sort.txt; cat $O/${TAG}_kprof_sort.txt ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac

# 1-stream kernel table only.  Usage: bash tools/r03_prof1.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-x}
RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profh_$TAG -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>&1
cp $(find /tmp/profh_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_1stream.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_kernel_stats_1stream.csv")))
for r in rows[:26]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), round(float(r["TotalDurationNs"])/1e6,1))
PY

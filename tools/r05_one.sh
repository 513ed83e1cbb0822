# sorter / golden parity on the GPU, the 3-stream bench line and the 1-stream kernel table of the build in the tree.  Usage: bash tools/r05_one.sh <tag> [kernel-name filter]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-one}; PAT=${2:-k_bs_}; mkdir -p $O
cd $R; timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "sort or golden or config2 or repeat or regions or human" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("3-stream", d["value"], d["ms_per_step"])
PY
cd /tmp; rm -rf /tmp/pf_$TAG
RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$TAG -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > $O/${TAG}_1s.json 2>/dev/null
cp $(find /tmp/pf_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python - <<PY
import csv, json
try:
    d=json.loads(open("$O/${TAG}_1s.json").read().strip().splitlines()[-1]); print("1-stream", d["value"], d["ms_per_step"])
except Exception as e: print("1-stream line:", e)
rows=list(csv.DictReader(open("$O/${TAG}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("k_ix","k_synth")))/1e6
print("1-stream kernel total", round(tot))
for r in rows:
    if "$PAT" in r["Name"] and float(r["TotalDurationNs"]) > 2e6: print("  ", r["Name"].replace("void ","")[:60], r["Calls"], round(float(r["TotalDurationNs"])/1e6,1))
PY

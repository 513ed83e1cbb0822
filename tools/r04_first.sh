# first call of a session: whole -m gpu suite, default bench line (driver form), 1-stream kernel table.  Usage: bash tools/r04_first.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-first}; mkdir -p $O
cd $R; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/${TAG}_pytest.log
timeout 900 python bench.py 2>$O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench.json")); print("driver-form", d["value"], d["ms_per_step"], d.get("roofline"), d.get("cpu_baseline"), d.get("paf_sample_identical"))
PY
cd /tmp; rm -rf /tmp/pf_$TAG
RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$TAG -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > $O/${TAG}_1s.json 2>/dev/null
cp $(find /tmp/pf_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_kernel_stats.csv")))
rows=[r for r in rows if not r["Name"].startswith(("k_ix","k_synth"))]
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6
print("1-stream total", round(tot))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print("%-60s %5s %9.1f"%(r["Name"].replace("void ","")[:60], r["Calls"], float(r["TotalDurationNs"])/1e6))
PY

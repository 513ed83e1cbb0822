# where an D. melanogaster-scale step goes: kernel table (one stream) + round trace.  Usage: bash tools/r04_ecoli_prof.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/pf_dm
RH_SUB_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_dm -o p -- python $R/bench.py --workload dmel --steps 2 --warmup 1 --cpu-sample 0 --no-h2d > $O/dm_1s.json 2>/dev/null
cp $(find /tmp/pf_dm -name "*kernel_stats.csv" | head -1) $O/dm_kernel_stats.csv
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/dm_kernel_stats.csv")))
rows=[r for r in rows if not r["Name"].startswith(("k_ix","k_synth"))]
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6; calls=sum(int(r["Calls"]) for r in rows)
d=json.loads(open("$O/dm_1s.json").read().strip().splitlines()[-1])
print("1-stream", d["value"], d["ms_per_step"], "kernel total ms (3 steps)", round(tot), "launches", calls)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:25]:
    print("%-60s %5s %9.2f"%(r["Name"].replace("void ","")[:60], r["Calls"], float(r["TotalDurationNs"])/1e6))
PY

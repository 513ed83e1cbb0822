# kernel table of the DTW / RMQ modes (E. coli scale, one stream).  Usage: bash tools/r04_dtw_prof.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
for m in dtw rmq; do
rm -rf /tmp/pf_$m
RH_SUB_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$m -o p -- python $R/bench.py --workload ecoli --reads 20000 --mapopt $m --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > $O/${m}_1s.json 2>/dev/null
cp $(find /tmp/pf_$m -name "*kernel_stats.csv" | head -1) $O/${m}_kernel_stats.csv
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/${m}_kernel_stats.csv")))
rows=[r for r in rows if not r["Name"].startswith(("k_ix","k_synth"))]
d=json.loads(open("$O/${m}_1s.json").read().strip().splitlines()[-1])
print("$m 1-stream", d["value"], d["ms_per_step"])
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:8]:
    print("%-60s %5s %9.2f"%(r["Name"].replace("void ","")[:60], r["Calls"], float(r["TotalDurationNs"])/1e6))
PY
done

# sorter parity + 1-stream kernel table for the two token-walker pops (RH_BS_TOK_ADV=0: compiler-scheduled; 1: rh_tok_advance + one period counter).  Usage: bash tools/r03_tokab.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-tokab}; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or golden or config2 or repeat" 2>&1 | tail -2; cd /tmp
for v in "RH_BS_TOK_ADV=0" "RH_BS_TOK_ADV=1"; do
  rm -rf /tmp/pf_$TAG
  env $v RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$TAG -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>&1
  cp $(find /tmp/pf_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_${v#*=}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_${v#*=}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("k_ix","k_synth")))/1e6
print("$v", "total", round(tot), [ (r["Name"][:28], round(float(r["TotalDurationNs"])/1e6,1)) for r in rows if "walk" in r["Name"] or "k_bs_scatter(" in r["Name"]])
PY
done

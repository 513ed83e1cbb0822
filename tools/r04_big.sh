# batches beyond the default: 131 072 and 262 144 reads per call (does the rate still fall beyond ~80 k reads?).  Usage: bash tools/r04_big.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in 32768 131072 262144; do
  timeout 900 python bench.py --reads $n --steps 2 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/big_$n.json
  python - <<PY
import json
d=json.load(open("$O/big_$n.json")); print($n, d["value"], d["ms_per_step"])
PY
done

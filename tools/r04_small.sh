# small batches (the reference's -K mini-batch is ~12.5 k reads of 40 k samples): sub-batch streams 1 / 2 / 3.  Usage: bash tools/r04_small.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in 8192 16384 32768; do for sb in 1 2 3; do
  RH_SUB_BATCHES=$sb timeout 600 python bench.py --reads $n --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/small.json
  python - <<PY
import json
d=json.load(open("$O/small.json")); print($n, "sub", $sb, d["value"], d["ms_per_step"])
PY
done; done

# sub-batch streams per call with the equal memory allowances.  Usage: bash tools/r04_sub.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for sb in 2 4 3; do
  RH_SUB_BATCHES=$sb timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/sub.json
  python - <<PY
import json
d=json.load(open("$O/sub.json")); print("sub", $sb, d["value"], d["ms_per_step"])
PY
done

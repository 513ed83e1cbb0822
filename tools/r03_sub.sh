# sub-batch streams sweep.  Usage: bash tools/r03_sub.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-sub}; mkdir -p $O
cd $R
for v in "RH_SUB_BATCHES=2" "RH_SUB_BATCHES=3" "RH_SUB_BATCHES=4"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
  python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("$v", d["value"], d["ms_per_step"])
PY
done

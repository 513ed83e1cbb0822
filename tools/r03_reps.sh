# what a sorter pass costs the 3-stream step: walks / placement launched twice (idempotent).  Usage: bash tools/r03_reps.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-reps}; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort" 2>&1 | tail -2
for v in "A=1" "RH_BS_WALK_REPS=2" "RH_BS_SCAT_REPS=2" "RH_SUB_BATCHES=1" "RH_SUB_BATCHES=1 RH_BS_WALK_REPS=2" "RH_SUB_BATCHES=1 RH_BS_SCAT_REPS=2"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
  python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("$v", d["value"], d["ms_per_step"])
PY
done

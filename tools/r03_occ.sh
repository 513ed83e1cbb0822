# occupancy caps of the long-lived one-wavefront kernels (unused dynamic LDS) against the 3-stream step.  Usage: bash tools/r03_occ.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-occ}; mkdir -p $O
cd $R
for v in "A=1" "RH_BS_WALK_LDS=6144" "RH_BS_WALK_LDS=16384" "RH_BS_WALK_LDS=36864" "RH_WAVE_LDS=8192" "RH_WAVE_LDS=16384" "RH_BS_WALK_LDS=16384 RH_WAVE_LDS=16384" "RH_SUB_BATCHES=1 RH_BS_WALK_LDS=16384"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
  python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("$v", d["value"], d["ms_per_step"])
PY
done

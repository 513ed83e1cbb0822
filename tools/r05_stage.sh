# 3-stream bench line + device time per kernel and stage of one human-scale step on one stream.  Usage: bash tools/r05_stage.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-stage}; mkdir -p $O
cd $R; timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("3-stream", d["value"], d["ms_per_step"])
PY
cd /tmp; timeout 900 python $R/profiles/collect_stage_kernels.py $O/${TAG}_stage_kernels.json 2>&1 | tail -50

# full GPU suite + 3-stream bench line.  Usage: bash tools/r04_full.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-full}; mkdir -p $O
cd $R; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("3-stream", d["value"], d["ms_per_step"])
PY

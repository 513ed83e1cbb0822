# whole -m gpu suite under a tight limit + bench line.  Usage: bash tools/r04_suite.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/suite_x.json
python - <<PY
import json
d=json.load(open("$O/suite_x.json")); print("3-stream", d["value"], d["ms_per_step"])
PY

# state check: gpu tests, 3-stream bench (no CPU sample), 1-stream kernel table.  Usage: bash tools/r03_now.sh <tag> [notest]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-now}; mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/${TAG}_gputest.log; cat $O/${TAG}_gputest.log; fi
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>$O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json
RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_bench_1stream.json
bash tools/r03_prof1.sh $TAG
python - <<PY
import json
for f in ("bench","bench_1stream"):
    try:
        d=json.load(open("$O/${TAG}_%s.json"%f)); print(f, d["value"], d["ms_per_step"], {k:round(v) for k,v in d["stage_ms_per_step"].items()})
    except Exception as e: print(f,"ERR",e)
PY

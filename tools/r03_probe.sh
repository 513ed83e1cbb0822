# round-3 probe: 3-stream bench (no CPU leg), 1-stream bench, 1-stream kernel table, sorter level trace.  Usage: bash tools/r03_probe.sh <tag> [pytest -k expr]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-base}
cd $R
if [ -n "$2" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" 2>&1 | tail -3 | tee $O/${TAG}_pytest.log; fi
timeout 600 python bench.py --cpu-sample 0 --no-h2d 2>$O/${TAG}_bench3.err | tail -1 > $O/${TAG}_bench3.json
RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_bench1.json
cd /tmp
RH_SUB_BATCHES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profh_$TAG -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>&1
cp $(find /tmp/profh_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_1stream.csv
cd $R && RH_SUB_BATCHES=1 RH_BS_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d 2>$O/${TAG}_bstrace.log >/dev/null
python - <<PY
import json
for f in ("bench3","bench1"):
    try:
        d=json.load(open("$O/${TAG}_%s.json"%f)); print(f, d["value"], d["ms_per_step"], d.get("paf_sample_identical"))
    except Exception as e: print(f, "ERR", e)
PY
head -24 $O/${TAG}_kernel_stats_1stream.csv | cut -c1-150

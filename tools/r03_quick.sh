# sort parity on the GPU + 3-stream and 1-stream bench lines.  Usage: bash tools/r03_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-q}; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or golden or config2 or repeat" 2>&1 | tail -3
for v in "A=1" "RH_SUB_BATCHES=1"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/${TAG}_x.json
  python - <<PY
import json
d=json.load(open("$O/${TAG}_x.json")); print("$v", d["value"], d["ms_per_step"], {k:round(v) for k,v in d["stage_ms_per_step"].items() if v>=30})
PY
done

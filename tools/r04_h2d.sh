# consumed-prefix staging: parity on the GPU + E. coli-scale and human-scale bench lines with the upload inside the timed region.  Usage: bash tools/r04_h2d.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-h2d}; mkdir -p $O
cd $R; timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "consumed_prefix or stage_regions or blow5 or cabi" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_cabi.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --workload ecoli --steps 5 --warmup 1 --cpu-sample 0 2>$O/${TAG}_ecoli.err | tail -1 > $O/${TAG}_ecoli.json
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>$O/${TAG}_human.err | tail -1 > $O/${TAG}_human.json
python - <<PY
import json
for w in ("ecoli","human"):
    try:
        d=json.load(open("$O/${TAG}_%s.json"%w)); print(w, "resident", d["value"], "h2d", d["value_h2d_included"], "full upload", d.get("value_h2d_full_upload"), "ms", d["ms_per_step"], d["ms_per_step_h2d_included"])
    except Exception as e: print(w, "ERR", e); print(open("$O/${TAG}_%s.err"%w).read()[-1500:])
PY

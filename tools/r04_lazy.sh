# compact_a's last copy left out (rr.lazy_reorder): parity at human scale first, under tight limits, then the bench line.  Usage: bash tools/r04_lazy.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "human_scale_index_vs_reference" 2>&1 | tail -2 || exit 1
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeat_rich or golden_paf or large_genome or large_batch" 2>&1 | tail -2
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/lazy_x.json
python - <<PY
import json
d=json.load(open("$O/lazy_x.json")); print("3-stream", d["value"], d["ms_per_step"], d["stage_ms_per_step"]["backtrack"])
PY

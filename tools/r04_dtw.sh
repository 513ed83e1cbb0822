# DTW mode with the long stretches' bands across the lanes: parity + bench line.  Usage: bash tools/r04_dtw.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dtw or golden" 2>&1 | tail -3
timeout 900 python bench.py --workload ecoli --reads 20000 --mapopt dtw --steps 2 --warmup 1 --cpu-sample 6000 --no-h2d 2>$O/r04_ecoli_dtw.err | tail -1 > $O/r04_ecoli_dtw.json
python - <<PY
import json
d=json.load(open("$O/r04_ecoli_dtw.json")); cb=d.get("cpu_baseline") or {}
print("dtw", d["value"], d["ms_per_step"], "cpu", cb.get("value"), cb.get("threads"), "paf", d.get("paf_sample_identical"), d["stage_ms_per_step"]["regions"])
PY

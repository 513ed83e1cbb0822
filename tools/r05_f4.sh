# bench lines of the RMQ chaining variants (E. coli / D. mel scale) + their goldens on the GPU.  Usage: bash tools/r05_f4.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "golden or rmq" 2>&1 | grep -E "passed|failed|error" | tail -2
for m in rmq bw_long; do
  timeout 900 python bench.py --workload ecoli --reads 20000 --mapopt $m --steps 2 --warmup 1 --cpu-sample 6000 --no-h2d 2>$O/r05_ecoli_$m.err | tail -1 > $O/r05_ecoli_$m.json
done
timeout 900 python bench.py --workload dmel --reads 8000 --mapopt rmq --steps 1 --warmup 1 --cpu-sample 3000 --no-h2d 2>$O/r05_dmel_rmq.err | tail -1 > $O/r05_dmel_rmq.json
python - <<PY
import json
for f in ["ecoli_rmq","ecoli_bw_long","dmel_rmq"]:
    try:
        d=json.loads(open("$O/r05_%s.json"%f).read().strip().splitlines()[-1]); cb=d.get("cpu_baseline") or {}
        print(f, d["value"], d["ms_per_step"], "cpu", cb.get("value"), cb.get("threads"), "paf", d.get("paf_sample_identical"), d.get("cpu_baseline_error"))
    except Exception as e: print(f, "ERR", e)
PY

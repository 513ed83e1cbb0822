# bench lines of the f4 chaining variants (E. coli-scale index) + sub-batch sweep of the headline.  Usage: bash tools/r04_f4.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in rmq bw_long dtw; do
  timeout 900 python bench.py --workload ecoli --reads 20000 --mapopt $m --steps 2 --warmup 1 --cpu-sample 6000 --no-h2d 2>$O/r04_ecoli_$m.err | tail -1 > $O/r04_ecoli_$m.json
done
timeout 900 python bench.py --workload dmel --reads 8000 --mapopt rmq --steps 1 --warmup 1 --cpu-sample 3000 --no-h2d 2>$O/r04_dmel_rmq.err | tail -1 > $O/r04_dmel_rmq.json
for sb in 2 4; do
  RH_SUB_BATCHES=$sb timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r04_sub$sb.json
done
python - <<PY
import json
for f in ["ecoli_rmq","ecoli_bw_long","ecoli_dtw","dmel_rmq","sub2","sub4"]:
    try:
        d=json.load(open("$O/r04_%s.json"%f)); cb=d.get("cpu_baseline") or {}
        print(f, d["value"], d["ms_per_step"], "cpu", cb.get("value"), cb.get("threads"), "paf", d.get("paf_sample_identical"), d.get("cpu_baseline_error"))
    except Exception as e: print(f, "ERR", e)
PY

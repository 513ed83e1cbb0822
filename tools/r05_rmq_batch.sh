# --rmq is bound by the longest read of a call (one wavefront walks a read's trees): throughput against reads per call.  Usage: bash tools/r05_rmq_batch.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() {  # workload reads cpu-sample
  timeout -k 10 600 python bench.py --workload $1 --reads $2 --mapopt rmq --steps 1 --warmup 1 --pool 2 --cpu-sample $3 --no-h2d > $O/r05_$1_rmq_n$2.out 2>$O/r05_$1_rmq_n$2.err
  tail -1 $O/r05_$1_rmq_n$2.out > $O/r05_$1_rmq_n$2.json
}
run dmel 8000 0; run dmel 24000 0; run dmel 48000 3000; run ecoli 65536 6000
python - <<PY
import json
for f in ("dmel_rmq_n8000", "dmel_rmq_n24000", "dmel_rmq_n48000", "ecoli_rmq_n65536"):
    try:
        d=json.loads(open("$O/r05_%s.json"%f).read()); cb=d.get("cpu_baseline") or {}
        print(f, d["value"], d["ms_per_step"], "cpu", cb.get("value"), "paf", d.get("paf_sample_identical"))
    except Exception as e: print(f, "ERR", e)
PY

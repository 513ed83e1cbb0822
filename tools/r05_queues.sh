# hardware queues x sub-batch streams x calls in flight.  Usage: bash tools/r05_queues.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { # reads steps hwq sub flight
  GPU_MAX_HW_QUEUES=$3 RH_SUB_BATCHES=$4 RH_BENCH_IN_FLIGHT=$5 timeout 600 python bench.py --reads $1 --steps $2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > $O/q.json
  python - <<PY
import json
d=json.load(open("$O/q.json")); print("reads $1 hwq $3 sub $4 in-flight $5:", d["value"], d["value_h2d_included"])
PY
}
run 65536 3 8 3 1
run 65536 3 8 4 1
run 65536 3 8 6 1
run 65536 3 4 4 1
run 12500 12 8 3 2
run 12500 12 8 2 2
run 12500 12 4 3 2
run 12500 12 8 6 1

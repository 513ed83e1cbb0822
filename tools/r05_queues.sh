# hardware queues x calls in flight.  Usage: bash tools/r05_queues.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { # reads steps hwq flight
  GPU_MAX_HW_QUEUES=$3 RH_BENCH_IN_FLIGHT=$4 timeout 600 python bench.py --reads $1 --steps $2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > $O/q.json
  python - <<PY
import json
d=json.load(open("$O/q.json")); print("reads $1 hwq $3 in-flight $4:", d["value"], d["value_h2d_included"])
PY
}
run 12500 12 4 2
run 12500 12 8 2
run 12500 12 6 2
run 65536 4 4 1
run 65536 4 8 1

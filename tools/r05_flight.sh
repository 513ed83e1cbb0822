# round traces of small calls with one / two calls in flight (-DRH_DEV build).  Usage: bash tools/r05_flight.sh <reads> <steps>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
N=${1:-12500}; K=${2:-6}
for fl in 1 2; do
  RH_TRACE_ROUNDS=1 RH_BENCH_IN_FLIGHT=$fl timeout 600 python bench.py --reads $N --steps $K --warmup 1 --cpu-sample 0 > $O/flight${fl}.json 2> $O/flight${fl}.log
  python - <<PY
import json
d=json.loads(open("$O/flight${fl}.json").read().strip().splitlines()[-1]); print("in flight", $fl, d["value"], d["value_h2d_included"], d["ms_per_step_h2d_included"])
PY
done

# reads per call on the human index: resident rate, upload-inclusive rate with one and with two calls in flight (rh_map_submit / rh_map_wait).
# Usage: bash tools/r05_batchsize.sh [sizes...]  -> gpurun_out/r05_batchsize.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
SIZES=${@:-8192 12500 16384 32768 65536 131072 262144}   # (two in flight only up to 65536 reads per call)
echo "[" > $O/r05_batchsize.json; first=1
for n in $SIZES; do
  steps=$(( 262144 / n )); [ $steps -lt 3 ] && steps=3; [ $steps -gt 16 ] && steps=16
  for fl in 1 2; do [ $fl = 2 ] && [ $n -gt 65536 ] && continue
    GPU_MAX_HW_QUEUES=8 RH_BENCH_IN_FLIGHT=$fl timeout 900 python bench.py --reads $n --steps $steps --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > $O/bsz.json
    python - <<PY
import json
d=json.load(open("$O/bsz.json"))
e={"reads_per_call": $n, "in_flight": $fl, "steps": $steps, "value_resident": d["value"], "value_h2d_included": d["value_h2d_included"], "ms_per_step": d["ms_per_step"], "ms_per_step_h2d_included": d["ms_per_step_h2d_included"]}
print(json.dumps(e))
open("$O/r05_batchsize.json","a").write(("" if $first else ",\n") + json.dumps(e))
PY
    first=0
  done
done
echo "]" >> $O/r05_batchsize.json

# sweep an environment variable over values with the 3-stream / 1-stream bench.  Usage: bash tools/r03_sweep.sh VAR v1 v2 ...
cd /root/repo; V=$1; shift
for x in "$@"; do
  a=$(env $V=$x timeout 600 python bench.py --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(env $V=$x RH_SUB_BATCHES=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$V=$x  3-stream: $a   1-stream: $b" | tee -a gpurun_out/sweep.log
done

# sweep an environment variable with the default (multi-stream) bench only.  Usage: bash tools/r03_sweep3.sh VAR v1 v2 ...
cd /root/repo; V=$1; shift
for x in "$@"; do
  a=$(env $V=$x timeout 600 python bench.py --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$V=$x  bench: $a" | tee -a gpurun_out/sweep.log
done

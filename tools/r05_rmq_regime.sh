# is the RMQ walk latency bound or issue bound?  step time against reads per call on ONE stream (D. mel scale: 10 chunk launches of ~35 k anchors per read).  Usage: bash tools/r05_rmq_regime.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in 1000 2000 4000 8000 16000; do
  RH_SUB_BATCHES=1 timeout -k 10 400 python bench.py --workload dmel --reads $n --mapopt rmq --steps 1 --warmup 1 --pool 2 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r05_regime_$n.json
  python -c "import json;d=json.load(open('$O/r05_regime_$n.json'));print($n, d['value'], d['ms_per_step'])"
done

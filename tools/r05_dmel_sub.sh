# D. mel scale against sub-batch streams x hardware queues.  Usage: bash tools/r05_dmel_sub.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for cfg in "4 2" "4 3" "8 3" "8 4" "8 6"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 RH_SUB_BATCHES=$2 timeout -k 10 400 python bench.py --workload dmel --steps 3 --warmup 1 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r05_dmel_q$1_s$2.json
  python -c "import json;d=json.load(open('$O/r05_dmel_q$1_s$2.json'));print('queues $1 streams $2', d['value'], d['ms_per_step'])"
done

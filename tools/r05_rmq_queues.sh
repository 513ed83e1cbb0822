# --rmq at D. mel scale: sub-batch streams x hardware queues (is the third stream's loss a shared hardware queue?).  Usage: bash tools/r05_rmq_queues.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for cfg in "4 2" "4 3" "8 2" "8 3" "8 4"; do
  set -- $cfg
  for n in 8000 48000; do
    GPU_MAX_HW_QUEUES=$1 RH_SUB_BATCHES=$2 timeout -k 10 400 python bench.py --workload dmel --reads $n --mapopt rmq --steps 1 --warmup 1 --pool 2 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r05_rmqq_$1_$2_$n.json
    python -c "import json;d=json.load(open('$O/r05_rmqq_$1_$2_$n.json'));print('queues $1 streams $2 reads $n', d['value'], d['ms_per_step'])"
  done
done

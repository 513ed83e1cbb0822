# E. coli scale: hardware queues x sub-batch streams (and the round's two new code paths).  Usage: bash tools/r05_ecoli_ab.sh
cd /root/repo
run() { env "$@" timeout 300 python bench.py --workload ecoli --steps 5 --warmup 2 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run RH_SUB_BATCHES=1
run RH_SUB_BATCHES=2
run RH_SUB_BATCHES=4
run RH_SUB_BATCHES=6
run GPU_MAX_HW_QUEUES=4 RH_SUB_BATCHES=2
run GPU_MAX_HW_QUEUES=2 RH_SUB_BATCHES=3

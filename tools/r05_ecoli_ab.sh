cd /root/repo
run() { env "$@" timeout 300 python bench.py --workload ecoli --steps 5 --warmup 2 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run A=1
run GPU_MAX_HW_QUEUES=4
run RH_BT_WAVE=1
run GPU_MAX_HW_QUEUES=4 RH_BT_WAVE=1
run RH_BS_PW=0

# --rmq / --bw-long against the number of sub-batch streams of a call.  Usage: bash tools/r05_rmq_sub.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() {  # tag workload reads mapopt sub
  RH_SUB_BATCHES=$5 timeout -k 10 500 python bench.py --workload $2 --reads $3 --mapopt $4 --steps 1 --warmup 1 --pool 2 --cpu-sample 0 --no-h2d 2>/dev/null | tail -1 > $O/r05_sub_$1.json
  python -c "import json;d=json.load(open('$O/r05_sub_$1.json'));print('$1', d['value'], d['ms_per_step'])"
}
for sb in 1 2 3; do
  run ecoli_rmq_s$sb ecoli 20000 rmq $sb
  run ecoli_bwl_s$sb ecoli 20000 bw_long $sb
  run dmel_rmq_s$sb dmel 8000 rmq $sb
  run dmel_rmq48k_s$sb dmel 48000 rmq $sb
done

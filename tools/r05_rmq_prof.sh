# where the RMQ chaining time goes: per-kernel stats of an E. coli-scale and a D. mel-scale --rmq step, one stream.  Usage: bash tools/r05_rmq_prof.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() {  # workload reads
  rm -rf /tmp/rp_$1
  RH_SUB_BATCHES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$1 -o p -- python bench.py --workload $1 --reads $2 --mapopt rmq --steps 1 --warmup 1 --pool 2 --cpu-sample 0 --no-h2d > $O/r05_$1_rmq_1s.out 2>$O/r05_$1_rmq_1s.err
  f=$(find /tmp/rp_$1 -name "*kernel_stats.csv" | head -1)
  head -12 $f | cut -c1-240 > $O/r05_$1_rmq_kernel_stats_1stream.csv
  tail -1 $O/r05_$1_rmq_1s.out | cut -c1-260; head -6 $O/r05_$1_rmq_kernel_stats_1stream.csv
}
run ecoli 20000; run dmel 8000

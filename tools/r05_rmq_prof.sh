# where the RMQ chaining time of the D. mel-scale run goes: per-kernel stats.  Usage: bash tools/r05_rmq_prof.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
n=8000
rm -rf /tmp/rp$n
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$n -o p -- python bench.py --workload dmel --reads $n --mapopt rmq --steps 1 --warmup 1 --cpu-sample 0 --no-h2d > $O/r05_dmel_rmq_$n.out 2>$O/r05_dmel_rmq_$n.err
f=$(find /tmp/rp$n -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-240 > $O/r05_dmel_rmq_${n}_kernel_stats.csv
tail -1 $O/r05_dmel_rmq_$n.out | cut -c1-400
cat $O/r05_dmel_rmq_${n}_kernel_stats.csv

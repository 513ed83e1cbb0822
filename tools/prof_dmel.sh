cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
RH_SUB_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o p -- python $R/bench.py --genome 144000000 --reads 24000 --steps 1 --warmup 0 --cpu-sample 0 >/dev/null 2>&1
cp $(find /tmp/prof1 -name "*kernel_stats.csv" | head -1) $O/r02_dmel_kernel_stats_a.csv
head -30 $O/r02_dmel_kernel_stats_a.csv

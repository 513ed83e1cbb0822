cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
READS=${1:-2048}; SUBS=${2:-1}
RH_SUB_BATCHES=$SUBS timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profh -o p -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --cpu-sample 0 > $O/prof_human_bench_${READS}_${SUBS}.json 2>/dev/null
cp $(find /tmp/profh -name "*kernel_stats.csv" | head -1) $O/r02_human_kernel_stats_${READS}_${SUBS}.csv
head -${3:-36} $O/r02_human_kernel_stats_${READS}_${SUBS}.csv | cut -c1-200

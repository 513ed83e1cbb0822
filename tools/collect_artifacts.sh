cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd $R && timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/r01_final_bench.json
RH_SUB_BATCHES=1 timeout 300 python bench.py --cpu-sample 0 2>/dev/null | tail -1 > $O/r01_final_bench_1stream.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 >/dev/null 2>&1
cp $(find /tmp/prof3 -name "*kernel_stats.csv" | head -1) $O/r01_final_kernel_stats.csv
RH_SUB_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 >/dev/null 2>&1
cp $(find /tmp/prof1 -name "*kernel_stats.csv" | head -1) $O/r01_final_kernel_stats_1stream.csv
timeout 900 python $R/profiles/collect_pmc.py > $O/pmc.log 2>&1; cp $R/profiles/pmc_traffic.json $O/
cd $R && timeout 300 python bench.py --cpu-sample 0 2>/dev/null | tail -1 > $O/r01_final_bench_nocpu.json
head -c 600 $O/r01_final_bench.json; echo; head -5 $O/r01_final_kernel_stats_1stream.csv

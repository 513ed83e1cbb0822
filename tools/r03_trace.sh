# per-round + per-sorter-level trace of one 1-stream step.  Usage: bash tools/r03_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-tr}; mkdir -p $O
cd $R
RH_SUB_BATCHES=1 RH_TRACE_ROUNDS=1 RH_BS_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-h2d 2>$O/${TAG}_trace.err | tail -1 > $O/${TAG}_trace.json
grep -c "BS level" $O/${TAG}_trace.err

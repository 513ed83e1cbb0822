# per-level trace of the multi-workgroup sorter on one human-scale step, one stream (needs a -DRH_DEV build).  Usage: bash tools/r05_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-trace}; mkdir -p $O; cd $R
RH_SUB_BATCHES=1 RH_BS_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > $O/${TAG}_line.json 2> $O/${TAG}_bs_trace.log
grep -c "BS level" $O/${TAG}_bs_trace.log

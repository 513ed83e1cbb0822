# round trace of a 131 072-read call (needs a -DRH_DEV build).  Usage: bash tools/r04_trace.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
RH_TRACE_ROUNDS=1 timeout 900 python bench.py --reads 131072 --steps 1 --warmup 1 --cpu-sample 0 --no-h2d 2>$O/trace131.err | tail -1 > $O/trace131.json
grep "round" $O/trace131.err | tail -80

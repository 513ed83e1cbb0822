#!/bin/bash
# development aid: cost of the gather in k_sort_block's write-out (RH_SORT_DBG_WO=1 replaces it by a straight copy: results invalid)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for v in 0 1; do
  if [ $v = 1 ]; then export RH_SORT_DBG_WO=1; fi
  for cfg in "40000 2800" "200000 400"; do
  rm -rf /tmp/sm_$v
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm_$v -o sm -- python $R/tests/r03_sortmicro.py $cfg > $R/gpurun_out/sm_$v.log 2>&1 < /dev/null
  grep -h "rep 2\|sorted" $R/gpurun_out/sm_$v.log
  f=$(find /tmp/sm_$v -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then echo "== dbg_wo=$v cfg=$cfg"; head -4 "$f" | cut -c1-160; fi
  done
done

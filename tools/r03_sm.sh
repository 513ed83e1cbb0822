#!/bin/bash
# development aid: k_sort_block on free-standing segments without / with ties, kernel times from rocprofv3
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for cfg in "40000 2800 27" "40000 2800 16" "200000 400 27"; do
  rm -rf /tmp/sm
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm -o sm -- python $R/tools/r03_sortmicro.py $cfg > $R/gpurun_out/sm.log 2>&1 < /dev/null
  grep -h "sorted" $R/gpurun_out/sm.log
  f=$(find /tmp/sm -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then echo "== cfg=$cfg"; head -4 "$f" | cut -c1-140; fi
done

#!/bin/bash
# development aid: two regions per lane in the token walker (RH_BS_TOK2) against the default
for v in 0 1; do
  if [ $v = 1 ]; then export RH_BS_TOK2=1; fi
  RH_SUB_BATCHES=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d > gpurun_out/tok2_1s_$v.json 2> gpurun_out/tok2_$v.err < /dev/null
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 2000 --no-h2d > gpurun_out/tok2_3s_$v.json 2>> gpurun_out/tok2_$v.err < /dev/null
  python - <<PY
import json
for f in ('gpurun_out/tok2_1s_$v.json','gpurun_out/tok2_3s_$v.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print($v, f, d['value'], d['ms_per_step'], d.get('paf_sample_identical'), {k:round(x,1) for k,x in d['stage_ms_per_step'].items() if x>20})
PY
done

#!/bin/bash
# development aid: four regions per lane (RH_BS_TOK4) for ranges with 129 .. 256 regions against the default
for v in 1; do
  export RH_BS_TOK4=1
  RH_SUB_BATCHES=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d > gpurun_out/tok4_1s.json 2> gpurun_out/tok4.err < /dev/null
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 2000 --no-h2d > gpurun_out/tok4_3s.json 2>> gpurun_out/tok4.err < /dev/null
  python - <<PY
import json
for f in ('gpurun_out/tok4_1s.json','gpurun_out/tok4_3s.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('paf_sample_identical'), {k:round(x,1) for k,x in d['stage_ms_per_step'].items() if x>20})
PY
done

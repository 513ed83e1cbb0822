#!/bin/bash
# development aid: sort parity + one-stream / three-stream bench lines of the current build
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sort or regions" 2>&1 | tail -2
RH_SUB_BATCHES=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2d > gpurun_out/tok3_1s.json 2> gpurun_out/tok3.err < /dev/null
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 2000 --no-h2d > gpurun_out/tok3_3s.json 2>> gpurun_out/tok3.err < /dev/null
python - <<PY
import json
for f in ('gpurun_out/tok3_1s.json','gpurun_out/tok3_3s.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('paf_sample_identical'), {k:round(x,1) for k,x in d['stage_ms_per_step'].items() if x>20})
PY

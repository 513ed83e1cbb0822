# the round-end checks as the driver runs them: GPU suite, smoke, default bench line.  Usage: bash tools/r05_check.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/r05_check_pytest.log 2>&1; grep -E "passed|failed|error" $O/r05_check_pytest.log | tail -3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python bench.py > $O/r05_check_bench.out 2>$O/r05_check_bench.err; tail -1 $O/r05_check_bench.out > $O/r05_check_bench.json
python - <<PY
import json
d=json.loads(open("$O/r05_check_bench.json").read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("paf_sample_identical"))
PY

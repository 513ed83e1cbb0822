"""Device time per kernel AND stage of one human-scale step on one stream (rocprofv3 --kernel-trace, dispatch order = program order):

    cd /tmp && export TMPDIR=/tmp && python /root/repo/profiles/collect_stage_kernels.py [out.json]

The segment sorters (k_sort_block, k_bs_*) serve four stages; a dispatch belongs to the stage whose producer kernel precedes it
(same rule as collect_pmc.py).  Output: ms per step for every kernel@stage, largest first.
"""
import csv
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_pmc import ROOT, WORKLOAD, READS, SORT_OWNER, NOT_PATH, stage_of  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "stage_kernels.json")
    d = "/tmp/stage_kernels"
    env = dict(os.environ, RH_SUB_BATCHES="1")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--workload", WORKLOAD, "--reads", str(READS), "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-h2d"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, env=env, timeout=900)
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    acc, owner = {}, "sort"
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(NOT_PATH):
            continue
        for name_, ow_ in SORT_OWNER.items():
            if k == name_ or k.startswith(name_ + "<"):
                owner = ow_
        key = k + "@" + owner if k.startswith(("k_sort", "k_bs_")) else k + "@" + (stage_of(k) or "?")
        e = acc.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    stages = {}
    for k, (n, ms) in acc.items():
        st = k.split("@")[1]
        stages[st] = stages.get(st, 0.0) + ms
    res = {"workload": WORKLOAD, "reads": READS, "streams": 1, "steps_in_trace": 1,
           "stage_ms": dict(sorted(((s, round(v, 1)) for s, v in stages.items()), key=lambda x: -x[1])),
           "kernel_ms": {k: {"launches": n, "ms": round(ms, 2)} for k, (n, ms) in sorted(acc.items(), key=lambda x: -x[1][1])}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["stage_ms"]))
    for k, v in list(res["kernel_ms"].items())[:45]:
        print(f"{k:60s} {v['launches']:6d} {v['ms']:9.1f}")


if __name__ == "__main__":
    main()

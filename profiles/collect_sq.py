"""Issue / wait breakdown per kernel with rocprofv3 SQ counters (one pass, kernel-trace only, one stream):

    cd /tmp && export TMPDIR=/tmp && python /root/repo/profiles/collect_sq.py

SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing), all in
quad-cycles summed over waves (MI355X_MICROARCH.md).  Output: profiles/sq_counters.json + a table on stdout.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]
READS = os.environ.get("RH_PMC_READS", "65536")


def main():
    out = "/tmp/pmc_sq"
    env = dict(os.environ, RH_SUB_BATCHES="1")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--reads", READS, "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-h2d"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        e = acc.setdefault(k, {"launches": set()})
        e["launches"].add(r["Dispatch_Id"])
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    res = {}
    for k, e in acc.items():
        e["launches"] = len(e["launches"])
        wc = e.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            e["frac_parked"] = e.get("SQ_WAIT_ANY", 0.0) / wc
            e["frac_issue_stall"] = e.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["frac_issuing"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        res[k] = e
    with open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w") as fo:
        json.dump({"reads": int(READS), "note": "one stream (RH_SUB_BATCHES=1); quad-cycles summed over waves", "kernels": res}, fo, indent=1)
    print(f"{'kernel':40s} {'launches':>8s} {'wave Gcyc':>10s} {'parked':>7s} {'stall':>7s} {'issuing':>8s} {'valu':>6s} {'lds':>6s}")
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:24]:
        wc = e.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        print(f"{k[:40]:40s} {e['launches']:8d} {4 * wc / 1e9:10.2f} {e.get('frac_parked', 0):7.2f} {e.get('frac_issue_stall', 0):7.2f} {e.get('frac_issuing', 0):8.2f} "
              f"{e.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f} {e.get('SQ_ACTIVE_INST_LDS', 0) / wc:6.2f}")


if __name__ == "__main__":
    main()

"""Collect HBM traffic per kernel with rocprofv3 PMC counters (separate passes, no tracing domains besides kernel-trace):

    cd /tmp && export TMPDIR=/tmp && python /root/repo/profiles/collect_pmc.py

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE x 1024.  k_prefilter is the calibration point: it streams
exactly 2 B x raw samples.  Output: profiles/pmc_traffic.json (bytes per launch for each bench stage).
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = os.environ.get("RH_PMC_WORKLOAD", "human")
READS = int(os.environ.get("RH_PMC_READS", "65536"))
SAMPLES, JUNK = 40000, 102

# every kernel of the mapping path belongs to a stage of bench.py's table (prefix match on the kernel name)
STAGE_OF = [("k_prefilter", "prefilter"), ("k_events_norm", "events_norm"), ("k_events_tstat", "events_norm"), ("k_events_peaks", "events_peaks"), ("k_events_means", "events_means"),
            ("k_sketch", "sketch"), ("k_probe", "probe"), ("k_scan_anchors", "scan"), ("k_rebase_offsets", "scan"), ("k_expand", "expand"), ("k_chain_wave", "chain"), ("k_chain_serial", "chain"), ("k_chain_rmq", "chain"), ("k_events_append", "events_means"), ("k_regions_dtw", "regions"), ("k_dtw_", "regions"),
            ("k_zbuild", "zsort"), ("k_backtrack_spec", "backtrack"), ("k_chain_reorder", "backtrack"), ("k_chain_keys", "backtrack"), ("k_need", "prefilter"), ("k_fetch", "prefilter"),
            ("k_regions_prep", "rsort"), ("k_regions", "regions"), ("k_carry_", "compact"), ("k_compact_active", "compact"), ("k_finalize", "finalize")]
NOT_PATH = ("k_synth_reads", "k_ix_", "__amd_rocclr")           # bench set-up (read generator, index construction, runtime copies)

# the segment sorters (k_sort_block, k_bs_*) serve four stages; which one a dispatch belongs to follows from the kernel that
# precedes the group of sort launches (single-stream run, so dispatch order = program order)
SORT_OWNER = {"k_expand": "sort", "k_zbuild": "zsort", "k_backtrack_spec": "backtrack", "k_regions_prep": "rsort"}


def stage_of(k):
    if "@" in k:
        return k.split("@")[1]
    for pre, st in STAGE_OF:
        if k.startswith(pre):
            return st
    return None


def one_pass(counter, out):
    env = dict(os.environ, RH_SUB_BATCHES="1")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counter.split() + ["--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--workload", WORKLOAD, "--reads", str(READS), "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-h2d"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    want = counter.split()
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] in want]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    if len(want) > 1:       # several counters in one pass: {counter: acc}
        return {c: _accumulate([r for r in rows if r["Counter_Name"] == c]) for c in want}
    return _accumulate(rows)


def _accumulate(rows):
    acc, owner = {}, "sort"
    seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for name_, ow_ in SORT_OWNER.items():
            if k == name_ or k.startswith(name_ + "<"):
                owner = ow_
        key = k + "@" + owner if k.startswith(("k_sort", "k_bs_")) else k
        e = acc.setdefault(key, [0, 0.0])
        if r["Dispatch_Id"] not in seen:
            e[0] += 1
            seen.add(r["Dispatch_Id"])
        e[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = one_pass("FETCH_SIZE", "/tmp/pmc_fetch")
    write = one_pass("WRITE_SIZE", "/tmp/pmc_write")
    # What did a read request really fetch?  FETCH_SIZE tallies every request at 64 bytes; the request-size counters say how many were 32 / 64 / 128-byte
    # requests (round 6, after tools/probes/gather_calib.hip: on gfx950 streams AND 8- / 16-byte gathers are all 128-byte line requests -
    # profiles/r06_gather_calib.txt - so the factor 2 of the guide holds for every access class measured).  Per kernel: read bytes = 32 n32 + 64 n64 + 128 n128,
    # fetch_factor = that / FETCH_SIZE.  RH_PMC_REQ_SIZES=0 skips the two extra passes and applies the factor 2 throughout.
    req, siz = {}, {}
    if os.environ.get("RH_PMC_REQ_SIZES", "1") != "0":
        try:
            req = one_pass("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum", "/tmp/pmc_req")
            siz = one_pass("TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum", "/tmp/pmc_siz")
        except Exception as ex:  # noqa: BLE001
            print("request-size counters not collected:", ex)
            req, siz = {}, {}
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, write.get(k))[0]
        fs = fetch.get(k, [0, 0.0])[1] * 1024.0
        rd, factor = 2.0 * fs, 2.0
        if req and siz and k in req.get("TCC_EA0_RDREQ_sum", {}):
            n_all = req["TCC_EA0_RDREQ_sum"][k][1]; n32 = req["TCC_EA0_RDREQ_32B_sum"].get(k, [0, 0.0])[1]
            n64 = siz["TCC_EA0_RDREQ_64B_sum"].get(k, [0, 0.0])[1]; n128 = siz["TCC_EA0_RDREQ_128B_sum"].get(k, [0, 0.0])[1]
            if n_all > 0 and abs((n32 + n64 + n128) - n_all) <= 0.05 * n_all + 64:      # (separate passes of the same command: counts agree to a fraction of a percent)
                rd = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
                factor = rd / fs if fs else 2.0
        wr = write.get(k, [0, 0.0])[1] * 1024.0
        kernels[k] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": (rd + wr) / max(n, 1), "fetch_factor": round(factor, 4)}
    stages, unattributed, setup = {}, [], {"bytes": 0.0}
    for k, v in kernels.items():
        if k.startswith(NOT_PATH):
            setup["bytes"] += v["read_bytes"] + v["write_bytes"]
            continue
        st = stage_of(k)
        if st is None:
            unattributed.append(k)
            continue
        e = stages.setdefault(st, {"launches": 0, "bytes": 0.0})
        e["launches"] += v["launches"]
        e["bytes"] += v["read_bytes"] + v["write_bytes"]
    for e in stages.values():
        e["bytes_per_step"] = e["bytes"]          # the profiled command runs exactly one step
    out = {"workload": WORKLOAD, "reads": READS, "samples": SAMPLES, "junk": JUNK, "commit": os.environ.get("RH_COMMIT"), "round": 6, "note": "read bytes = 32 n32 + 64 n64 + 128 n128 from the request-size counters (fetch_factor = that / FETCH_SIZE; 2.0 where they were not collected), write bytes = WRITE_SIZE KiB; one stream (RH_SUB_BATCHES=1)",
           "path_bytes_per_step": sum(e["bytes"] for e in stages.values()), "unattributed_kernels": unattributed, "setup_bytes": setup["bytes"],
           "kernels": kernels, "stages": stages}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    pf = kernels.get("k_prefilter")
    if pf:
        print("calibration: k_prefilter read bytes", pf["read_bytes"], "expected", 2 * READS * SAMPLES)
    print("path total GB/step", out["path_bytes_per_step"] / 1e9, "unattributed", unattributed)
    for st, e in sorted(stages.items(), key=lambda kv: -kv[1]["bytes"]):
        print(f"{st:14s} {e['bytes'] / 1e9:9.2f} GB  {e['launches']} launches")
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["read_bytes"] - kv[1]["write_bytes"])[:14]:
        print(f"{k:44s} launches {v['launches']:5d}  read {v['read_bytes']/1e9:8.3f} GB  write {v['write_bytes']/1e9:8.3f} GB  fetch factor {v['fetch_factor']:.3f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One script for the bench sweeps behind profiles/*_batchsize.json, *_queues.json, *_rmq_*.json ...: the cartesian product of environment settings and
bench.py options, one bench line per point, the interesting fields of every line collected in gpurun_out/<out>.json.  Runs on the GPU box through gpurun:

    tools/gpu_retry.sh 1500 /tmp/log python tools/sweep.py --out r06_batchsize --env GPU_MAX_HW_QUEUES=8 --env RH_BENCH_IN_FLIGHT=1,2 \\
        --opt reads=8192,12500,16384,65536,131072,262144 --steps-for-reads 262144 -- --warmup 1 --cpu-sample 0
    python tools/sweep.py --out r06_rmq_calls --opt reads=8000,24000,48000 -- --workload dmel --mapopt rmq --steps 1 --warmup 1 --pool 2 --cpu-sample 0 --no-h2d
    python tools/sweep.py --out r06_dmel_streams --env GPU_MAX_HW_QUEUES=4,8 --env RH_SUB_BATCHES=2,3,4 -- --workload dmel --steps 3 --warmup 1 --cpu-sample 0 --no-h2d

--env NAME=v1,v2 and --opt name=v1,v2 (-> `--name v`) may be repeated; --steps-for-reads T picks --steps = clamp(T / reads, 3, 16) per point.
"""
import argparse, itertools, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("value", "ms_per_step", "value_h2d_included", "ms_per_step_h2d_included", "value_h2d_full_upload", "paf_sample_identical")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--steps-for-reads", type=int, default=0)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = [x for x in a.rest if x != "--"]
    axes = [("env", *e.split("=", 1)) for e in a.env] + [("opt", *o.split("=", 1)) for o in a.opt]
    points = list(itertools.product(*[[(kind, name, v) for v in vals.split(",")] for kind, name, vals in axes])) or [()]
    out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
    results = []
    for pt in points:
        env = dict(os.environ); opts = []; label = {}
        for kind, name, v in pt:
            label[name] = v
            if kind == "env": env[name] = v
            else: opts += ["--" + name, v]
        if a.steps_for_reads and "reads" in label:
            opts += ["--steps", str(max(3, min(16, a.steps_for_reads // int(label["reads"]))))]
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + opts + rest
        try:
            p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=a.timeout, text=True)
            d = json.loads(p.stdout.strip().splitlines()[-1])
            e = dict(label, **{k: d.get(k) for k in KEEP if k in d})
            cb = d.get("cpu_baseline") or {}
            if cb: e["cpu_value"] = cb.get("value"); e["cpu_cores"] = cb.get("cores")
        except Exception as ex:  # noqa: BLE001
            e = dict(label, error=str(ex)[:200])
        print(json.dumps(e), flush=True)
        results.append(e)
        json.dump(results, open(os.path.join(out_dir, a.out + ".json"), "w"), indent=1)


if __name__ == "__main__":
    main()

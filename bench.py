#!/usr/bin/env python3
"""Benchmark of the mapping hot path (BASELINE.json metric: reads/s mapped + raw-signal Gsamples/s).

    python bench.py --gpus N --steps K --warmup W

Workload (config[1] of BASELINE.json): E. coli-sized synthetic genome (4.6 Mbp, preset `sensitive`), synthetic R9.4
reads of 40 000 raw samples (= max_num_chunk x chunk_size), 100 000 reads PER GPU (weak scaling), 10 % unmappable.
A "step" = one pass of the whole hot path (rh_map_batch: prefilter .. finalize, all chunk rounds) over the rank's
batch, with the int16 samples already resident in HBM (generated there) and the index resident in HBM.
For N > 1 launch with torch.distributed.run (one rank per GPU): rank 0 builds + uploads the index and the flattened
device blob is broadcast to the other GPUs over RCCL; reads are sharded, there is no collective on the data path.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the largest share of device time: algorithmic bytes
per launch (DESIGN.md section "Algorithmic bytes") / its average launch duration measured with HIP events on the
context's stream.  `cpu_baseline` (N = 1 only) = the CPU oracle (oracle/rh_oracle.c, a port of the reference path,
bit-identical to it on the goldens) on all host cores over a bounded sample of the same reads.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic bytes per stage as a function of the step's counters; S = samples, Ns seeds, Nh hits,
# Na anchors (incl. carried), Nc chained anchors, Ne events
ALGO_BYTES = {
    "prefilter": lambda c: 2 * c["n_samples_raw"],
    "events_norm": lambda c: 2 * c["n_samples_used"],                      # int16 in; z/t1/t2 rows are intermediates
    "events_peaks": lambda c: 8 * c["n_samples_used"] + 2 * c["n_events"],  # interface of the stage: t1,t2 in, peaks out
    "events_means": lambda c: 4 * c["n_samples_used"] + 6 * c["n_events"],  # z + peaks in, events out
    "sketch": lambda c: 4 * c["n_events"] + 16 * c["n_seeds"],
    "probe": lambda c: 16 * c["n_seeds"] + 16 * c["n_seeds"],
    "expand": lambda c: 8 * c["n_hits"] + 16 * c["n_anchors"],
    "sort": lambda c: 32 * c["n_anchors"],
    "chain": lambda c: 16 * c["n_anchors"] + 12 * c["n_anchors"],           # anchors in, f/p/v out
    "zsort": lambda c: 4 * c["n_anchors"] + 32 * c["n_chained"],            # f in; candidates >= chained anchors
    "backtrack": lambda c: 8 * c["n_anchors"] + 32 * c["n_chained"],        # f,p in; chained anchors out (+ carried copy)
    "rsort": lambda c: 16 * c["n_chained"],
    "regions": lambda c: 16 * c["n_chained"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU")
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--samples", type=int, default=40_000)
    ap.add_argument("--junk", type=int, default=102, help="unmappable reads per 1024")
    ap.add_argument("--preset", default="sensitive")
    ap.add_argument("--cpu-sample", type=int, default=40000, help="reads of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from rawhash_amd import Context, Index, MapOptions, SynthWorkload
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RH_BENCH_BACKEND=gloo RH_BENCH_ONE_DEVICE=1 lets the multi-rank flow be exercised on a single-GPU box (tests only)
        backend = os.environ.get("RH_BENCH_BACKEND", "nccl")
        if os.environ.get("RH_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)

    wl = SynthWorkload(chrom_len=args.genome, n_chrom=1, n_samples=args.samples, junk_per_1024=args.junk)
    opts = MapOptions(args.preset)
    port = os.environ.get("MASTER_PORT", "0")
    workdir = os.path.join(tempfile.gettempdir(), f"rawhash_amd_bench_{port}_{os.getppid() if world > 1 else os.getpid()}")
    fasta, model = os.path.join(workdir, "ref.fa"), os.path.join(workdir, "model.txt")
    ind = os.path.join(workdir, "ref.ind")
    ctx = Context(local_rank)
    t_setup = time.time()
    index = None
    if rank == 0:
        wl.write_reference(workdir)
        index = Index.build(fasta, model, opts, out_ind=ind if args.cpu_sample and world == 1 else None, n_threads=os.cpu_count() or 8)
        opts.update(index)
        ctx.upload(index)
    keep = []
    if world > 1:
        # replicate the HBM-resident index: one RCCL broadcast of the flattened blob over xGMI (the only collective)
        from rawhash_amd.dist import replicate_index
        keep.append(replicate_index(ctx, opts, None, device=f"cuda:{local_rank}"))
        dist.barrier()
    # this rank's shard of the read set, generated straight into HBM
    batch = wl.reads_device(ctx, model, rank * args.reads, args.reads)
    t_setup = time.time() - t_setup

    def sync():
        if world > 1:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.map_batch(opts, batch)
    acc = {}
    stage_ms, stage_n = {}, {}
    n_mapped = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs = ctx.map_batch(opts, batch)      # synchronous: returns after the records are back on the host
        st = ctx.stats()
        for k, v in st.items():
            if k not in ("stages", "ms_total"):
                acc[k] = acc.get(k, 0) + v
        for k, (ms, n) in st["stages"].items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ms
            stage_n[k] = stage_n.get(k, 0) + n
        n_mapped = int(recs["mapped"].sum())
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        acc["n_samples_raw"] = args.reads * args.samples * args.steps
        total_reads = args.reads * world * args.steps
        value = total_reads / elapsed
        kernels = {k: v for k, v in stage_ms.items() if k in ALGO_BYTES and stage_n.get(k)}
        dom = max(kernels, key=kernels.get)
        dom_bytes = ALGO_BYTES[dom](acc)
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        path_bytes = 2 * acc["n_samples_used"] + 16 * acc["n_seeds"] + 8 * acc["n_hits"] + 32 * acc["n_anchors"] + 16 * acc["n_chained"] + 64 * acc["n_reads"]
        scale = "E. coli" if args.genome <= 10_000_000 else "D. melanogaster" if args.genome <= 200_000_000 else "human"
        dev_ms = sum(kernels.values())
        out = {
            "metric": f"reads/sec mapped ({scale}-scale index resident in HBM)", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16 signal; fp32/fp64 events; u64/i32 seeding+chaining", "data": "synthetic",
            "config": {"workload": f"{scale}-sized synthetic genome {args.genome} bp + {args.reads} synthetic R9.4 reads/GPU x {args.samples} samples, preset {args.preset}, "
                                   f"{args.junk}/1024 unmappable reads, index + int16 signal resident in HBM",
                       "reads_per_gpu": args.reads, "samples_per_read": args.samples, "mid_occ": int(opts.mo.mid_occ), "parallelism": f"reads sharded x{world}, index replicated"},
            "gsamples_per_s_consumed": round(acc["n_samples_used"] * world / elapsed / 1e9, 4),
            "gsamples_per_s_input": round(args.reads * args.samples * world * args.steps / elapsed / 1e9, 4),
            "mapped_fraction": round(n_mapped / args.reads, 4),
            "chunks_per_read": round(acc["n_chunks"] / acc["n_reads"], 3),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(dom, args, stage_n[dom] / args.steps),
                         "avg_launch_ms": round(stage_ms[dom] / stage_n[dom], 4), "launches": stage_n[dom],
                         "algorithmic_bytes_per_launch": int(dom_bytes / stage_n[dom]),
                         "concurrent_streams": int(os.environ.get("RH_SUB_BATCHES", "3")),
                         "note": "launch durations are HIP-event times on each sub-batch's own stream; with >1 concurrent streams a launch shares the "
                                 "chip with the other streams' kernels, so frac understates the kernel alone (RH_SUB_BATCHES=1: profiles/r01_final_bench_1stream.json)"},
            "path": {"algorithmic_GB_per_step": round(path_bytes / args.steps / 1e9, 4), "device_ms_per_step": round(dev_ms / args.steps, 3),   # sum over concurrent sub-batch streams
                    
                     "achieved_GBs": round(path_bytes / elapsed / 1e9, 3)},
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items() if stage_n.get(k)},
            "setup_s": round(t_setup, 2),
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(wl, model, ind, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def pmc_traffic(stage, args, launches_per_step):
    """HBM bytes per launch of the dominant stage from the committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_traffic.json, made by profiles/collect_pmc.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    if d.get("reads") != args.reads or d.get("samples") != args.samples or d.get("junk") != args.junk:
        return None
    e = d.get("stages", {}).get(stage)
    return None if e is None or "bytes_per_step" not in e else int(e["bytes_per_step"] / max(launches_per_step, 1))


def cpu_baseline(wl, model, ind, args):
    """CPU baseline on every host core over a bounded sample of the same reads.

    kind "reference": the unmodified RawHash2 sources (oracle/_ref/ref_harness, prebuilt where /root/reference exists; its
    `map` command runs the reference's own kt_for(map_worker_for) and reports the map-phase time, file loading excluded).
    kind "port": oracle/rh_oracle.c (bit-identical restatement), also reported when the reference binary is present."""
    import re
    import subprocess
    import oracle_lib as O
    cores = os.cpu_count() or 1
    n = min(args.cpu_sample, args.reads)
    reads = wl.reads(model, 0, n, n_threads=cores, with_names=True)
    oix = O.OracleIndex(ind)
    _, mo = O.preset(args.preset)
    O.lib().ro_mapopt_update(C.byref(mo), oix.h)
    b = reads.batch()
    t0 = time.perf_counter()
    recs = O.map_batch(oix, mo, b, n_threads=cores)
    dt = time.perf_counter() - t0
    port = {"value": round(n / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n} reads of the same synthetic set, {dt:.2f} s wall, oracle/rh_oracle.c with {cores} pthreads",
            "mapped_fraction": round(float(recs['mapped'].mean()), 4)}
    if not O.have_reference():
        return port
    try:
        n_ref = min(n, 20000)
        sub = reads.subset(range(n_ref))
        rhr = os.path.join(os.path.dirname(ind), "cpu_sample.rhr")
        sub.write(rhr, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
        p = subprocess.run([O.REF_HARNESS, "map", args.preset, ind, rhr, str(cores)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
        m = re.search(r"map phase ([0-9.]+) s", p.stderr)
        os.remove(rhr)
        if p.returncode != 0 or not m:
            return port
        t_ref = float(m.group(1))
        return {"value": round(n_ref / t_ref, 1), "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": f"first {n_ref} reads of the same synthetic set, map phase {t_ref:.2f} s (file loading excluded), unmodified RawHash2 "
                          f"sources built by oracle/Makefile (-O3 -ffp-contract=off), kt_for with {cores} threads",
                "port": port}
    except Exception:   # the baseline is a reported extra: never fail the bench because of it
        return port


if __name__ == "__main__":
    main()

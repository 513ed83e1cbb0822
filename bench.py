#!/usr/bin/env python3
"""Benchmark of the mapping hot path (BASELINE.json metric: reads/s mapped + raw-signal Gsamples/s vs a human index).

    python bench.py --gpus N --steps K --warmup W [--workload human|dmel|ecoli]

Workloads (BASELINE.json configs; all synthetic, seeded):
  human (default, the configuration the metric is quoted on): 24 chromosomes x 129 166 667 bp = 3.1 Gbp, preset `fast`
        (what the reference's own human runs use), reads of 40 000 raw samples (= max_num_chunk x chunk_size), 10 %
        unmappable; the index is built ON THE DEVICE (rh_index_build_device) and stays resident in HBM
  dmel  6 x 24 Mbp = 144 Mbp, preset `sensitive`
  ecoli 1 x 4.6 Mbp, preset `sensitive`
A "step" = one pass of the whole hot path (rh_map_batch: prefilter .. finalize, all chunk rounds) over the rank's batch of
reads, with the int16 samples already resident in HBM (generated there) and the index resident in HBM.  `value` = reads of
all ranks / time of the K timed steps (barrier + synchronize on both sides, max over ranks).
For N > 1 launch with torch.distributed.run (one rank per GPU): rank 0 builds the index, the flattened device blob is
broadcast to the other GPUs over RCCL (the only collective); reads are sharded (weak scaling: --reads per GPU).

Prints ONE JSON line (rank 0).  `roofline` is for the stage with the largest share of device time: algorithmic bytes per
launch (DESIGN.md "Algorithmic bytes") / its average launch duration measured with HIP events on the launch stream.
At human scale four stages (anchor sort, candidate sort, backtrack, region sort) take within a few percent of each other,
so stages within 10 % of the longest count as tied and the tie goes to the one that moves the most algorithmic bytes -
the line then names the same stage from run to run; `roofline_stages` lists every stage of the tie with its own fraction.
`cpu_baseline` (N = 1 only): the unmodified reference (oracle/_ref/ref_harness: its own kt_for(map_worker_for), its own
index loader reading the .ind this library wrote) on the host cores over a bounded sample of the same reads, best of a
thread sweep; the PAF it prints for the sample is compared with the HIP path's (`paf_sample_identical`).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MAX_CALL_READS = 262_144   # --reads beyond this: the set is generated and mapped shard by shard (bench_big_set)

WORKLOADS = {
    #          chrom_len, n_chrom, preset, reads/GPU, cpu sample, name
    "human": (129_166_667, 24, "fast", 65536, 20000, "human-scale (3.1 Gbp)"),
    "dmel": (24_000_000, 6, "sensitive", 50_000, 12_000, "D. melanogaster-scale (144 Mbp)"),
    "ecoli": (4_600_000, 1, "sensitive", 100_000, 40_000, "E. coli-scale (4.6 Mbp)"),
}

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


def dominant_stage(kernels, acc):
    """(stage the roofline object is about, [(stage, ms, algorithmic bytes)] of the stages within 10 % of the longest)."""
    top = max(kernels.values())
    tied = [(k, v, ALGO_BYTES[k](acc)) for k, v in kernels.items() if v >= 0.9 * top]
    return max(tied, key=lambda t: t[2])[0], tied


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="human", choices=sorted(WORKLOADS) + ["ava"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's default)")
    ap.add_argument("--samples", type=int, default=40_000)
    ap.add_argument("--junk", type=int, default=102, help="unmappable reads per 1024")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads of the CPU-baseline sample (0 = skip, -1 = the workload's default)")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false", help="skip the PCIe-inclusive measurement (batches in pinned host memory)")
    ap.add_argument("--dry-run", action="store_true", help="parse the launch (arguments + torch.distributed environment), print the plan as JSON and exit before touching a GPU")
    ap.add_argument("--mapopt", default="", choices=["", "rmq", "bw_long", "dtw"],
                    help="secondary lines for the chaining variants of SURVEY 8 f4: rmq = --rmq (mg_lchain_rmq), bw_long = --bw-long 2000 (RMQ re-chaining), "
                         "dtw = --dtw-evaluate-chains on a --store-sig index (built on the device)")
    ap.add_argument("--pool", type=int, default=4, help="distinct read batches kept resident and mapped in turn (a step never maps the batch of the step before)")
    ap.add_argument("--shard", type=int, default=65_536, help="--reads beyond what one call takes: reads per rh_map_batch call (the set is generated shard by shard)")
    ap.add_argument("--cpu-threads", default="", help="thread counts of the CPU sweep, comma separated (default: cores/8 .. cores)")
    args = ap.parse_args()
    if args.workload == "ava":
        return bench_ava(args)
    chrom_len, n_chrom, preset, d_reads, d_sample, wl_name = WORKLOADS[args.workload]
    if args.reads <= 0:
        args.reads = d_reads
    if args.cpu_sample < 0:
        args.cpu_sample = d_sample

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    if args.dry_run:
        print(json.dumps({"dry_run": True, "world": world, "rank": rank, "local_rank": local_rank, "gpus": args.gpus, "workload": args.workload,
                          "reads_per_gpu": args.reads, "first_read": rank * args.reads, "steps": args.steps, "warmup": args.warmup,
                          "backend": os.environ.get("RH_BENCH_BACKEND", "nccl"), "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}",
                          "cpu_baseline": world == 1 and args.cpu_sample > 0}))
        return
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

    cores = os.cpu_count() or 8
    wl = SynthWorkload(chrom_len=chrom_len, n_chrom=n_chrom, n_samples=args.samples, junk_per_1024=args.junk)
    opts = MapOptions(preset)
    ref_env = {}                                   # the same option for the reference harness (oracle/ref_harness.cpp reads it from the environment)
    if args.mapopt == "rmq":
        opts.mo.flag |= 0x2; ref_env = {"RH_RMQ": "1"}
    elif args.mapopt == "bw_long":
        opts.mo.bw_long = 2000; ref_env = {"RH_BW_LONG": "2000"}
    elif args.mapopt == "dtw":
        opts.io.flag |= 0x10; opts.mo.flag |= 0x40; ref_env = {"RH_STORE_SIG": "1", "RH_DTW": "1"}
    args.ref_env = ref_env
    port = os.environ.get("MASTER_PORT", "0")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    workdir = os.path.join(shm, f"rawhash_amd_bench_{port}_{os.getppid() if world > 1 else os.getpid()}")
    os.makedirs(workdir, exist_ok=True)
    model = os.path.join(workdir, "model.txt")
    ctx = Context(local_rank)
    t_setup = time.time()
    index = None
    t_index = 0.0
    if rank == 0:
        wl._l.rh_synth_write_model(C.byref(wl.cfg), model.encode())
        seqs = [wl.genome(c, n_threads=min(cores, 64)) for c in range(n_chrom)]
        t0 = time.time()
        index = Index.build_device_seqs(ctx, [f"chr{i + 1}" for i in range(n_chrom)], seqs, model, opts, n_threads=min(cores, 64))
        t_index = time.time() - t0
        del seqs
        opts.update(index)
    keep = []
    if world > 1:
        # replicate the HBM-resident index: one RCCL broadcast of the flattened blob over xGMI (the only collective)
        from rawhash_amd.dist import replicate_index
        dist.barrier()                                     # the model file is there
        keep.append(replicate_index(ctx, opts, None, device=f"cuda:{local_rank}"))
        dist.barrier()
    if args.reads > MAX_CALL_READS:
        return bench_big_set(args, ctx, wl, opts, index, model, workdir, preset, wl_name, n_chrom, chrom_len, rank, world, cores)
    # this rank's shard of the read set, generated straight into HBM: `pool` distinct batches, mapped in turn (batch b = reads
    # [(b * world + rank) * reads, +reads) of the synthetic set; each lives in the buffers of a context of its own that never maps)
    n_pool = max(1, min(args.pool, args.steps + args.warmup, int(24e9 // (2 * args.reads * args.samples)) or 1))   # (at most ~24 GB of resident batches: the anchor arenas take what is free)
    gens = [ctx] + [Context(local_rank) for _ in range(n_pool - 1)]
    batches = [wl.reads_device(g, model, (b * world + rank) * args.reads, args.reads) for b, g in enumerate(gens)]
    batch = batches[0]
    t_setup = time.time() - t_setup

    def sync():
        if world > 1:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    recs = None                                # the records of batch 0 = the first reads of the set: what the CPU sample and the upload-inclusive legs map
    turn = 0
    for _ in range(args.warmup):
        r_ = ctx.map_batch(opts, batches[turn % n_pool])
        if turn % n_pool == 0:
            recs = r_
        turn += 1
    acc = {}
    stage_ms, stage_n = {}, {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r_ = ctx.map_batch(opts, batches[turn % n_pool])      # synchronous: returns after the records are back on the host
        if turn % n_pool == 0:
            recs = r_
        turn += 1
        st = ctx.stats()
        for k, v in st.items():
            if k not in ("stages", "ms_total") and not isinstance(v, (list, tuple)):
                acc[k] = acc.get(k, 0) + v
        for k, (ms, n) in st["stages"].items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ms
            stage_n[k] = stage_n.get(k, 0) + n
    sync()
    elapsed = time.perf_counter() - t0
    if recs is None:
        recs = ctx.map_batch(opts, batch)
    for g in gens[1:]:
        g.close()                              # (the other batches' buffers go before the upload-inclusive legs and the index download)
    n_mapped = int(recs["mapped"].sum())
    # The same steps with the batch handed over in (page-locked) HOST memory: every step uploads the int16 signal over PCIe,
    # each sub-batch its slice on its own stream (the upload of one overlaps the kernels of the others).  Reported next to
    # `value`, never instead of it.  (RH_BENCH_IN_FLIGHT=2 keeps two steps in flight with rh_map_submit / rh_map_wait, what
    # kt_pipeline does in the reference; on one GPU it halves each batch's arena share and measured slower, see DESIGN.md.)
    elapsed_h2d = elapsed_full = count_s = None
    if args.h2d:
        host = host_copy(ctx, batch, args)
        pending, depth = [], int(os.environ.get("RH_BENCH_IN_FLIGHT", "1"))
        if depth > 1:                              # the batch slots' contexts allocate their arenas on their first call (seconds): untimed, like the warm-up steps above
            for t_ in [ctx.map_submit(opts, host["batch"]) for _ in range(depth)]:
                ctx.map_wait(t_)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if depth <= 1:
                recs_h = ctx.map_batch(opts, host["batch"])
                continue
            pending.append(ctx.map_submit(opts, host["batch"]))
            if len(pending) == depth:
                recs_h = ctx.map_wait(pending.pop(0))
        while pending:
            recs_h = ctx.map_wait(pending.pop(0))
        sync()
        elapsed_h2d = time.perf_counter() - t0
        assert (recs_h["mapped"] == recs["mapped"]).all() and (recs_h["tag_sl"] == recs["tag_sl"]).all()
        count_s = host["count_s"]
        # ... and without the reader's counts: the whole int16 batch is uploaded before the rounds start (a few steps, for the record)
        n_full = min(args.steps, 3)
        sync()
        t0 = time.perf_counter()
        for _ in range(n_full):
            recs_f = ctx.map_batch(opts, host["batch_full_upload"])
        sync()
        elapsed_full = (time.perf_counter() - t0) / n_full
        assert (recs_f["mapped"] == recs["mapped"]).all()
        ctx._l.rh_pinned_free(host["pin"])
    if world > 1:
        import torch
        t = torch.tensor([elapsed, elapsed_h2d or 0.0, elapsed_full or 0.0], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        elapsed_h2d = float(t[1].item()) or None
        elapsed_full = float(t[2].item()) or None

    if rank == 0:
        acc["n_samples_raw"] = args.reads * args.samples * args.steps
        total_reads = args.reads * world * args.steps
        value = total_reads / elapsed
        kernels = {k: v for k, v in stage_ms.items() if k in ALGO_BYTES and stage_n.get(k)}
        dom, dom_tied = dominant_stage(kernels, acc)
        dom_bytes = ALGO_BYTES[dom](acc)
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        path_bytes = 2 * acc["n_samples_used"] + 16 * acc["n_seeds"] + 8 * acc["n_hits"] + 32 * acc["n_anchors"] + 16 * acc["n_chained"] + 64 * acc["n_reads"]
        dev_ms = sum(kernels.values())
        n_streams = int(os.environ.get("RH_SUB_BATCHES", "3"))
        out = {
            "metric": f"reads/sec mapped ({wl_name} index resident in HBM" + (f", {args.mapopt}" if args.mapopt else "") + ")", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16 signal; fp32/fp64 events; u64/i32 seeding+chaining", "data": "synthetic",
            "config": {"workload": f"{wl_name}: synthetic genome {n_chrom} x {chrom_len} bp + {args.reads} synthetic R9.4 reads/GPU x {args.samples} samples, preset {preset}, "
                                   f"{args.junk}/1024 unmappable reads, index built on the device and resident in HBM ({index.n_keys} keys, {index.n_positions} positions), int16 signal resident in HBM",
                       "reads_per_gpu": args.reads, "samples_per_read": args.samples, "mid_occ": int(opts.mo.mid_occ), "parallelism": f"reads sharded x{world}, index replicated"},
            "value_h2d_included": None if not elapsed_h2d else round(total_reads / elapsed_h2d, 1),
            "ms_per_step_h2d_included": None if not elapsed_h2d else round(1e3 * elapsed_h2d / args.steps, 3),
            "value_h2d_full_upload": None if not elapsed_full else round(args.reads * world / elapsed_full, 1),
            "count_filtered_s_per_batch": None if count_s is None else round(count_s, 4),   # the reader's pass that produces n_filtered (rh_count_filtered, host, one thread per the library's default): outside the timed region
            "value_h2d_included_with_counting": None if not elapsed_h2d or count_s is None else round(total_reads / (elapsed_h2d + count_s * args.steps), 1),
            "metric_note": "`value` = the task contract's number (the int16 signal resident in HBM when the timed region starts); SURVEY.md 8(d) words the metric with the "
                           "host-to-device copy of the batches INCLUDED: that is `value_h2d_included`",
            "batches": f"{n_pool} distinct batches of {args.reads} reads per GPU resident in HBM, mapped in turn (no step maps the batch of the step before)",
            "h2d_note": "value_h2d_included: batch in page-locked host memory with the reader's per-read filtered counts (rh_read_batch_t.n_filtered), the device fetches "
                        "the stretches of signal the rounds consume; value_h2d_full_upload: no counts, the whole int16 batch is uploaded first",
            "gsamples_per_s_consumed": round(acc["n_samples_used"] * world / elapsed / 1e9, 4),
            "gsamples_per_s_input": round(args.reads * args.samples * world * args.steps / elapsed / 1e9, 4),
            "mapped_fraction": round(n_mapped / args.reads, 4),
            "chunks_per_read": round(acc["n_chunks"] / acc["n_reads"], 3),
            "anchors_per_chunk": round(acc["n_anchors"] / max(acc["n_chunks"], 1), 1),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(dom, args, stage_n[dom] / args.steps), "traffic_source": pmc_source(args),
                         "avg_launch_ms": round(stage_ms[dom] / stage_n[dom], 4), "launches": stage_n[dom],
                         "algorithmic_bytes_per_launch": int(dom_bytes / stage_n[dom]),
                         "concurrent_streams": n_streams,
                         "note": "stage durations are HIP-event times on each sub-batch's own stream; with >1 concurrent streams a launch shares the "
                                 "chip with the other streams' kernels, so frac understates the kernel alone (RH_SUB_BATCHES=1 runs: profiles/)"},
            "roofline_stages": [{"kernel": k, "ms_per_step": round(v / args.steps, 3), "achieved": round(b / (v * 1e-3) / 1e9, 3), "frac": round(b / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
                                for k, v, b in sorted(dom_tied, key=lambda t: -t[1])],   # the stages within 10 % of the longest (see dominant_stage)
            "path": {"algorithmic_GB_per_step": round(path_bytes / args.steps / 1e9, 4), "device_ms_per_step": round(dev_ms / args.steps, 3),   # sum over concurrent sub-batch streams
                     "achieved_GBs": round(path_bytes / elapsed / 1e9, 3), "frac_of_hbm_peak": round(path_bytes / elapsed / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items() if stage_n.get(k)},
            "index_build_s": round(t_index, 2), "setup_s": round(t_setup, 2),
        }
        if world == 1 and args.cpu_sample > 0:
            try:
                out["cpu_baseline"], out["paf_sample_identical"] = cpu_baseline(ctx, index, opts, wl, model, workdir, preset, recs, args, cores)
            except Exception as e:   # the baseline is a reported extra: never lose the bench line because of it
                out["cpu_baseline_error"] = repr(e)[:300]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        import shutil
        shutil.rmtree(workdir, ignore_errors=True)


def bench_big_set(args, ctx, wl, opts, index, model, workdir, preset, wl_name, n_chrom, chrom_len, rank, world, cores):
    """`--reads N` beyond one call (BASELINE config 4's per-GPU share: 10 M reads / 8 GPUs = 1.25 M): the set is produced shard by shard the way a reader
    would hand it over - generated, brought to page-locked host memory with the per-read filtered counts - and every shard is mapped by one rh_map_batch
    call FROM HOST MEMORY (the upload is inside the timed region; generation, the copy to the host and the counting are the reader's, outside it).
    value = reads / sum of the calls' times.  The CPU sample is 16 blocks of reads spread evenly over the whole set, PAF compared with the reference's."""
    import copy
    import numpy as np
    import oracle_lib as O
    from rawhash_amd import paf_lines, strip_mt
    from rawhash_amd.api import Reads
    if world != 1:
        sys.exit("--reads beyond one call is a one-GPU measurement (the per-GPU share of a sharded set)")
    N, shard = args.reads, max(1024, args.shard)
    n_blocks = 16
    blk = max(1, (args.cpu_sample if args.cpu_sample > 0 else 4096) // n_blocks)
    starts = [min(N - blk, k * (N // n_blocks) + (N // n_blocks) // 3) for k in range(n_blocks)]     # (not the shards' first reads)
    got_recs = {}
    t_calls, t_count, t_reader, per_call = 0.0, 0.0, 0.0, []
    acc, n_mapped = {}, 0
    ctx.map_batch(opts, wl.reads_device(ctx, model, N, min(shard, N)))      # warm-up (reads beyond the set): arenas as the calls need them, kernels
    for s0 in range(0, N, shard):
        n = min(shard, N - s0)
        t0 = time.perf_counter()
        b = wl.reads_device(ctx, model, s0, n)
        a2 = copy.copy(args); a2.reads = n
        host = host_copy(ctx, b, a2)
        t_reader += time.perf_counter() - t0
        t_count += host["count_s"]
        t0 = time.perf_counter()
        recs = ctx.map_batch(opts, host["batch"])
        dt = time.perf_counter() - t0
        t_calls += dt
        per_call.append(round(1e3 * dt, 1))
        n_mapped += int(recs["mapped"].sum())
        for k, v in ctx.stats().items():
            if k not in ("stages", "ms_total") and not isinstance(v, (list, tuple)):
                acc[k] = acc.get(k, 0) + v
        for st in starts:
            lo, hi = max(st, s0), min(st + blk, s0 + n)
            if lo < hi:
                got_recs.setdefault(st, []).append(recs[lo - s0:hi - s0].copy())
        ctx._l.rh_pinned_free(host["pin"])
    value = N / t_calls
    path_bytes = 2 * acc["n_samples_used"] + 16 * acc["n_seeds"] + 8 * acc["n_hits"] + 32 * acc["n_anchors"] + 16 * acc["n_chained"] + 64 * acc["n_reads"]
    out = {"metric": f"reads/sec mapped ({wl_name} index resident in HBM), {N} reads in {len(per_call)} consecutive calls from host memory", "value": round(value, 1), "unit": "reads/s",
           "n_gpus": 1, "steps": len(per_call), "warmup": 1, "ms_per_step": round(1e3 * t_calls / len(per_call), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int16 signal; fp32/fp64 events; u64/i32 seeding+chaining", "data": "synthetic",
           "config": {"workload": f"{wl_name}: synthetic genome {n_chrom} x {chrom_len} bp, {N} synthetic R9.4 reads x {args.samples} samples (BASELINE config 4's share of one of 8 GPUs), preset {preset}, "
                                  f"{args.junk}/1024 unmappable reads, index built on the device and resident in HBM ({index.n_keys} keys, {index.n_positions} positions); every call gets "
                                  f"{shard} reads in page-locked host memory with the reader's filtered counts and fetches the signal it consumes over PCIe inside the timed region",
                      "reads_total": N, "reads_per_call": shard, "samples_per_read": args.samples, "mid_occ": int(opts.mo.mid_occ)},
           "metric_note": "a `step` here is one rh_map_batch call on one shard of the set; value = upload-inclusive (SURVEY.md 8(d)'s wording of the metric)",
           "ms_per_call": per_call, "reader_s_total": round(t_reader, 1), "count_filtered_s_total": round(t_count, 2),
           "value_with_counting": round(N / (t_calls + t_count), 1),
           "mapped_fraction": round(n_mapped / N, 4), "chunks_per_read": round(acc["n_chunks"] / acc["n_reads"], 3),
           "path": {"algorithmic_GB_total": round(path_bytes / 1e9, 2), "achieved_GBs": round(path_bytes / t_calls / 1e9, 3), "frac_of_hbm_peak": round(path_bytes / t_calls / 1e9 / HBM_PEAK_GBS, 5)}}
    if args.cpu_sample != 0 and O.have_reference():
        try:
            t0 = time.time()
            index.download(ctx, n_threads=min(cores, 64))
            ind = os.path.join(workdir, "ref.ind")
            index.write(ind)
            parts = [wl.reads(model, st, blk, n_threads=min(cores, 64), with_names=True) for st in starts]
            off = np.zeros(n_blocks * blk + 1, dtype=np.uint64)
            off[1:] = np.cumsum(np.concatenate([np.diff(p.offsets.astype(np.int64)) for p in parts]))
            sample = Reads(np.concatenate([p.samples for p in parts]), off, sum((p.names for p in parts), []), parts[0].cal_offset[0], parts[0].cal_scale[0])
            srecs = np.concatenate([np.concatenate(got_recs[st]) for st in starts])
            srecs["read_idx"] = np.arange(len(srecs), dtype=srecs["read_idx"].dtype)       # (they were numbered within their calls)
            got = [strip_mt(x) for x in paf_lines(index, srecs, sample.names)]
            rhr, paf = os.path.join(workdir, "cpu_sample.rhr"), os.path.join(workdir, "ref.paf")
            sample.write(rhr, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
            sweep = [int(x) for x in args.cpu_threads.split(",")] if args.cpu_threads else sorted({max(1, cores // 16), max(1, cores // 8)})
            with open(paf, "w") as fo:
                p = subprocess.run([O.REF_HARNESS, "map", preset, ind, rhr, ",".join(str(t) for t in sweep)], stdout=fo, stderr=subprocess.PIPE, text=True, timeout=3000)
            runs = [(int(t), float(sec)) for sec, t in re.findall(r"map phase ([0-9.]+) s, threads (\d+)", p.stderr)]
            with open(paf) as f:
                want = [O.strip_mt(x) for x in f]
            bt, bs = min(runs, key=lambda r: r[1])
            out["cpu_baseline"] = {"value": round(len(sample) / bs, 1), "unit": "reads/s", "cores": bt, "threads": bt, "host_threads": cores, "kind": "reference",
                                   "sample": f"{n_blocks} blocks of {blk} reads spread over the {N} (first reads {starts[0]}, {starts[1]}, ... {starts[-1]}), map phase {bs:.2f} s, unmodified RawHash2 sources (oracle/Makefile)",
                                   "thread_sweep_reads_per_s": {str(t): round(len(sample) / sec, 1) for t, sec in runs}, "paf_sha1": hashlib.sha1("\n".join(want).encode()).hexdigest()}
            out["paf_sample_identical"] = got == want
            if got != want:
                out["cpu_baseline"]["paf_lines_differing"] = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
        except Exception as e:
            out["cpu_baseline_error"] = repr(e)[:300]
    print(json.dumps(out), flush=True)
    ctx.close()
    import shutil
    shutil.rmtree(workdir, ignore_errors=True)


def bench_ava(args):
    """BASELINE.json configs[4], a secondary line (`--workload ava`): Rawsamble all-vs-all overlapping of 50 k synthetic reads
    of 3000 bases (27 k samples) drawn from a 1 Mbp genome (~150x).  A step = every read overlapped against the signal-target
    index of all of them (one whole-read round, every reported chain a record); the index is built on the device from the same
    reads before the timed region (its time is reported).  One GPU: queries shard like reads of the headline path, the
    index build does not (it is a second or so)."""
    import numpy as np
    import oracle_lib as O
    from rawhash_amd import Context, Index, MapOptions, SynthWorkload, paf_lines, strip_mt
    # N GPUs: the 50 k reads are the targets on every GPU (each rank builds the same signal-target index from them: a fraction of
    # a second, no broadcast needed) and the queries are sharded: rank r overlaps its contiguous share of the reads against all
    # of them.  Total work is fixed: "scaling": "strong".  No collective on the data path.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RH_BENCH_BACKEND", "nccl")
        if os.environ.get("RH_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    n = args.reads if args.reads > 0 else 50_000
    n_samples, genome, preset = 27_000, 1_000_000, "ava"
    cores = os.cpu_count() or 8
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    workdir = os.path.join(shm, f"rawhash_amd_bench_ava_{os.getpid()}_{rank}")
    os.makedirs(workdir, exist_ok=True)
    wl = SynthWorkload(chrom_len=genome, n_chrom=1, n_samples=n_samples, junk_per_1024=50, noise_q24=150_000, read_seed=23)
    _, model = wl.write_reference(workdir)
    opts = MapOptions(preset)
    targets = wl.reads(model, 0, n, n_threads=min(cores, 64), with_names=True)
    ctx = Context(local_rank)
    t0 = time.perf_counter()
    index = Index.build_signals_device(ctx, targets, model, opts)
    t_index = time.perf_counter() - t0
    opts.update(index)
    q0, q1 = n * rank // world, n * (rank + 1) // world             # this rank's queries
    reads = targets if world == 1 else targets.subset(range(q0, q1))
    cap = 400 * len(reads) + 1024
    dev = wl.reads_device(ctx, model, q0, q1 - q0)                 # the same reads generated straight into HBM (resident when the timed region starts)

    def sync():
        if world > 1:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        recs, off = ctx.map_batch_multi(opts, reads, index, max_records=cap, device_batch=dev)
    stage_ms, stage_n, acc = {}, {}, {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs, off = ctx.map_batch_multi(opts, reads, index, max_records=cap, device_batch=dev)
        st = ctx.stats()
        for k, v in st.items():
            if k not in ("stages", "ms_total") and not isinstance(v, (list, tuple)):
                acc[k] = acc.get(k, 0) + v
        for k, (ms, c) in st["stages"].items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ms
            stage_n[k] = stage_n.get(k, 0) + c
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_h2d = None
    if args.h2d:                                                   # the same steps from host memory: the int16 signal is uploaded inside every step
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            recs_h, _ = ctx.map_batch_multi(opts, reads, index, max_records=cap)
        sync()
        elapsed_h2d = time.perf_counter() - t0
        assert len(recs_h) == len(recs)
    n_records = len(recs)
    if world > 1:
        import torch
        t = torch.tensor([elapsed, elapsed_h2d or 0.0], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_h2d = float(t[0].item()), (float(t[1].item()) or None)
        c = torch.tensor([n_records], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_records = int(c.item())
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
        import shutil
        shutil.rmtree(workdir, ignore_errors=True)
        return
    kernels = {k: v for k, v in stage_ms.items() if k in ALGO_BYTES and stage_n.get(k)}
    dom, dom_tied = dominant_stage(kernels, acc)
    dom_bytes = ALGO_BYTES[dom](acc)
    achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
    out = {
        "metric": "reads/sec overlapped all-vs-all (Rawsamble, signal-target index resident in HBM)", "value": round(n * args.steps / elapsed, 1), "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "int16 signal; fp32/fp64 events; u64/i32 seeding+chaining", "data": "synthetic",
        "config": {"workload": f"Rawsamble all-vs-all: {n} synthetic R9.4 reads x {n_samples} samples from a {genome} bp genome, preset {preset}, signal-target index built "
                               f"on the device from the same reads ({index.n_keys} keys, {index.n_positions} positions), int16 signal resident in HBM",
                   "reads_per_gpu": q1 - q0, "samples_per_read": n_samples, "mid_occ": int(opts.mo.mid_occ), "parallelism": f"queries sharded x{world}, index built on every GPU"},
        "value_h2d_included": None if not elapsed_h2d else round(n * args.steps / elapsed_h2d, 1),
        "records_per_step": int(n_records),
        "index_build_s": round(t_index, 3), "value_with_index_build": round(n / (elapsed / args.steps + t_index), 1),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                     "avg_launch_ms": round(stage_ms[dom] / stage_n[dom], 4), "launches": stage_n[dom], "algorithmic_bytes_per_launch": int(dom_bytes / stage_n[dom])},
        "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items() if stage_n.get(k)},
    }
    sample = args.cpu_sample if args.cpu_sample >= 0 else 20000
    if world == 1 and sample > 0 and O.have_reference():
        # the unmodified reference: its functions build the signal-target index of all reads (`ref_harness sigindex`), then its
        # kt_for(map_worker_for) overlaps the first `sample` reads against it; PAF of that sample compared with the device's
        try:
            sample = min(sample, n)
            rhr_all, rhr_s, ind = os.path.join(workdir, "all.rhr"), os.path.join(workdir, "sample.rhr"), os.path.join(workdir, "ref.ind")
            targets.write(rhr_all, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
            targets.subset(range(sample)).write(rhr_s, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
            t0 = time.perf_counter()
            subprocess.run([O.REF_HARNESS, "sigindex", preset, rhr_all, model, ind, str(min(cores, 64))], check=True, stderr=subprocess.DEVNULL, timeout=3000)
            t_ref_index = time.perf_counter() - t0
            sweep = [int(x) for x in args.cpu_threads.split(",")] if args.cpu_threads else sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores})
            paf = os.path.join(workdir, "ref.paf")
            with open(paf, "w") as fo:
                p = subprocess.run([O.REF_HARNESS, "map", preset, ind, rhr_s, ",".join(str(t) for t in sweep)], stdout=fo, stderr=subprocess.PIPE, text=True, timeout=3000)
            runs = [(int(t), float(sec)) for sec, t in re.findall(r"map phase ([0-9.]+) s, threads (\d+)", p.stderr)]
            with open(paf) as f:
                want = [O.strip_mt(x) for x in f]
            got = [strip_mt(x) for x in paf_lines(index, recs[: int(off[sample])], reads.names)]
            best_t, best_s = min(runs, key=lambda r: r[1])
            out["cpu_baseline"] = {"value": round(sample / best_s, 1), "unit": "reads/s", "cores": best_t, "threads": best_t, "host_threads": cores, "kind": "reference",
                                   "sample": f"first {sample} reads overlapped against the index of all {n}, map phase {best_s:.2f} s, unmodified RawHash2 sources (oracle/Makefile)",
                                   "thread_sweep_reads_per_s": {str(t): round(sample / sec, 1) for t, sec in runs},
                                   "reference_index_build_s": round(t_ref_index, 2)}
            out["paf_sample_identical"] = got == want
            if got != want:
                out["cpu_baseline"]["paf_lines_differing"] = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
        except Exception as e:
            out["cpu_baseline_error"] = repr(e)[:300]
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    import shutil
    shutil.rmtree(workdir, ignore_errors=True)


def host_copy(ctx, batch, args):
    """The device-resident batch copied into one page-locked host allocation (the int16 staging buffer a reader fills)."""
    import numpy as np
    from rawhash_amd import _capi
    l = ctx._l
    n = args.reads
    n_smp = n * args.samples
    sizes = [n_smp * 2, (n + 1) * 8, n * 8, n * 4]
    offs = [0]
    for sz in sizes:
        offs.append((offs[-1] + sz + 255) // 256 * 256)
    pin = l.rh_pinned_alloc(offs[-1])
    if not pin:
        raise RuntimeError(_capi.last_error(l))
    if l.rh_read_batch_to_host(ctx.h, C.byref(batch), pin + offs[0], pin + offs[1], pin + offs[2], pin + offs[3]) != 0:
        raise RuntimeError(_capi.last_error(l))
    b = _capi.ReadBatch(n, pin + offs[0], pin + offs[1], pin + offs[2], pin + offs[3], None, 0)
    # what a reader knows when it has decoded a read (ri_read_sig's l_sig, rsig.c:496-503; rh_reads_* count while they stage): with it the
    # device fetches only the stretches of signal the rounds consume.  Counted here, outside the timed region, like the file loading of the CPU baseline.
    t0 = time.perf_counter()
    nf = _capi.count_filtered(b, lib=l)
    t_count = time.perf_counter() - t0
    b_counted = _capi.ReadBatch(n, pin + offs[0], pin + offs[1], pin + offs[2], pin + offs[3], None, 0, 0, nf.ctypes.data)
    b_counted._keep = nf
    return {"pin": pin, "batch": b_counted, "batch_full_upload": b, "count_s": t_count}


def pmc_source(args):
    """Where roofline.traffic comes from: NOT measured in this run - a file the builder committed (rocprofv3 --pmc passes of the same command)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    if d.get("workload") != args.workload or d.get("reads") != args.reads or d.get("samples") != args.samples or d.get("junk") != args.junk:
        return None
    return {"file": "profiles/pmc_traffic.json", "measured_in_this_run": False, "made_by": "profiles/collect_pmc.py (builder-run rocprofv3 --pmc passes of this command, one stream)",
            "code_state": d.get("commit"), "round": d.get("round")}


def pmc_traffic(stage, args, launches_per_step):
    """HBM bytes per launch of the dominant stage from the committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_traffic.json, made by profiles/collect_pmc.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    if d.get("workload") != args.workload or d.get("reads") != args.reads or d.get("samples") != args.samples or d.get("junk") != args.junk:
        return None
    e = d.get("stages", {}).get(stage)
    return None if e is None or "bytes_per_step" not in e else int(e["bytes_per_step"] / max(launches_per_step, 1))


def cpu_baseline(ctx, index, opts, wl, model, workdir, preset, recs, args, cores):
    """The reference on the host cores over the first `cpu_sample` reads of the same set, and PAF parity on that sample.

    kind "reference": the unmodified RawHash2 sources (oracle/_ref/ref_harness, prebuilt where /root/reference exists): its
    own .ind loader reads the index this library wrote, its `kt_for(map_worker_for)` maps mini-batches of 500 M samples (-K)
    and its own printer writes the PAF; reported = map-phase time (file loading excluded), best of a thread sweep.
    kind "port": oracle/rh_oracle.c (the bit-identical restatement) when the reference binary is not there."""
    import numpy as np
    import oracle_lib as O
    from rawhash_amd import paf_lines, strip_mt
    n = min(args.cpu_sample, args.reads)
    t0 = time.time()
    index.download(ctx, n_threads=min(cores, 64))
    ind = os.path.join(workdir, "ref.ind")
    index.write(ind)
    t_ind = time.time() - t0
    reads = wl.reads(model, 0, n, n_threads=min(cores, 64), with_names=True)
    got = [strip_mt(x) for x in paf_lines(index, recs[:n], reads.names)]
    sweep = [int(x) for x in args.cpu_threads.split(",")] if args.cpu_threads else sorted({max(1, cores // 16), max(1, cores // 8), max(1, cores * 3 // 16), max(1, cores // 4), max(1, cores // 2), cores})
    if O.have_reference():
        rhr = os.path.join(workdir, "cpu_sample.rhr")
        reads.write(rhr, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
        paf = os.path.join(workdir, "ref.paf")
        # Two passes of the harness: memory policy as the process finds it (the single-threaded .ind loader first-touches the whole
        # index on its own NUMA node), and every page interleaved over all nodes (RH_REF_INTERLEAVE=1 = `numactl --interleave=all`,
        # which the image lacks).  Both thread sweeps are reported; the baseline is the best run of either.
        half = [t for t in sweep if t <= max(1, cores // 4)] or sweep[:1]
        passes = [("default", {}, half if len(sweep) > 3 else sweep), ("interleave", {"RH_REF_INTERLEAVE": "1"}, [t for t in sweep if t >= max(1, cores // 8)])]
        if args.cpu_threads:
            passes = [(nm, ev, sweep) for nm, ev, _ in passes]
        runs, want, load_s, notes = {}, None, {}, []
        for nm, ev, ts in passes:
            with open(paf, "w") as fo:
                p = subprocess.run([O.REF_HARNESS, "map", preset, ind, rhr, ",".join(str(t) for t in ts)], stdout=fo, stderr=subprocess.PIPE, text=True, timeout=3000,
                                   env=dict(os.environ, **ev, **getattr(args, "ref_env", {})))
            rr = [(int(t), float(sec)) for sec, t in re.findall(r"map phase ([0-9.]+) s, threads (\d+)", p.stderr)]
            if p.returncode != 0 or not rr:
                notes.append(f"{nm}: harness rc {p.returncode}")
                continue
            runs[nm] = rr
            load = re.search(r"index loaded in ([0-9.]+) s", p.stderr)
            load_s[nm] = float(load.group(1)) if load else None
            il = re.search(r"interleaved over (\d+) NUMA", p.stderr)
            if il:
                notes.append(f"{nm}: {il.group(1)} NUMA node(s)")
            with open(paf) as f:
                w = [O.strip_mt(x) for x in f]
            if want is not None and w != want:
                notes.append(f"{nm}: PAF differs from the first pass")
            want = want or w
        if runs:
            identical = got == want
            best_nm, (best_t, best_s) = min(((nm, min(rr, key=lambda r: r[1])) for nm, rr in runs.items()), key=lambda x: x[1][1])
            try:
                numa_nodes = len([d for d in os.listdir("/sys/devices/system/node") if re.fullmatch(r"node\d+", d)])
            except OSError:
                numa_nodes = None
            base = {"value": round(n / best_s, 1), "unit": "reads/s", "cores": best_t, "threads": best_t, "host_threads": cores, "kind": "reference", "memory_policy": best_nm,
                    "sample": f"first {n} reads of the same synthetic set, map phase {best_s:.2f} s (file loading excluded), unmodified RawHash2 sources built by "
                              f"oracle/Makefile (-O3 -ffp-contract=off), kt_for over 500 M-sample mini-batches",
                    "thread_sweep_reads_per_s": {nm: {str(t): round(n / sec, 1) for t, sec in rr} for nm, rr in runs.items()},
                    "numa_nodes": numa_nodes, "notes": notes,
                    "index_file_s": round(t_ind, 1), "reference_index_load_s": load_s,
                    "paf_sha1": hashlib.sha1("\n".join(want).encode()).hexdigest()}
            if not identical:
                base["paf_lines_differing"] = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
            # the same sources with the stock vector flags (src/Makefile:7 has -march=native; oracle/Makefile ref_v4 builds -march=x86-64-v4 in its place - the binary is
            # made in a container whose CPU is not this host's): the other denominator of the GPU / CPU ratio, at the best thread counts of the sweep above
            try:
                if os.path.exists(O.REF_HARNESS_V4) and "avx512f" in open("/proc/cpuinfo").read():
                    ts = sorted({t for rr in runs.values() for t, _ in sorted(rr, key=lambda r: r[1])[:2]})
                    with open(paf, "w") as fo:
                        p = subprocess.run([O.REF_HARNESS_V4, "map", preset, ind, rhr, ",".join(str(t) for t in ts)], stdout=fo, stderr=subprocess.PIPE, text=True, timeout=3000,
                                           env=dict(os.environ, **passes[[nm for nm, _, _ in passes].index(best_nm)][1], **getattr(args, "ref_env", {})))
                    rr = [(int(t), float(sec)) for sec, t in re.findall(r"map phase ([0-9.]+) s, threads (\d+)", p.stderr)]
                    if p.returncode == 0 and rr:
                        bt, bs = min(rr, key=lambda r: r[1])
                        with open(paf) as f:
                            w4 = [O.strip_mt(x) for x in f]
                        base["value_stock_flags"] = round(n / bs, 1)
                        base["stock_flags"] = {"flags": "-O3 -march=x86-64-v4 -ffp-contract=off (for the stock -march=native)", "threads": bt, "memory_policy": best_nm,
                                               "thread_sweep_reads_per_s": {str(t): round(n / sec, 1) for t, sec in rr}, "paf_identical_to_portable_build": w4 == want}
                    else:
                        base["stock_flags"] = {"error": f"harness rc {p.returncode}"}
                else:
                    base["stock_flags"] = {"error": "no AVX-512 host or no oracle/_ref/ref_harness_v4"}
            except Exception as e:
                base["stock_flags"] = {"error": repr(e)[:200]}
            return base, identical
    oix = O.OracleIndex(ind)
    _, mo = O.preset(preset)
    O.lib().ro_mapopt_update(C.byref(mo), oix.h)
    b = reads.batch()
    t0 = time.perf_counter()
    orecs = O.map_batch(oix, mo, b, n_threads=cores)
    dt = time.perf_counter() - t0
    want = [O.strip_mt(x) for x in O.paf_lines(oix, orecs, reads.names)]
    return ({"value": round(n / dt, 1), "unit": "reads/s", "cores": cores, "threads": cores, "kind": "port",
             "sample": f"first {n} reads of the same synthetic set, {dt:.2f} s wall, oracle/rh_oracle.c with {cores} pthreads",
             "mapped_fraction": round(float(orecs['mapped'].mean()), 4)}, got == want)


if __name__ == "__main__":
    main()

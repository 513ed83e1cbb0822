"""N > 1 path on CPU: 2 gloo processes shard the reads, broadcast the index blob, map their shard (kernel sources under
the SIMT emulator) and gather; the result must equal the single-process result and the oracle's PAF."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rawhash_amd.dist import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "emu"))
import numpy as np, torch, torch.distributed as dist
import build_emu
from rawhash_amd import _capi
from rawhash_amd.api import Context, Index, MapOptions, SynthWorkload, paf_lines, strip_mt
from rawhash_amd.dist import shard_bounds, replicate_index, gather_records
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = _capi.load(build_emu.build())
d = sys.argv[2]
wl = SynthWorkload(chrom_len=300_000, n_chrom=2, n_samples=12_000, junk_per_1024=150, noise_q24=150_000, lib=lib)
opts = MapOptions("sensitive", lib=lib)
ctx = Context(0, lib=lib)
index = None
if rank == 0:
    fasta, model = wl.write_reference(d)
    index = Index.build(fasta, model, opts, out_ind=os.path.join(d, "ref.ind"), lib=lib)
    opts.update(index)
dist.barrier()
keep = replicate_index(ctx, opts, index, device="cpu")
n = 21
lo, hi = shard_bounds(n, rank, world)
reads = wl.reads(os.path.join(d, "model.txt"), lo, hi - lo)
recs = ctx.map_batch(opts, reads)
allr = gather_records(recs, lo)
if rank == 0:
    full = wl.reads(os.path.join(d, "model.txt"), 0, n)
    single = ctx.map_batch(opts, full)
    assert np.array_equal(allr, single), "sharded result differs from the single-process result"
    with open(os.path.join(d, "paf.pkl"), "wb") as f:
        pickle.dump([strip_mt(x) for x in paf_lines(index, allr, full.names, lib=lib)], f)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 101):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process_and_oracle(tmp_path, emu_lib):
    import pickle
    import ctypes as C
    import oracle_lib as O
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RH_BCAST_PIECE_BYTES="100000")   # (the index blob goes over in dozens of pieces)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", str(script), ROOT, str(tmp_path)], check=True, env=env, timeout=600)
    got = pickle.load(open(tmp_path / "paf.pkl", "rb"))
    from rawhash_amd.api import SynthWorkload
    wl = SynthWorkload(chrom_len=300_000, n_chrom=2, n_samples=12_000, junk_per_1024=150, noise_q24=150_000, lib=emu_lib)
    reads = wl.reads(str(tmp_path / "model.txt"), 0, 21)
    oix = O.OracleIndex(str(tmp_path / "ref.ind"))
    _, mo = O.preset("sensitive")
    O.lib().ro_mapopt_update(C.byref(mo), oix.h)
    want = [O.strip_mt(x) for x in O.paf_lines(oix, O.map_batch(oix, mo, reads.batch()), reads.names)]
    assert got == want


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """The multi-GPU flow of bench.py (rank 0 builds the index on its device, blob broadcast, every rank adopts it and maps its
    shard, barrier + max-over-ranks timing) run as the driver launches it, with two ranks sharing the one GPU of the test box
    (gloo instead of RCCL: two RCCL ranks cannot share a device)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RH_BENCH_BACKEND="gloo", RH_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "ecoli", "--reads", "6000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert 0.85 < d["mapped_fraction"] < 0.95 and d["value_h2d_included"] > 0


@pytest.mark.gpu
def test_bench_ava_two_ranks_on_one_gpu():
    """bench.py --workload ava on N > 1: every rank builds the signal-target index of all reads on its device and overlaps its
    share of the queries (strong scaling, no collective on the data path) - two ranks sharing the test box's GPU over gloo; the
    records of the two shards add up to the single-rank count."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RH_BENCH_BACKEND="gloo", RH_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--workload", "ava", "--reads", "3000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2"] + args
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["records_per_step"] == d1["records_per_step"] > 3000


def test_bench_launch_plan_for_eight_gpus():
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run environment), parsed without touching a GPU:
    every rank takes its own contiguous shard, only a single-GPU run carries the CPU baseline leg, and a bare
    `--gpus 8` without the launcher is refused."""
    import json
    plans = []
    for r in range(8):
        env = dict(os.environ, WORLD_SIZE="8", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--dry-run"],
                             check=True, capture_output=True, text=True, env=env, timeout=120).stdout
        plans.append(json.loads(out.strip().splitlines()[-1]))
    assert [p["rank"] for p in plans] == list(range(8)) and all(p["world"] == 8 and p["gpus"] == 8 for p in plans)
    assert [p["first_read"] for p in plans] == [r * plans[0]["reads_per_gpu"] for r in range(8)]
    assert all(p["steps"] == 4 and p["warmup"] == 2 and p["backend"] == "nccl" and not p["cpu_baseline"] for p in plans)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], check=True, capture_output=True, text=True, env=env, timeout=120)
    assert json.loads(one.stdout.strip().splitlines()[-1])["cpu_baseline"] is True
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert bad.returncode != 0 and "torch.distributed.run" in (bad.stderr + bad.stdout)

"""Parity of the HIP path (through the C ABI) with the CPU oracle on a real MI355X."""
import os

import numpy as np
import pytest

import parity_checks as pc
from rawhash_amd.api import Context, paf_lines, strip_mt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wl(make_workload):
    return make_workload(n_reads=400, n_samples=24_000, chrom_len=600_000)


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory, wl):
    c = gpu_ctx_factory()
    c.upload(wl.index)
    return c


def test_native_library_loaded(product_lib):
    assert product_lib.rh_device_count() >= 1
    import os
    maps = open("/proc/self/maps").read()
    assert "librawhash_amd.so" in maps


def test_events_bit_exact(ctx, wl):
    pc.check_events(ctx, wl, chunks=(0, 1, 3, 5))
    pc.check_events_variants(ctx, wl, chunks=(0, 4))
    for seed in (1, 2, 3):
        pc.check_events_odd_signals(ctx, wl, seed=seed)


def test_window_division_shortcut(ctx):
    """The t-statistic kernel divides by the window width with a reciprocal product + one remainder correction; on this GPU
    that must equal the IEEE quotient for every fp32 input the prefix sums can produce (all 2^32 patterns are tried)."""
    import ctypes, os, rawhash_amd
    lib = ctypes.CDLL(os.path.join(os.path.dirname(rawhash_amd.__file__), "librawhash_amd.so"))
    lib.rh_debug_div_const_check.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    for w in range(2, 16):
        out = (ctypes.c_ulonglong * 2)()
        assert lib.rh_debug_div_const_check(w, out) == 0
        assert out[1] == 0, f"w={w}: {out[1]} reachable inputs differ ({out[0]} overall)"


def test_stage_chain(ctx, wl):
    pc.check_stages(ctx, wl)


def test_exact_sort(ctx):
    pc.check_sort(ctx, seed=2, n_seg=400)
    pc.check_sort(ctx, seed=5, n_seg=40, big=(8193, 9000, 20000, 70000, 30000, 12345, 100000, 16384, 50000, 65537, 33333, 9999))
    pc.check_sort(ctx, seed=6, n_seg=5000, tiny=True)


def test_sort_one_word_records(ctx):
    """The anchor sort on 8-byte records (the mapping path's anchor format): the tie-free LDS path (rh_sort.hip: sort_fast) and what it hands back
    to the exact passes, at every LDS class boundary, direct and through the any-order levels of the multi-workgroup sorter (the strand x target
    buckets of a human-scale chunk: ~2 k records each), thousands of segments at once."""
    pc.check_sort_packed(ctx, seed=31)
    pc.check_sort_packed(ctx, seed=32, sizes=[int(x) for x in np.random.default_rng(32).integers(0, 8193, size=1500)])
    pc.check_sort_packed(ctx, seed=33, sizes=[90_000, 60_000, 9_000, 200_000, 12_000, 700, 40_000, 150_000, 8_193], any_order=True)
    pc.check_sort_packed(ctx, seed=34, sizes=[20_000 + 331 * i for i in range(120)], any_order=True)
    pc.check_sort_packed(ctx, seed=35, sizes=[3_000 + 7 * i for i in range(400)], lo=23, mid=1)      # a small index: whole chunks in one LDS segment


def test_stage_regions(ctx, wl):
    """a17-a19 (mm_gen_regs / mm_set_parent / mm_select_sub / mm_set_mapq) at stage level against the oracle."""
    checked, with_regs = pc.check_regions(ctx, wl, seed=9, n_reads=300, max_n=2500)
    assert checked > 300 and with_regs > 150


def test_consumed_prefix_staging(ctx, wl):
    """rh_read_batch_t::n_filtered + page-locked samples: the device fetches the stretches of signal the rounds consume straight from host memory
    (k_need / k_fetch); records identical to the whole-batch upload, which is pinned to the oracle; a wrong count fails the call."""
    assert pc.check_consumed_prefix_staging(ctx, wl, pinned=True, seed=5) > 0
    assert pc.check_consumed_prefix_staging(ctx, wl, pinned=False, seed=6) > 0   # pageable memory: the batch is copied whole, the counts are only checked


@pytest.mark.parametrize("mapopt", [{"flag": 2}, {"flag": 2, "rmq_size_cap": 40, "rmq_inner_dist": 300}, {"bw_long": 2000}, {"flag": 2, "bw_long": 1500}],
                         ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_rmq_chaining(make_workload, product_lib, gpu_ctx_factory, mapopt):
    """f4: --rmq / --bw-long (mg_lchain_rmq, lchain.c:606) on the device: adversarial anchor sets at stage level and 300 reads end to
    end against the oracle; the goldens printed by the reference for these variants are part of test_golden_paf."""
    w = make_workload(n_reads=300, n_samples=20_000, mapopt=mapopt)
    c = gpu_ctx_factory()
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=16, n_reads=200, max_n=1500)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)


@pytest.mark.parametrize("mapopt", [{"max_num_skips": 2, "max_chain_iter": 40}, {"max_num_skips": 25, "max_chain_iter": 5}, {"max_num_skips": 0, "max_chain_iter": 3}],
                         ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_chain_skip_and_iter_limits(make_workload, product_lib, gpu_ctx_factory, mapopt):
    """--max-skips / --max-iterations below the size of a small cluster: the generic DP step of k_chain_wave's small-cluster path (every preset takes the skip-free
    one, k_chain_wave<true>); plus the generic step forced under the default options (RH_CHAIN_GENERIC) in test_chain_generic_step."""
    w = make_workload(n_reads=300, n_samples=20_000, mapopt=mapopt)
    c = gpu_ctx_factory()
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=26, n_reads=200, max_n=1500)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)


def test_chain_generic_step(ctx, wl, monkeypatch):
    monkeypatch.setenv("RH_CHAIN_GENERIC", "1")
    n_an, n_ch, n_u = pc.check_chain_synthetic(ctx, wl, seed=27, n_reads=200, max_n=1500)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(ctx, wl)


@pytest.mark.parametrize("bt_class", [1, 2, 3])
def test_backtrack_mark_classes(ctx, wl, monkeypatch, bt_class):
    """The backtrack's used-marks and claim stamps: LDS bits + a stamp table for reads of up to 131 072 / 262 144 / 524 288 anchors (k_backtrack_spec<256 | 512, words>,
    rh_post.hip), HBM beyond.  Small inputs only reach the first class by themselves: RH_BT_LDS_MIN_CLASS sends every read through the second, the third and the HBM form."""
    monkeypatch.setenv("RH_BT_LDS_MIN_CLASS", str(bt_class))
    n_an, n_ch, n_u = pc.check_chain_synthetic(ctx, wl, seed=30 + bt_class, n_reads=200, max_n=1500)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(ctx, wl)


@pytest.mark.parametrize("min_class", [0, 1, 2, 3])
def test_rmq_storage_classes_on_device(make_workload, product_lib, gpu_ctx_factory, monkeypatch, min_class):
    """The RMQ trees' four storage classes (LDS rings of 64 / 128 / 512 nodes, HBM: rh_chain.hip k_chain_rmq<RING, class>) each forced on the real device -
    RH_RQ_MIN_CLASS puts every read into that class or a wider one, which is as exact - against the oracle, end to end and on adversarial anchor sets;
    the per-class counters of the call say which classes ran."""
    monkeypatch.setenv("RH_RQ_MIN_CLASS", str(min_class))
    w = make_workload(n_reads=300, n_samples=20_000, mapopt={"flag": 2, "rmq_inner_dist": 300})
    c = gpu_ctx_factory()
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=17 + min_class, n_reads=120, max_n=1500)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)
    cls = c.stats()["n_rmq_class"]
    assert sum(cls[:min_class]) == 0 and cls[min_class] > 0, cls


def test_backtrack_widths_agree(make_workload, product_lib, gpu_ctx_factory, monkeypatch):
    """k_backtrack_spec<256> (a workgroup per read, 256 candidates a round: the default) and <64> (a wavefront per read, RH_BT_WAVE=1) against the oracle
    on the same reads - the speculative walks commit in a different order in the two, the chains must not differ."""
    w = make_workload(n_reads=300, n_samples=20_000)
    for wave in (False, True):
        if wave:
            monkeypatch.setenv("RH_BT_WAVE", "1")
        c = gpu_ctx_factory()
        c.upload(w.index)
        for n_reads, max_n, seed in ((48, 900, 41), (1200, 260, 42)):
            n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=seed, n_reads=n_reads, max_n=max_n)
            assert n_ch > 0 and n_u > 0
        pc.check_e2e(c, w)


@pytest.mark.parametrize("mapopt", [{"flag": 0x40}, {"flag": 0x40, "dtw_border_constraint": 0}, {"flag": 0x40, "dtw_fill_method": 0, "dtw_min_score": 5.0}],
                         ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_dtw_rescoring(make_workload, product_lib, gpu_ctx_factory, mapopt):
    """f4: --dtw-evaluate-chains on a --store-sig index, 300 reads end to end against the oracle (the reference-printed goldens of these
    variants are part of test_golden_paf)."""
    w = make_workload(n_reads=300, n_samples=20_000, idxflag=0x10, mapopt=mapopt)
    c = gpu_ctx_factory()
    c.upload(w.index)
    recs = pc.check_e2e(c, w)
    st = c.stats()      # MAPQ and the mapping decision are the device's wherever the host's logf cannot change the truncated MAPQ (k_dtw_decide): nearly all reads
    assert st["n_dtw_device"] > 0 and st["n_dtw_host"] * 50 <= st["n_dtw_device"], (st["n_dtw_device"], st["n_dtw_host"])
    assert recs["mapped"].sum() > 100


def test_dtw_host_mapq_path(make_workload, product_lib, gpu_ctx_factory, monkeypatch):
    """The reads k_dtw_decide leaves to the host's libm take the round trip every read took before round 6; RH_DTW_HOST_MAPQ=1 sends all of them that way."""
    monkeypatch.setenv("RH_DTW_HOST_MAPQ", "1")
    w = make_workload(n_reads=300, n_samples=20_000, idxflag=0x10, mapopt={"flag": 0x40})
    c = gpu_ctx_factory()
    c.upload(w.index)
    pc.check_e2e(c, w)
    st = c.stats()
    assert st["n_dtw_device"] == 0 and st["n_dtw_host"] > 0


def test_device_index_store_sig_and_dtw(make_workload, product_lib, gpu_ctx_factory, tmp_path):
    """--store-sig on the DEVICE index builder (the levels k_ix_levels computes, laid out as rindex.c:590-598 writes them): the .ind written
    from the device-built index is the host builder's byte for byte (which tests/test_oracle.py pins to the reference's, --store-sig
    included), and DTW re-scoring against the device-built, never-uploaded index prints the oracle's PAF."""
    from rawhash_amd.api import Index
    w = make_workload(n_reads=200, n_samples=20_000, idxflag=0x10, mapopt={"flag": 0x40})
    c = gpu_ctx_factory()
    dev = Index.build_device(c, w.fasta, w.model, w.opts, n_threads=8)
    dev.download(c)
    out = str(tmp_path / "device_store_sig.ind")
    dev.write(out)
    assert open(out, "rb").read() == open(w.ind, "rb").read(), "device-built --store-sig .ind differs from the host builder's"
    recs = pc.check_e2e(c, w)                      # (the context serves reads from the index it has just built)
    assert recs["mapped"].sum() > 60


def test_any_order_sort(ctx):
    """Region keys (hit.c:111-126) through the sorter's any-order levels + tie check, many long segments."""
    assert pc.check_sort_any(ctx, seed=5) >= 1
    assert pc.check_sort_any(ctx, seed=6, sizes=tuple([9000 + 911 * i for i in range(120)])) >= 1


def test_exact_sort_multi_workgroup(ctx):
    """Segments beyond the LDS classes (rh_bigsort.hip): every key kind, several levels, one segment of more than 2^20
    records, and many segments at once (thousands of ranges per level)."""
    pc.check_sort_big(ctx, seed=11, sizes=(30000, 70000, 9000, 8193, 250000, 40000, 100000, 16385))
    pc.check_sort_big(ctx, seed=12, sizes=(1_200_000, 300_000), kinds=(0, 1))
    pc.check_sort_big(ctx, seed=13, sizes=(1_100_000,), kinds=(2,))
    pc.check_sort_big(ctx, seed=15, sizes=(400_000, 90_000, 20_000), kinds=(5,))
    pc.check_sort_big(ctx, seed=14, sizes=tuple([30000 + 17 * i for i in range(300)]), kinds=(0, 1, 0, 0, 3))


def test_exact_sort_block_parallel_walk(ctx):
    """rh_bigsort.hip, k_bs_pw_count / k_bs_pw_walk: backtrack-candidate keys at the sizes of a human-scale chunk (10^5 candidates a read,
    half of them chains of one anchor = the lowest score), one and two levels, up to 64 and up to 256 regions with holes, hundreds of
    ranges at once - the exact permutation of radix_sort_128x from snapshots of 64 cycles followed at once + all blocks walked at once."""
    pc.check_sort_big(ctx, seed=21, sizes=(110_000, 60_000, 240_000, 9_000, 30_000), kinds=(6,))
    pc.check_sort_big(ctx, seed=22, sizes=(150_000, 80_000, 33_000), kinds=(7, 8, 6))
    pc.check_sort_big(ctx, seed=23, sizes=tuple([40_000 + 173 * i for i in range(200)]), kinds=(6, 6, 7, 8, 1))


def test_index_built_on_device(product_lib, tmp_path):
    """rh_index_build_device = the host builder (pinned to the reference by tests/test_oracle.py): keys, counts, position
    lists, mid_occ; targets with gaps / lower case / shorter than a seed; then a 6 Mbp reference (thousands of filter blocks)."""
    checked, n = pc.check_device_index(product_lib, tmp_path)
    assert checked > 1000 and n > 10000
    pc.check_device_index(product_lib, tmp_path / "b", preset="fast", chrom_len=3_000_000, n_chrom=2, with_gaps=True, seed=9)
    pc.check_device_index(product_lib, tmp_path / "c", preset="faster", chrom_len=3_000_000, n_chrom=2, with_gaps=True, seed=10)   # minimisers (w = 3)


def test_device_index_maps_like_uploaded_index(make_workload, product_lib):
    """The table filled on the device serves reads exactly like the host-built, uploaded one (PAF vs the oracle)."""
    from rawhash_amd.api import Index
    w = make_workload(n_reads=300, n_samples=24_000, chrom_len=600_000)
    c = Context(0, lib=product_lib)
    dev = Index.build_device(c, w.fasta, w.model, w.opts, n_threads=8)
    assert dev.n_keys == w.index.n_keys
    recs = c.map_batch(w.opts, w.reads)
    got = [strip_mt(x) for x in paf_lines(dev, recs, w.reads.names)]
    assert got == w.oracle_paf()
    c.close()


def test_chain_adversarial(ctx, wl):
    """few reads -> workgroup walk in LDS; thousands -> the 64-candidates-per-round wave kernel"""
    for n_reads, max_n, seed in ((48, 900, 1), (2300, 260, 2), (2100, 60, 3)):
        n_an, n_ch, n_u = pc.check_chain_synthetic(ctx, wl, seed=seed, n_reads=n_reads, max_n=max_n)
        assert n_ch > 0 and n_u > 0


def test_end_to_end_paf(ctx, wl):
    recs = pc.check_e2e(ctx, wl)
    assert recs["mapped"].sum() > 0 and (recs["mapped"] == 0).sum() > 0


@pytest.mark.parametrize("preset", ["fast", "faster", "viral"])
def test_presets(make_workload, gpu_ctx_factory, preset):
    w = make_workload(preset=preset, n_reads=200, n_samples=24_000, chrom_len=600_000)
    c = gpu_ctx_factory()
    c.upload(w.index)
    pc.check_e2e(c, w)


def test_ragged_and_empty_reads(ctx, wl):
    """empty read, read shorter than one chunk, read shorter than min_events worth of signal, all-filtered read."""
    from rawhash_amd.api import Reads
    base = wl.reads
    parts = [np.zeros(0, np.int16), base.samples[:700].copy(), base.samples[:3999].copy(), base.samples[:4001].copy(),
             np.full(5000, 30000, np.int16), base.samples[24_000:24_000 + 9000].copy()]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    rd = Reads(np.concatenate(parts), off, [f"edge{i}" for i in range(len(parts))], base.cal_offset[0], base.cal_scale[0])
    pc.check_e2e(ctx, wl, rd)


def test_two_batches_in_flight(ctx, wl):
    """rh_map_submit / rh_map_wait: two batches in flight on one context return what rh_map_batch returns for each."""
    idx = list(range(len(wl.reads)))
    ra, rb = wl.reads.subset(idx[:150]), wl.reads.subset(idx[150:])
    want_a, want_b = ctx.map_batch(wl.opts, ra).copy(), ctx.map_batch(wl.opts, rb).copy()
    for _ in range(3):
        ha = ctx.map_submit(wl.opts, ra)
        hb = ctx.map_submit(wl.opts, rb)
        with pytest.raises(Exception):
            ctx.map_submit(wl.opts, ra)                      # a third one is refused
        assert np.array_equal(ctx.map_wait(ha), want_a) and np.array_equal(ctx.map_wait(hb), want_b)


def test_index_bcast_between_contexts(product_lib, wl):
    """rh_index_bcast: the resident index of one context copied device-to-device into another (here on the same GPU)."""
    import ctypes as C
    a, b = Context(0, lib=product_lib), Context(0, lib=product_lib)
    a.upload(wl.index)
    arr = (C.c_void_p * 2)(a.h, b.h)
    assert product_lib.rh_index_bcast(arr, 2) == 0
    assert np.array_equal(a.map_batch(wl.opts, wl.reads), b.map_batch(wl.opts, wl.reads))
    a.close(); b.close()


def test_rccl_in_process_selftest(product_lib):
    """The C/C++ host's own RCCL call site (rh_index_bcast -> ncclBroadcast through dlopen'ed librccl.so.1): the library loads, every
    entry point resolves and a one-rank communicator broadcasts in place on this GPU - what a one-GPU box can run of that path."""
    c = Context(0, lib=product_lib)
    assert product_lib.rh_rccl_selftest(c.h) == 0, product_lib.rh_last_error().decode()
    c.close()


def test_index_bcast_across_devices(product_lib, wl):
    """rh_index_bcast between DISTINCT GPUs (hipMemcpyPeerAsync, doubling tree): needs >= 2 visible devices."""
    import ctypes as C
    n_dev = product_lib.rh_device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} visible GPU(s): the peer-to-peer broadcast needs two")
    n = min(n_dev, 4)
    ctxs = [Context(i, lib=product_lib) for i in range(n)]
    ctxs[0].upload(wl.index)
    arr = (C.c_void_p * n)(*[c.h for c in ctxs])
    assert product_lib.rh_index_bcast(arr, n) == 0
    assert product_lib.rh_index_bcast_path() == 1, "distinct devices: the broadcast is expected to go through RCCL (ncclBroadcast)"
    want = ctxs[0].map_batch(wl.opts, wl.reads)
    for c in ctxs[1:]:
        assert np.array_equal(c.map_batch(wl.opts, wl.reads), want)
    for c in ctxs:
        c.close()


NCCL_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from rawhash_amd.api import Context, Index, MapOptions, SynthWorkload
from rawhash_amd.dist import shard_bounds, replicate_index, gather_records
rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{lr}"))
d = sys.argv[2]
wl = SynthWorkload(chrom_len=300_000, n_chrom=2, n_samples=12_000, junk_per_1024=150, noise_q24=150_000)
opts = MapOptions("sensitive")
ctx = Context(lr)
index = None
if rank == 0:
    fasta, model = wl.write_reference(d)
    index = Index.build(fasta, model, opts, out_ind=os.path.join(d, "ref.ind"))
    opts.update(index)
dist.barrier()
keep = replicate_index(ctx, opts, index, device=f"cuda:{lr}")      # RH_BCAST_PIECE_BYTES (environment): many pieces
n = 64
lo, hi = shard_bounds(n, rank, world)
recs = ctx.map_batch(opts, wl.reads(os.path.join(d, "model.txt"), lo, hi - lo))
allr = gather_records(recs, lo)
if rank == 0:
    single = ctx.map_batch(opts, wl.reads(os.path.join(d, "model.txt"), 0, n))
    assert np.array_equal(allr, single), "sharded result differs from the single-GPU result"
    open(os.path.join(d, "ok"), "w").write("ok")
dist.barrier()
dist.destroy_process_group()
'''


def test_nccl_replicate_index_across_devices(product_lib, tmp_path):
    """The headline multi-GPU flow on real devices: one process per GPU, backend "nccl" (= RCCL), rank 0's index blob broadcast in
    pieces (rawhash_amd.dist.replicate_index) and adopted, reads sharded, records gathered = the single-GPU result."""
    import subprocess, sys
    n_dev = product_lib.rh_device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} visible GPU(s): the RCCL broadcast needs two")
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RH_BCAST_PIECE_BYTES="262144", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", "29555", str(script), root, str(tmp_path)], check=True, env=env, timeout=900)
    assert (tmp_path / "ok").exists()


def test_blow5_file_maps_from_pinned_staging(ctx, wl, product_lib, tmp_path):
    """zstd records + svb-zd signals (slow5tools' defaults) decoded straight into the page-locked staging buffer; mapping from that
    buffer gives the records of the in-memory batch."""
    from rawhash_amd.api import ReadsFile, write_blow5
    cfg = wl.wl.cfg
    p = str(tmp_path / "reads.blow5")
    write_blow5(wl.reads, p, cfg.digitisation, cfg.range, cfg.offset, records="zstd", svb_zd=True, lib=product_lib)
    f = ReadsFile(p, lib=product_lib)
    assert f.pinned and f.names == wl.reads.names
    assert np.array_equal(ctx.map_batch(wl.opts, f.batch()), ctx.map_batch(wl.opts, wl.reads))
    f.close()


def test_blow5_hand_assembled_file_maps(ctx, wl, product_lib, tmp_path):
    """The independent evidence for the BLOW5 reader on the GPU box: a file assembled byte by byte from the published format
    (tests/blow5_fixtures.py - Python's zlib / libzstd through ctypes / its own StreamVByte encoder, NOT the library's writer), zstd
    and zlib records with svb-zd signals, loaded through rh_reads_load -> rh_reads_batch -> rh_map_batch: the records of the in-memory batch."""
    import blow5_fixtures as B
    from rawhash_amd.api import ReadsFile
    cfg = wl.wl.cfg
    n = min(len(wl.reads), 96)
    rd = [(wl.reads.names[i], wl.reads.samples[int(wl.reads.offsets[i]):int(wl.reads.offsets[i + 1])], cfg.digitisation, cfg.offset, cfg.range) for i in range(n)]
    want = ctx.map_batch(wl.opts, wl.reads.subset(list(range(n))))
    zs = B.load_zstd()
    for rec_comp in ((1, 2) if zs is not None else (1,)):
        p = str(tmp_path / f"hand_{rec_comp}.blow5")
        open(p, "wb").write(B.assemble(rd, rec_comp, 1, 8, zs))
        f = ReadsFile(p, lib=product_lib)
        assert f.names == wl.reads.names[:n]
        assert np.array_equal(ctx.map_batch(wl.opts, f.batch()), want)
        f.close()


def test_batch_split_invariance(ctx, wl):
    """Mapping is per-read independent: any split of the batch gives the same records (property used by sharding)."""
    full = ctx.map_batch(wl.opts, wl.reads)
    idx = list(range(len(wl.reads)))
    a, b = idx[:137], idx[137:]
    ra = ctx.map_batch(wl.opts, wl.reads.subset(a))
    rb = ctx.map_batch(wl.opts, wl.reads.subset(b))
    drop = lambda r: r[[n for n in r.dtype.names if n != "read_idx"]]
    assert np.array_equal(drop(full[:137]), drop(ra)) and np.array_equal(drop(full[137:]), drop(rb))


def test_golden_paf(product_lib, gpu_ctx_factory, tmp_path):
    """HIP path vs PAF produced by the pinned reference build itself (tests/golden/, made by make_golden.py)."""
    import golden
    from rawhash_amd.api import Index
    for case in golden.cases():
        w = golden.build_case(case, tmp_path / case["name"], product_lib)
        c = gpu_ctx_factory()
        if case.get("gpu_only"):     # config-scale reference (144 Mbp): index built on the device
            w.index = Index.build_device(c, w.fasta, w.model, w.opts, n_threads=32)
            w.opts.update(w.index)
        else:
            c.upload(w.index)
        recs = c.map_batch(w.opts, w.reads)
        got = [strip_mt(x) for x in paf_lines(w.index, recs, w.reads.names)]
        want = golden.expected_paf(case)
        bad = [(g, x) for g, x in zip(got, want) if g != x]
        assert len(got) == len(want) and not bad, f"{case['name']}: {len(bad)} PAF lines differ, first: {bad[:1]}"
        if case["name"] == "config3_dmel_144M_rmq":     # 35 k anchors a chunk: the windows of live nodes still fit the two small LDS rings (measured: 224 / 13 / 0 / 0
            cls = c.stats()["n_rmq_class"]              # (read, chunk) pairs) - the wider classes run on the hardware in test_rmq_storage_classes_on_device
            assert cls[0] > 0 and cls[1] > 0, f"RMQ storage classes at config-3 size: {cls}"


def test_repeat_rich_golden(product_lib, gpu_ctx_factory, tmp_path):
    """Repeat-rich references (tandem repeats of 2-200 bp units, duplicated 10-100 kb blocks with 1 % divergence, runs of N;
    tests/repeat_workload.py) against the PAF the reference printed: mid_occ filtering, the tandem flag (rseed.c:105-154), rep_len, and -
    on the 52 Mbp case, whose chunks hold more than 8192 anchors - the multi-workgroup exact sorter on real tie-ridden keys.
    Indexes are built on the device from the FASTA."""
    import golden
    from rawhash_amd.api import Index
    for case in golden.repeat_cases():
        w = golden.build_repeat_case(case, tmp_path / case["name"], product_lib)
        c = gpu_ctx_factory()
        index = Index.build_device(c, w.fasta, w.model, w.opts, n_threads=32)
        w.opts.update(index)
        recs = c.map_batch(w.opts, w.reads)
        st = c.stats()
        got = [strip_mt(x) for x in paf_lines(index, recs, w.reads.names)]
        want = golden.expected_paf(case)
        bad = [(g, x) for g, x in zip(got, want) if g != x]
        assert len(got) == len(want) and not bad, f"{case['name']}: {len(bad)} PAF lines differ, first: {bad[:1]}"
        if case.get("gpu_only"):
            assert st["n_anchors"] / max(st["n_chunks"], 1) > 8192, "the large case is meant to reach the multi-workgroup sorter"


def test_config2_ecoli_scale_vs_oracle(make_workload, product_lib):
    """BASELINE.json configs[1] at its index size: 4.6 Mbp index, 2560 reads of the bench's read set (10 % unmappable: all ten
    chunk rounds with carried chains), HIP path vs the oracle on all host cores."""
    import os
    w = make_workload(n_reads=2560, n_samples=40_000, chrom_len=4_600_000, n_chrom=1, junk=102, noise=0, read_seed=3)
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    recs = c.map_batch(w.opts, w.reads)
    got = [strip_mt(x) for x in paf_lines(w.index, recs, w.reads.names)]
    want = w.oracle_paf(n_threads=os.cpu_count() or 8)
    bad = [(g, x) for g, x in zip(got, want) if g != x]
    assert not bad, f"{len(bad)} of {len(want)} PAF lines differ, first: {bad[0]}"
    assert (recs["tag_ci"] == 10).sum() > 100 and recs["mapped"].mean() > 0.85
    c.close()


@pytest.mark.timeout(1200)
def test_human_scale_index_vs_reference(product_lib, tmp_path):
    """BASELINE.json configs[3] at its index size: 3.1 Gbp in 24 targets, index built on the device; 3 072 reads (round 6; 512 until then) of the bench's
    read set (preset fast) mapped by the HIP path and by the CPU side reading the .ind this library wrote - the unmodified
    reference (oracle/_ref/ref_harness) where its binary travelled, else the oracle."""
    import ctypes as C
    import os
    import subprocess
    import oracle_lib as O
    from rawhash_amd.api import Index, MapOptions, SynthWorkload
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else str(tmp_path)
    wd = os.path.join(shm, f"rh_human_test_{os.getpid()}")
    os.makedirs(wd, exist_ok=True)
    try:
        cores = os.cpu_count() or 8
        wl = SynthWorkload(chrom_len=129_166_667, n_chrom=24, n_samples=40_000, junk_per_1024=102, lib=product_lib)
        opts = MapOptions("fast", lib=product_lib)
        model = os.path.join(wd, "model.txt")
        product_lib.rh_synth_write_model(C.byref(wl.cfg), model.encode())
        seqs = [wl.genome(ch, n_threads=min(cores, 64)) for ch in range(24)]
        c = Context(0, lib=product_lib)
        index = Index.build_device_seqs(c, [f"chr{i + 1}" for i in range(24)], seqs, model, opts, n_threads=min(cores, 64))
        del seqs
        opts.update(index)
        assert index.n_positions > 4_000_000_000
        reads = wl.reads(model, 0, 3072, n_threads=min(cores, 64))
        recs = c.map_batch(opts, reads)
        got = [strip_mt(x) for x in paf_lines(index, recs, reads.names)]
        index.download(c, n_threads=min(cores, 64))
        ind = os.path.join(wd, "ref.ind")
        index.write(ind)
        c.close()
        if O.have_reference():
            rhr = os.path.join(wd, "reads.rhr")
            reads.write(rhr, wl.cfg.digitisation, wl.cfg.range, wl.cfg.offset)
            out = subprocess.run([O.REF_HARNESS, "map", "fast", ind, rhr, str(min(cores, 32))], check=True, capture_output=True, text=True).stdout
            want = [O.strip_mt(x) for x in out.splitlines()]
        else:
            oix = O.OracleIndex(ind)
            _, mo = O.preset("fast")
            O.lib().ro_mapopt_update(C.byref(mo), oix.h)
            want = [O.strip_mt(x) for x in O.paf_lines(oix, O.map_batch(oix, mo, reads.batch(), n_threads=cores), reads.names)]
        bad = [(g, x) for g, x in zip(got, want) if g != x]
        assert len(got) == len(want) == 3072 and not bad, f"{len(bad)} PAF lines differ, first: {bad[:1]}"
        assert recs["mapped"].mean() > 0.85
    finally:
        import shutil
        shutil.rmtree(wd, ignore_errors=True)


def test_large_batch_paths(make_workload, gpu_ctx_factory):
    """6000 reads: exercises the many-reads dispatch (one read per lane walks) against the oracle on all host cores."""
    import os
    w = make_workload(n_reads=6000, n_samples=20_000, chrom_len=1_500_000, n_chrom=1, junk=120, noise=0, read_seed=31)
    c = gpu_ctx_factory()
    c.upload(w.index)
    recs = c.map_batch(w.opts, w.reads)
    got = [strip_mt(x) for x in paf_lines(w.index, recs, w.reads.names)]
    want = w.oracle_paf(n_threads=os.cpu_count() or 8)
    bad = [(g, x) for g, x in zip(got, want) if g != x]
    assert not bad, f"{len(bad)} of {len(want)} PAF lines differ, first: {bad[0]}"


def test_large_genome_many_anchors(make_workload, gpu_ctx_factory):
    """48 Mbp index: ~10 k anchors per chunk, beyond the LDS size classes of sort / backtrack (HBM fallbacks), several targets."""
    import os
    w = make_workload(n_reads=96, n_samples=16_000, chrom_len=24_000_000, n_chrom=2, junk=100, noise=100_000, read_seed=41)
    c = gpu_ctx_factory()
    c.upload(w.index)
    recs = c.map_batch(w.opts, w.reads)
    got = [strip_mt(x) for x in paf_lines(w.index, recs, w.reads.names)]
    want = w.oracle_paf(n_threads=os.cpu_count() or 8)
    assert got == want
    assert c.stats()["n_anchors"] / max(c.stats()["n_chunks"], 1) > 6000


def test_sub_batches_give_identical_records(make_workload, product_lib, monkeypatch):
    """RH_SUB_BATCHES=2: concurrent sub-batches on two streams return exactly the records of the single-stream run."""
    from rawhash_amd.api import Context
    w = make_workload(n_reads=6000, n_samples=20_000, chrom_len=1_500_000, n_chrom=1, junk=120, noise=0, read_seed=31)
    c1 = Context(0, lib=product_lib); c1.upload(w.index)
    a = c1.map_batch(w.opts, w.reads)
    c1.close()
    monkeypatch.setenv("RH_SUB_BATCHES", "2")
    c2 = Context(0, lib=product_lib); c2.upload(w.index)
    b = c2.map_batch(w.opts, w.reads)
    c2.close()
    assert np.array_equal(a, b)


@pytest.mark.timeout(600)
def test_concurrent_sub_batches_repeatable(product_lib, monkeypatch, tmp_path):
    """The bench's shape at a third of its size, 3 sub-batches on 3 streams, 12 calls: every call returns the same records
    (a barrier missing in the sorter once made roughly one call in thirty spin forever under exactly this load)."""
    import os
    from rawhash_amd import Context, Index, MapOptions, SynthWorkload
    monkeypatch.setenv("RH_SUB_BATCHES", "3")
    wl = SynthWorkload(chrom_len=4_600_000, n_chrom=1, n_samples=40_000, junk_per_1024=102)
    opts = MapOptions("sensitive")
    wl.write_reference(str(tmp_path))
    fasta, model = os.path.join(str(tmp_path), "ref.fa"), os.path.join(str(tmp_path), "model.txt")
    index = Index.build(fasta, model, opts, out_ind=None, n_threads=os.cpu_count() or 8)
    opts.update(index)
    c = Context(0, lib=product_lib)
    c.upload(index)
    batch = wl.reads_device(c, model, 0, 36_000)
    first = c.map_batch(opts, batch).copy()
    assert int(first["mapped"].sum()) > 30_000
    for _ in range(11):
        assert np.array_equal(first, c.map_batch(opts, batch))
    c.close()


def test_batch_sliced_when_arenas_do_not_fit(make_workload, product_lib, monkeypatch):
    """RH_ARENA_MAX_BYTES far below what the batch needs: rh_map_batch maps it in halving slices and returns the same records."""
    w = make_workload(n_reads=48)
    monkeypatch.setenv("RH_SUB_BATCHES", "1")
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    whole = [strip_mt(x) for x in paf_lines(w.index, c.map_batch(w.opts, w.reads), w.reads.names)]
    c.close()
    monkeypatch.setenv("RH_ARENA_MAX_BYTES", str(3 << 20))       # the 128 B/anchor scratch of 48 reads x ~3 k anchors is ~18 MB
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    sliced = [strip_mt(x) for x in paf_lines(w.index, c.map_batch(w.opts, w.reads), w.reads.names)]
    again = [strip_mt(x) for x in paf_lines(w.index, c.map_batch(w.opts, w.reads), w.reads.names)]   # second call starts from the remembered slice size
    c.close()
    assert sliced == whole and again == whole and len(whole) == 48


def test_batch_in_consecutive_calls_when_rows_do_not_fit(make_workload, product_lib, monkeypatch):
    """A batch whose event / seeding rows exceed the device (a million reads) is mapped in consecutive calls: forced here with a
    limit of 13 reads per call on a 48-read batch, with and without sub-batches."""
    w = make_workload(n_reads=48)
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    whole = c.map_batch(w.opts, w.reads)
    c.close()
    monkeypatch.setenv("RH_CALL_READS_MAX", "13")
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    cut = c.map_batch(w.opts, w.reads)
    c.close()
    assert np.array_equal(whole, cut) and len(cut) == 48


def test_rawsamble_all_vs_all_golden(product_lib, tmp_path):
    """BASELINE.json configs[4] (Rawsamble) in the small: signal-target index built on the GPU = the reference's .ind byte for
    byte, all-vs-all overlaps = the reference's PAF (both presets of tests/golden/ava_cases.json)."""
    import golden
    import parity_checks as pc
    for case in golden.ava_cases():
        pc.check_ava(product_lib, case, tmp_path / case["name"], oracle_threads=os.cpu_count() or 8)


def test_rawsamble_scale_vs_oracle(product_lib, tmp_path):
    """All-vs-all at depth: 3000 reads of 3000 bases over a 300 kbp genome (30x), index built on the GPU, every overlap
    record against the oracle (which loads the .ind the GPU wrote)."""
    from conftest import AvaWorkload
    from rawhash_amd.api import Index
    w = AvaWorkload(tmp_path, product_lib, preset="ava", chrom_len=300_000, n_samples=27_000, n_reads=3000, junk=50, noise=150_000, read_seed=23)
    c = Context(0, lib=product_lib)
    ix = Index.build_signals_device(c, w.reads, w.model, w.opts)
    ix.download(c)
    ind = str(tmp_path / "dev.ind")
    ix.write(ind)
    w.opts.update(ix)
    recs, off = c.map_batch_multi(w.opts, w.reads, ix, max_records=400 * len(w.reads))
    got = [strip_mt(x) for x in paf_lines(ix, recs, w.reads.names)]
    want = w.oracle_paf(ind, n_threads=os.cpu_count() or 8)
    bad = [(g, x) for g, x in zip(got, want) if g != x]
    assert len(got) == len(want) and not bad, f"{len(bad)} of {len(want)} PAF lines differ, first: {bad[:1]}"
    assert len(got) > len(w.reads)                                # (reads with several overlaps)
    c.close()


def test_rawsamble_long_reads_vs_oracle(product_lib, tmp_path):
    """Whole-read rounds far beyond a chunk: 600 k-sample reads (66 k bases, ~62 k events each; the prefilter's tile counts no
    longer fit LDS, the seeds of a read take 30 probe tiles), ragged, all-vs-all against the oracle."""
    from conftest import AvaWorkload
    from rawhash_amd.api import Index
    w = AvaWorkload(tmp_path, product_lib, preset="ava", chrom_len=200_000, n_samples=600_000, n_reads=40, junk=50, noise=150_000, read_seed=25, ragged=True)
    c = Context(0, lib=product_lib)
    ix = Index.build_signals_device(c, w.reads, w.model, w.opts)
    ix.download(c)
    ind = str(tmp_path / "dev.ind")
    ix.write(ind)
    w.opts.update(ix)
    recs, off = c.map_batch_multi(w.opts, w.reads, ix)
    got = [strip_mt(x) for x in paf_lines(ix, recs, w.reads.names)]
    want = w.oracle_paf(ind, n_threads=os.cpu_count() or 8)
    bad = [(g, x) for g, x in zip(got, want) if g != x]
    assert len(got) == len(want) and not bad, f"{len(bad)} of {len(want)} PAF lines differ, first: {bad[:1]}"
    assert sum(1 for x in got if x.split("\t")[4] != "*") > 10
    c.close()


def test_whole_read_rounds_golden(product_lib, tmp_path):
    """RI_M_NO_ADAPTIVE on a sequence index (`--disable-adaptive`): the reference's PAF."""
    import golden
    case = [c for c in golden.cases() if c.get("no_adaptive")][0]
    w = golden.build_case(case, tmp_path, product_lib)
    c = Context(0, lib=product_lib)
    c.upload(w.index)
    recs = c.map_batch(w.opts, w.reads)
    got = [strip_mt(x) for x in paf_lines(w.index, recs, w.reads.names)]
    assert got == golden.expected_paf(case)
    c.close()

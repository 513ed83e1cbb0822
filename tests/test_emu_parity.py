"""CPU-side checks of the KERNEL SOURCES (rh_kernels.hip compiled against the SIMT emulator in tests/emu) against the
oracle.  These are logic tests of the device code; the real-hardware parity tests are in test_gpu_parity.py."""
import pytest

import parity_checks as pc
from rawhash_amd.api import Context


@pytest.fixture(scope="module")
def wl(make_workload, emu_lib):
    return make_workload(lib=emu_lib, n_reads=24, n_samples=12_000)


@pytest.fixture(scope="module")
def ctx(emu_lib, wl):
    c = Context(0, lib=emu_lib)
    c.upload(wl.index)
    yield c
    c.close()


def test_events_bit_exact(ctx, wl):
    pc.check_events(ctx, wl, chunks=(0, 2))
    pc.check_events_variants(ctx, wl, chunks=(1,))


def test_stage_chain(ctx, wl):
    n_anchors, n_chained = pc.check_stages(ctx, wl)
    assert n_anchors > 0 and n_chained > 0


def test_exact_sort(ctx):
    pc.check_sort(ctx, seed=1, n_seg=24)
    pc.check_sort(ctx, seed=3, n_seg=330, tiny=True)


def test_sort_one_word_records(ctx, emu_lib_smallcaps):
    """The anchor sort on 8-byte records: sort_fast (segments without equal keys: records in LDS, one bucket pass, a register network per bucket) and
    what it hands back to the general path - at the production LDS classes and at the small ones of the test build."""
    pc.check_sort_packed(ctx, seed=21)
    pc.check_sort_packed(ctx, seed=22, sizes=[20000, 9000, 3000, 12000, 700, 40000], any_order=True)
    c = Context(0, lib=emu_lib_smallcaps)
    pc.check_sort_packed(c, seed=23, sizes=[0, 1, 40, 100, 128, 129, 200, 256, 257, 300, 384, 385, 500, 640, 641, 900, 1024, 1025, 3000])
    pc.check_sort_packed(c, seed=24, sizes=[2000, 1100, 5000, 90, 700], any_order=True, lo=20, mid=3)
    c.close()


def test_exact_sort_multi_workgroup(ctx, emu_lib_smallcaps):
    """rh_bigsort.hip: segments longer than the LDS classes, at the production sizes of tiles / windows and with tiny ones
    (several levels, window refills of the token walk)."""
    pc.check_sort_big(ctx, seed=7, sizes=(20000, 12000, 9000, 8300, 8193))
    pc.check_sort_big(ctx, seed=9, sizes=(15000, 9500), kinds=(5,))
    c = Context(0, lib=emu_lib_smallcaps)
    pc.check_sort_big(c, seed=8, sizes=(5000, 3000, 7000, 2500, 1500, 6000, 9000, 1100, 4000, 2000))
    c.close()


def test_exact_sort_block_parallel_walk(ctx, emu_lib_smallcaps):
    """rh_bigsort.hip, k_bs_pw_count / k_bs_pw_walk: backtrack-candidate keys (the lowest score holds half the records), whose token walk is
    done block-parallel - pointer snapshots from 64 cycles followed at once, then every block walked in the reference's order, all at once.
    Production sizes, and a build with a tiny window / few snapshot slots / small rings (blocks walked by one lane, ranges that run out of
    slots, ring misses)."""
    pc.check_sort_big(ctx, seed=11, sizes=(30000, 50000, 12000, 9000), kinds=(6,))
    pc.check_sort_big(ctx, seed=12, sizes=(40000, 21000), kinds=(7, 8))
    c = Context(0, lib=emu_lib_smallcaps)
    pc.check_sort_big(c, seed=13, sizes=(5000, 3000, 7000, 2500, 9000, 4000), kinds=(6, 7, 8))
    pc.check_sort_big(c, seed=14, sizes=(6000, 2600, 9000, 3100), kinds=(9, 10))   # a few equal keys: exact passes only on the way to them
    c.close()


def test_exact_sort_tie_path(ctx):
    """Segments with one to three groups of equal keys among distinct ones (anchor keys, region keys): the any-order pass marks where the equal keys lie,
    the exact re-run walks only the ranges on the way to them and places the rest in any order again - result = radix_sort_128x's permutation."""
    pc.check_sort_big(ctx, seed=15, sizes=(40000, 26000, 90000, 12000), kinds=(9, 10))


def test_any_order_sort(ctx):
    """Region keys (hit.c:111-126) through the sorter's any-order levels + tie check."""
    assert pc.check_sort_any(ctx, seed=3) >= 1


def test_index_built_on_device(emu_lib, emu_lib_smallcaps, tmp_path):
    checked, n = pc.check_device_index(emu_lib, tmp_path)
    assert checked > 1000 and n > 10000
    pc.check_device_index(emu_lib, tmp_path / "b", preset="fast", chrom_len=30_000, n_chrom=2, with_gaps=False, seed=6)
    # blocks of 64 events with a 3-event warm-up: the speculative starts of the event filter are often wrong and get re-run
    pc.check_device_index(emu_lib_smallcaps, tmp_path / "c", chrom_len=20_000, n_chrom=2, seed=7)
    # minimiser indexes (w > 0, ri_sketch_min rsketch.c:55-141): preset faster (w = 3) with gaps; low-entropy sequence for equal minima
    pc.check_device_index(emu_lib, tmp_path / "d", preset="faster", chrom_len=40_000, n_chrom=2, seed=8)
    pc.check_device_index(emu_lib_smallcaps, tmp_path / "e", preset="faster", chrom_len=15_000, n_chrom=3, with_gaps=False, seed=9)


def test_chain_adversarial(ctx, wl):
    n_an, n_ch, n_u = pc.check_chain_synthetic(ctx, wl, seed=2, n_reads=24, max_n=600)
    assert n_ch > 0 and n_u > 0


def test_stage_regions(ctx, wl):
    checked, with_regs = pc.check_regions(ctx, wl, seed=4, n_reads=40, max_n=500)
    assert checked > 40 and with_regs > 20


RMQ_VARIANTS = [{"flag": 2}, {"flag": 2, "rmq_inner_dist": 0}, {"flag": 2, "rmq_size_cap": 40, "rmq_inner_dist": 300}, {"bw_long": 2000}, {"flag": 2, "bw_long": 1500}]


@pytest.mark.parametrize("mapopt", RMQ_VARIANTS, ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_rmq_chaining(make_workload, emu_lib, mapopt):
    """f4: mg_lchain_rmq (--rmq: krmq AVL tree reproduced operation by operation, with / without the inner tree, with a size cap
    that evicts) and the RMQ re-chaining of chains with a long bandwidth (--bw-long), stage level on adversarial anchor sets and
    end to end against the oracle (which test_oracle pins to PAF printed by the reference for each variant)."""
    w = make_workload(lib=emu_lib, n_reads=14, n_samples=12_000, mapopt=mapopt)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=6, n_reads=24, max_n=400)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)
    c.close()


CHAIN_LIMITS = [{"max_num_skips": 2, "max_chain_iter": 40}, {"max_num_skips": 25, "max_chain_iter": 5}, {"max_num_skips": 0, "max_chain_iter": 3}]


@pytest.mark.parametrize("mapopt", CHAIN_LIMITS, ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_chain_skip_and_iter_limits(make_workload, emu_lib, mapopt):
    """--max-skips / --max-iterations below the size of a small cluster: k_chain_wave's one-anchor-per-lane path then runs its generic step (skip counting, the
    clamped window, max_ii) instead of the skip-free one every preset takes (k_chain_wave<FS>, rh_chain.hip)."""
    w = make_workload(lib=emu_lib, n_reads=14, n_samples=12_000, mapopt=mapopt)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=9, n_reads=24, max_n=400)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)
    c.close()


def test_rmq_tree_storage_classes(make_workload, emu_lib_smallcaps):
    """The RMQ trees live in LDS rings where a read's window of live nodes fits (three classes: RQ_RING_S / RQ_RING / RQ_RING_BIG nodes, a launch each) and in
    the read's scratch in HBM beyond that: a build with rings of 4 / 8 / 32 nodes sends reads through all four on small inputs."""
    w = make_workload(lib=emu_lib_smallcaps, n_reads=14, n_samples=12_000, mapopt={"flag": 2})
    c = Context(0, lib=emu_lib_smallcaps)
    c.upload(w.index)
    n_an, n_ch, n_u = pc.check_chain_synthetic(c, w, seed=6, n_reads=24, max_n=400)
    assert n_ch > 0 and n_u > 0
    pc.check_e2e(c, w)
    c.close()


DTW_VARIANTS = [{"flag": 0x40}, {"flag": 0x40, "dtw_border_constraint": 0}, {"flag": 0x40, "dtw_fill_method": 0, "dtw_min_score": 5.0},
                {"flag": 0x40, "dtw_border_constraint": 0, "dtw_fill_method": 0}]


@pytest.mark.parametrize("mapopt", DTW_VARIANTS, ids=lambda m: "_".join(f"{k}{v}" for k, v in m.items()))
def test_dtw_rescoring(make_workload, emu_lib, mapopt):
    """f4: --dtw-evaluate-chains on a --store-sig index (align_chain rmap.cpp:128-208, the band / full DTW of dtw.cpp, DTW MAPQ and
    decision) end to end against the oracle, which test_oracle pins to PAF printed by the reference for these variants."""
    w = make_workload(lib=emu_lib, n_reads=16, n_samples=12_000, idxflag=0x10, mapopt=mapopt)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    recs = pc.check_e2e(c, w)
    assert recs["mapped"].sum() > 0
    st = c.stats()
    assert st["n_dtw_device"] > 0, st       # MAPQ and decision on the device (k_dtw_decide) where the host's logf cannot change the truncated MAPQ
    c.close()


def test_dtw_host_mapq_path(make_workload, emu_lib, monkeypatch):
    """... and the round trip through the host's libm that the other reads take (RH_DTW_HOST_MAPQ=1: all of them)."""
    monkeypatch.setenv("RH_DTW_HOST_MAPQ", "1")
    w = make_workload(lib=emu_lib, n_reads=16, n_samples=12_000, idxflag=0x10, mapopt={"flag": 0x40})
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    pc.check_e2e(c, w)
    st = c.stats()
    assert st["n_dtw_device"] == 0 and st["n_dtw_host"] > 0, st
    c.close()


def test_end_to_end_paf(ctx, wl):
    recs = pc.check_e2e(ctx, wl)
    assert recs["mapped"].sum() > 0 and (recs["mapped"] == 0).sum() > 0   # both outcomes exercised


def test_consumed_prefix_staging(ctx, wl):
    assert pc.check_consumed_prefix_staging(ctx, wl, seed=3) > 0


@pytest.mark.parametrize("preset", ["fast", "faster", "viral"])
def test_presets(make_workload, emu_lib, preset):
    w = make_workload(lib=emu_lib, preset=preset, n_reads=6 if preset == "viral" else 16, n_samples=12_000)   # (viral: dense index, thousands of anchors per chunk - slow under the emulator)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    pc.check_e2e(c, w)
    c.close()


def test_oversized_read_fallbacks(make_workload, emu_lib_smallcaps):
    """Same kernels built with tiny LDS caps: reads overflow into the *_big / serial-tail paths and must still match."""
    w = make_workload(lib=emu_lib_smallcaps, n_reads=12, n_samples=12_000)
    c = Context(0, lib=emu_lib_smallcaps)
    c.upload(w.index)
    pc.check_stages(c, w)
    pc.check_e2e(c, w)
    pc.check_sort(c, seed=3, n_seg=16)
    pc.check_sort(c, seed=4, n_seg=20, big=(1025, 2000, 3500, 4096, 1600, 2049))
    pc.check_chain_synthetic(c, w, seed=5, n_reads=40, max_n=500)
    c.close()


def test_whole_read_rounds_golden(emu_lib, tmp_path):
    """RI_M_NO_ADAPTIVE (`--disable-adaptive`): one round over the whole read, rows in HBM, against the reference's PAF."""
    import golden
    from rawhash_amd.api import paf_lines
    import oracle_lib as O
    case = [c for c in golden.cases() if c.get("no_adaptive")][0]
    w = golden.build_case(case, tmp_path, emu_lib)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    sub = w.reads.subset(range(20))
    recs = c.map_batch(w.opts, sub)
    got = [O.strip_mt(x) for x in paf_lines(w.index, recs, sub.names, lib=emu_lib)]
    assert got == golden.expected_paf(case)[:20]
    c.close()


def test_r10_golden(emu_lib, tmp_path):
    """`--r10` (main.cpp:396-406: 9-mers, segmentation windows 3 / 6, thresholds 6.5 / 4.0, peak height 0.2, gap scale 1.2) against the PAF the
    reference prints; the index from the device builder (4^9 model levels) must be the host builder's."""
    import golden
    from rawhash_amd.api import paf_lines, Index
    import oracle_lib as O
    case = [c for c in golden.cases() if c["name"] == "small_r10"][0]
    w = golden.build_case(case, tmp_path, emu_lib)
    assert w.opts.io.k == 9 and w.opts.mo.window_length2 == 6
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    sub = w.reads.subset(range(24))
    recs = c.map_batch(w.opts, sub)
    got = [O.strip_mt(x) for x in paf_lines(w.index, recs, sub.names, lib=emu_lib)]
    assert got == golden.expected_paf(case)[:24]
    c.close()


def test_fast5_ingest_golden(emu_lib, tmp_path):
    """rh_read_batch_t.fast5_ingest: raw -> pA as the FAST5 reader does it (float arithmetic, kept values truncated to int16,
    rsig.c:346-374), against the PAF the reference prints for that ingest."""
    import golden
    from rawhash_amd.api import paf_lines
    import oracle_lib as O
    case = [c for c in golden.cases() if c["name"] == "small_sensitive_fast5_ingest"][0]
    w = golden.build_case(case, tmp_path, emu_lib)
    c = Context(0, lib=emu_lib)
    c.upload(w.index)
    sub = w.reads.subset(range(24))
    assert sub.fast5
    recs = c.map_batch(w.opts, sub)
    got = [O.strip_mt(x) for x in paf_lines(w.index, recs, sub.names, lib=emu_lib)]
    assert got == golden.expected_paf(case)[:24]
    sub.fast5 = False                      # the same samples the SLOW5 way: different signal values (no truncation)
    ev_a, _, _ = c.events(w.opts, sub, 0)
    sub.fast5 = True
    ev_b, _, _ = c.events(w.opts, sub, 0)
    assert len(ev_a) and (len(ev_a) != len(ev_b) or (ev_a != ev_b).any())
    c.close()


def test_rawsamble_all_vs_all_golden(emu_lib, tmp_path):
    """Signal-target index built by the device kernels = the reference's .ind; all-vs-all overlaps = the reference's PAF."""
    import golden
    case = golden.ava_cases()[0]
    assert pc.check_ava(emu_lib, case, tmp_path) > len(golden.expected_paf(case)) // 2


def test_rawsamble_ragged_reads_golden(emu_lib, tmp_path, monkeypatch):
    """The same with reads of very different lengths (row strides set by the longest), an empty read and reads too short for
    min_events, under a row budget that cuts the read set into groups of a few reads (index build and mapping; the GPU suite
    runs the same case in one group)."""
    import golden
    case = [c for c in golden.ava_cases() if c["name"] == "ava_ragged"][0]
    monkeypatch.setenv("RH_WHOLE_ROWS_MAX_SAMPLES", "300000")
    pc.check_ava(emu_lib, case, tmp_path / "groups")

"""Backend-independent parity checks: `lib` is either the HIP product library (GPU tests) or the SIMT-emulator build
of the same sources (CPU tests).  The checker is always the CPU oracle (oracle/rh_oracle.c)."""
import ctypes as C

import os

import numpy as np

import oracle_lib as O
from rawhash_amd._capi import MM128, ptr
from rawhash_amd.api import Context, paf_lines, strip_mt


def check_events(ctx, wl, chunks=(0, 1, 2)):
    b = wl.reads.batch()
    n = len(wl.reads)
    for c in chunks:
        ev, off, lsig = ctx.events(wl.opts, wl.reads, c)
        cap = n * 2048 + 16
        oev = np.zeros(cap, dtype=np.float32)
        ooff = np.zeros(n + 1, dtype=np.uint64)
        olsig = np.zeros(n, dtype=np.uint32)
        assert O.lib().ro_events_batch(C.byref(wl.opts.mo), C.byref(b), c, ptr(oev), cap, ptr(ooff), ptr(olsig)) == 0
        assert np.array_equal(lsig, olsig)
        assert np.array_equal(off, ooff), f"event counts differ in chunk {c}"
        # bit-exact fp32 (the emitted seed set depends on it); tolerance = 0 ulp
        assert np.array_equal(ev.view(np.uint32), oev[: int(ooff[n])].view(np.uint32)), f"event values differ in chunk {c}"
    return ev, off


def check_events_variants(ctx, wl, chunks=(0, 2)):
    """Every chunks-per-block variant of the prefix-sum / t-statistic kernel (the launcher picks by round size; RH_TSTAT_CB pins
    it), then windows too wide for that kernel's ring (the one-block-per-chunk layout computes them)."""
    import copy, os
    try:
        for cb in ("64", "16", "8"):
            os.environ["RH_TSTAT_CB"] = cb
            check_events(ctx, wl, chunks=chunks)
    finally:
        os.environ.pop("RH_TSTAT_CB", None)
    wide = copy.copy(wl.opts)
    wide.mo = type(wl.opts.mo).from_buffer_copy(wl.opts.mo)
    wide.mo.window_length1, wide.mo.window_length2 = 5, 20
    check_events(ctx, _ReadsOnly(wide, wl.reads), chunks=chunks[:1])


class _ReadsOnly:
    """what check_events needs of a workload: options + reads"""
    def __init__(self, opts, reads):
        self.opts, self.reads = opts, reads


def check_events_odd_signals(ctx, wl, seed=0):
    """Event detection on signals with long dwells (segments of hundreds of samples: the whole-wavefront / serial branches of
    the segment sort), flat stretches, out-of-range samples (pA filter) and very short reads."""
    from rawhash_amd.api import Reads
    rng = np.random.default_rng(seed)
    dig, rng_pa, off = 8192.0, 1402.882, 6.0
    scale = rng_pa / dig
    sigs = []
    for r in range(24):
        n = int(rng.integers(200, 12000)) if r % 6 else int(rng.integers(0, 60))
        pa = np.empty(n, dtype=np.float64)
        i = 0
        while i < n:
            kind = rng.integers(0, 5)
            dwell = int(rng.integers(3, 20)) if kind < 3 else int(rng.integers(40, 900)) if kind == 3 else int(rng.integers(1, 4))
            level = rng.uniform(60, 130) if kind != 4 else rng.choice([10.0, 250.0])      # kind 4: outside (30, 200) pA
            m = min(dwell, n - i)
            pa[i:i + m] = level + rng.normal(0, 1.5 if r % 3 else 0.2, size=m)
            i += m
        sigs.append(np.round(pa / scale - off).clip(-32768, 32767).astype(np.int16))
    offs = np.zeros(len(sigs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in sigs])
    reads = Reads(np.concatenate(sigs), offs, [f"odd{r}" for r in range(len(sigs))],
                  np.full(len(sigs), off, dtype=np.float64), np.full(len(sigs), scale, dtype=np.float32))
    return check_events(ctx, _ReadsOnly(wl.opts, reads), chunks=(0, 1, 2))


def oracle_events(wl, chunk=0):
    b = wl.reads.batch()
    n = len(wl.reads)
    cap = n * 2048 + 16
    oev = np.zeros(cap, dtype=np.float32)
    ooff = np.zeros(n + 1, dtype=np.uint64)
    assert O.lib().ro_events_batch(C.byref(wl.opts.mo), C.byref(b), chunk, ptr(oev), cap, ptr(ooff), None) == 0
    return oev[: int(ooff[n])], ooff


def oracle_seeds(wl, ev, eoff):
    oix, mo = wl.oracle()
    n = len(eoff) - 1
    cap = len(ev) * 4 + 16
    sd = np.zeros(cap, dtype=MM128)
    so = np.zeros(n + 1, dtype=np.uint64)
    assert O.lib().ro_sketch_batch(oix.h, n, ptr(ev), ptr(eoff), ptr(sd), cap, ptr(so)) == 0
    return sd[: int(so[n])], so


def oracle_anchors(wl, sd, so, q_offset=None, prev=None, prev_off=None):
    oix, mo = wl.oracle()
    n = len(so) - 1
    cap = len(sd) * (mo.mid_occ + 1) + (0 if prev is None else len(prev)) + 1024
    an = np.zeros(cap, dtype=MM128)
    ao = np.zeros(n + 1, dtype=np.uint64)
    rep = np.zeros(n, dtype=np.int32)
    assert O.lib().ro_seed_batch(oix.h, C.byref(mo), n, ptr(sd), ptr(so), ptr(q_offset), ptr(prev), ptr(prev_off), ptr(an), cap, ptr(ao), ptr(rep)) == 0
    return an[: int(ao[n])], ao, rep


def oracle_chains(wl, an, ao):
    oix, mo = wl.oracle()
    n = len(ao) - 1
    cap = len(an) + 16
    ch = np.zeros(cap, dtype=MM128)
    pv = np.zeros(cap, dtype=MM128)
    co = np.zeros(n + 1, dtype=np.uint64)
    u = np.zeros(cap, dtype=np.uint64)
    uo = np.zeros(n + 1, dtype=np.uint64)
    assert O.lib().ro_chain_batch(oix.h, C.byref(mo), n, ptr(an), ptr(ao), ptr(ch), cap, ptr(co), ptr(u), cap, ptr(uo), ptr(pv)) == 0
    return ch[: int(co[n])], co, u[: int(uo[n])], uo, pv[: int(co[n])]


def check_stages(ctx, wl):
    """events -> sketch -> seed/anchors(+exact sort) -> chain, each stage fed with the oracle's output of the previous one."""
    ev, eoff = oracle_events(wl, 0)
    sd, so = ctx.sketch(ev, eoff)
    osd, oso = oracle_seeds(wl, ev, eoff)
    assert np.array_equal(so, oso) and np.array_equal(sd, osd), "seeds differ"
    n = len(oso) - 1
    qoff = (np.arange(n, dtype=np.uint32) * 7) % 500
    oan, oao, orep = oracle_anchors(wl, osd, oso, qoff)
    an, ao, rep = ctx.seed(wl.opts, osd, oso, q_offset=qoff, cap=len(oan) + 1024)
    assert np.array_equal(ao, oao), "anchor counts differ"
    assert np.array_equal(rep, orep), "rep_len differs"
    assert np.array_equal(an, oan), "sorted anchors differ (exact radix_sort_128x permutation)"
    och, oco, ou, ouo, opv = oracle_chains(wl, oan, oao)
    ch, co, u, uo, pv = ctx.chain(wl.opts, oan, oao)
    assert np.array_equal(co, oco) and np.array_equal(uo, ouo), "chain counts differ"
    assert np.array_equal(u, ou) and np.array_equal(ch, och) and np.array_equal(pv, opv), "chains differ"
    # second round: carried anchors merged into the next chunk's anchor set
    ev1, eoff1 = oracle_events(wl, 1)
    osd1, oso1 = oracle_seeds(wl, ev1, eoff1)
    qoff1 = (eoff[1:] - eoff[:-1]).astype(np.uint32)
    oan1, oao1, orep1 = oracle_anchors(wl, osd1, oso1, qoff1, opv, oco)
    an1, ao1, rep1 = ctx.seed(wl.opts, osd1, oso1, q_offset=qoff1, prev=opv, prev_off=oco, cap=len(oan1) + 1024)
    assert np.array_equal(ao1, oao1) and np.array_equal(an1, oan1) and np.array_equal(rep1, orep1), "anchors with carried chain differ"
    return len(oan), len(och)


def synthetic_anchor_sets(seed=0, n_reads=64, max_n=700):
    """Adversarial anchor sets: collinear runs (long chains, every member a backtrack candidate), bundles of anchors competing
    for the same predecessor (fan-in), random clutter, equal target positions."""
    rng = np.random.default_rng(seed)
    segs, off = [], [0]
    for r in range(n_reads):
        n = int(rng.integers(0, max_n)) if r % 7 else int(rng.integers(0, 6))
        xs, ys = [], []
        while len(xs) < n:
            kind = rng.integers(0, 4)
            rev = int(rng.integers(0, 2)); rid = int(rng.integers(0, 2))
            r0 = int(rng.integers(1000, 60000)); q0 = int(rng.integers(0, 1500))
            if kind == 0:      # collinear run with jitter
                m = int(rng.integers(2, 120)); st = int(rng.integers(3, 40))
                for t in range(m):
                    xs.append((rev << 63) | (rid << 32) | (r0 + t * st + int(rng.integers(0, 3)))); ys.append(q0 + t * st + int(rng.integers(0, 3)))
            elif kind == 1:    # fan-in: many successors of one anchor
                xs.append((rev << 63) | (rid << 32) | r0); ys.append(q0)
                for t in range(int(rng.integers(2, 30))):
                    d = int(rng.integers(5, 60)); xs.append((rev << 63) | (rid << 32) | (r0 + d)); ys.append(q0 + d + int(rng.integers(-2, 3)))
            elif kind == 2:    # clutter
                for t in range(int(rng.integers(1, 40))):
                    xs.append((rev << 63) | (rid << 32) | int(rng.integers(1000, 60000))); ys.append(int(rng.integers(0, 3000)))
            else:              # two interleaved diagonals sharing target positions
                m = int(rng.integers(2, 40))
                for t in range(m):
                    xs.append((rev << 63) | (rid << 32) | (r0 + t * 11)); ys.append(q0 + t * 11)
                    xs.append((rev << 63) | (rid << 32) | (r0 + t * 11)); ys.append(q0 + 400 + t * 11)
        x = np.array(xs[:n], dtype=np.uint64); y = np.array(ys[:n], dtype=np.int64).clip(0, 1 << 20).astype(np.uint64) | (np.uint64(10 + r % 9) << np.uint64(32))
        order = np.argsort(x, kind="stable")
        seg = np.zeros(n, dtype=MM128); seg["x"] = x[order]; seg["y"] = y[order]
        segs.append(seg); off.append(off[-1] + n)
    an = np.concatenate(segs) if segs else np.zeros(0, dtype=MM128)
    ao = np.array(off, dtype=np.uint64)
    return an, ao


def check_chain_synthetic(ctx, wl, seed=0, n_reads=64, max_n=700):
    """DP + backtrack + compact_a on the adversarial anchor sets."""
    an, ao = synthetic_anchor_sets(seed, n_reads, max_n)
    och, oco, ou, ouo, opv = oracle_chains(wl, an, ao)
    ch, co, u, uo, pv = ctx.chain(wl.opts, an, ao)
    assert np.array_equal(co, oco) and np.array_equal(uo, ouo), "chain counts differ"
    assert np.array_equal(u, ou) and np.array_equal(ch, och) and np.array_equal(pv, opv), "chains differ"
    return len(an), len(och), len(ou)


def oracle_regions(wl, ch, co, u, uo, rep_len, qlen, variant=None):
    """hit.c:100-367, 502-539 on the oracle's chains: (regs (k, 18) int32, offsets).  variant: map options to override ("flag" is OR-ed in)."""
    _, mo = wl.oracle()
    for k, v in (variant or {}).items():
        if k == "flag":
            mo.flag |= v
        else:
            setattr(mo, k, v)
    n = len(co) - 1
    cap = len(u) + 16
    regs = np.zeros((cap, 18), dtype=np.int32)
    ro = np.zeros(n + 1, dtype=np.uint64)
    rl = np.ascontiguousarray(rep_len, dtype=np.int32)
    ql = np.ascontiguousarray(qlen, dtype=np.uint32)
    assert O.lib().ro_regions_batch(C.byref(mo), n, ptr(ch), ptr(co), ptr(u), ptr(uo), ptr(rl), ptr(ql), ptr(regs), cap, ptr(ro)) == 0
    return regs[: int(ro[n])], ro


REG_FIELDS = ("id", "cnt", "rid", "score", "qs", "qe", "rs", "re", "parent", "subsc", "as", "mlen", "blen", "n_sub", "score0", "mapq", "rev", "hash")
# region-stage variants: the default selection (secondaries dropped: the primaries-only kernels), secondaries kept (best_n > 0: the serial
# core's mm_select_sub + mm_sync_regs), all-vs-all (mm_select_sub skipped, rmap.cpp:353), the hard mask level of mm_set_parent (hit.c:136)
REG_VARIANTS = ({}, {"best_n": 5}, {"flag": 0x2000}, {"flag": 0x4}, {"best_n": 2, "pri_ratio": 0.8})


def check_regions(ctx, wl, seed=0, n_reads=64, max_n=700):
    """a17-a19 at stage level: region keys + exact sort (mm_gen_regs), parents / secondaries (mm_set_parent, mm_select_sub) and
    MAPQ (mm_set_mapq) of the device against the oracle, on the chunk-0 anchors of the workload's reads and on the adversarial
    anchor sets (hundreds of chains per read, many equal scores): n_cregs, the summary of creg[0] the record is built from, and
    ALL 18 FIELDS OF EVERY KEPT REGION (the order of the reference's mm_reg1_t dump), under every variant of REG_VARIANTS."""
    import copy
    ev, eoff = oracle_events(wl, 0)
    osd, oso = oracle_seeds(wl, ev, eoff)
    n = len(oso) - 1
    oan, oao, orep = oracle_anchors(wl, osd, oso, np.zeros(n, dtype=np.uint32))
    sets = [(oan, oao, orep, (eoff[1:] - eoff[:-1]).astype(np.uint32))]
    san, sao = synthetic_anchor_sets(seed, n_reads, max_n)
    rng = np.random.default_rng(seed + 1)
    sets.append((san, sao, rng.integers(0, 400, size=len(sao) - 1).astype(np.int32), rng.integers(100, 3000, size=len(sao) - 1).astype(np.uint32)))
    checked = with_regs = n_regions = 0
    for an, ao, rep, qlen in sets:
        och, oco, ou, ouo, _ = oracle_chains(wl, an, ao)
        for variant in REG_VARIANTS:
            regs, ro = oracle_regions(wl, och, oco, ou, ouo, rep, qlen, variant)
            opts = copy.copy(wl.opts)
            opts.mo = type(wl.opts.mo).from_buffer_copy(wl.opts.mo)
            for k, v in variant.items():
                if k == "flag":
                    opts.mo.flag |= v
                else:
                    setattr(opts.mo, k, v)
            got, gregs, gro = ctx.regions(opts, an, ao, rep, qlen, all_regions=True)
            for r in range(len(ao) - 1):
                k0, k1 = int(ro[r]), int(ro[r + 1])
                assert got[r, 0] == k1 - k0, f"{variant} read {r}: n_cregs {got[r, 0]} != {k1 - k0}"
                assert int(gro[r + 1]) - int(gro[r]) == k1 - k0
                if k1 > k0:
                    q = regs[k0]     # id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0, mapq, rev, hash
                    want = [q[1], q[3], q[15], q[4], q[5], q[6], q[7], q[2], q[16]]
                    assert list(got[r, 1:]) == [int(v) for v in want], f"{variant} read {r}: creg[0] {list(got[r, 1:])} != {want}"
                    g = gregs[int(gro[r]): int(gro[r + 1])]
                    if not np.array_equal(g, regs[k0:k1]):
                        bad = np.argwhere(g != regs[k0:k1])[0]
                        raise AssertionError(f"{variant} read {r}: region {bad[0]} of {k1 - k0}, field {REG_FIELDS[bad[1]]}: device {g[bad[0]].tolist()} != oracle {regs[k0 + bad[0]].tolist()}")
                    with_regs += 1
                    n_regions += k1 - k0
                checked += 1
    assert n_regions > with_regs, "no read with more than one kept region: the per-region comparison did not see a secondary or a second primary"
    return checked, with_regs


def check_sort(ctx, seed=0, n_seg=40, big=(), tiny=False):
    """Exact unstable-permutation emulation of radix_sort_128x, with heavy key ties in every byte position.
    tiny: hundreds of segments of 0 .. 40 records (one lane per segment up to 32), every key kind."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(0, 900, size=n_seg)
    sizes[:8] = [0, 1, 64, 65, 4096, 4097, 3000, 5500]
    if tiny:
        sizes = rng.integers(0, 41, size=n_seg)
        sizes[:10] = [2, 16, 17, 31, 32, 33, 1, 0, 3, 15]
    if len(big):
        sizes[8:8 + len(big)] = big             # beyond the LDS classes: the workgroup sorter on HBM scratch, all key kinds
    off = np.zeros(n_seg + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    tot = int(off[-1])
    a = np.zeros(tot, dtype=MM128)
    for s in range(n_seg):
        b, e = int(off[s]), int(off[s + 1])
        mode = s % 7
        if mode == 6:     # anchor-like keys with one to four pairs / triples of equal positions (the pop-order short cut of the sorter)
            x = (rng.integers(0, 2, size=e - b, dtype=np.uint64) << np.uint64(63)) | rng.integers(0, 1 << 23, size=e - b, dtype=np.uint64)
            if e - b > 10:
                for _ in range(int(rng.integers(1, 5))):
                    j = rng.integers(0, e - b, size=int(rng.integers(2, 4)))
                    x[j] = x[j[0]]
        elif mode == 4:     # anchor-like keys: strand bit | small rid | 23-bit position, a few duplicated positions
            x = (rng.integers(0, 2, size=e - b, dtype=np.uint64) << np.uint64(63)) | (rng.integers(0, 3, size=e - b, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 23, size=e - b, dtype=np.uint64)
            if e - b > 10:
                x[rng.integers(0, e - b, size=(e - b) // 10)] = x[rng.integers(0, e - b, size=(e - b) // 10)]
        elif mode == 5:   # unique anchor-like keys (fast path result must already be exact)
            x = (rng.integers(0, 2, size=e - b, dtype=np.uint64) << np.uint64(63)) | rng.permutation(1 << 20)[: e - b].astype(np.uint64) * np.uint64(5)
        elif mode == 0:
            x = rng.integers(0, 1 << 62, size=e - b, dtype=np.uint64)
        elif mode == 1:
            x = rng.integers(0, 40, size=e - b, dtype=np.uint64)                      # small scores: many ties
        elif mode == 2:
            x = (rng.integers(0, 2, size=e - b, dtype=np.uint64) << np.uint64(63)) | rng.integers(0, 3000, size=e - b, dtype=np.uint64)
        else:
            x = rng.integers(0, 6, size=e - b, dtype=np.uint64) << np.uint64(8 * int(rng.integers(0, 8)))
        a["x"][b:e] = x
    a["y"] = np.arange(tot, dtype=np.uint64)
    want = a.copy()
    assert O.lib().ro_sort128x_batch(n_seg, ptr(want), ptr(off)) == 0
    got = ctx.sort128x(a, off)
    assert np.array_equal(got, want)
    for s in range(n_seg):   # property: it IS a sort
        seg = got["x"][int(off[s]):int(off[s + 1])]
        assert np.all(seg[:-1] <= seg[1:])


def check_sort_big(ctx, seed=0, sizes=(20000, 33000, 9000), kinds=(0, 1, 2, 3, 4)):
    """Segments beyond the LDS classes (the multi-workgroup level-by-level sorter): anchor-like keys over many targets with
    duplicated positions, chain-score keys with one dominant value, random 64-bit keys, one hot byte, all keys equal."""
    rng = np.random.default_rng(seed)
    segs = []
    for i, n in enumerate(sizes):
        kind = kinds[i % len(kinds)]
        if kind == 0:      # strand | 24 targets | position, 5 % duplicated keys (a seed repeated inside the chunk)
            x = (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)) | (rng.integers(0, 24, size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 27, size=n, dtype=np.uint64)
            j = rng.integers(0, n, size=n // 20)
            x[rng.integers(0, n, size=n // 20)] = x[j]
        elif kind == 1:    # backtrack candidates: scores, nearly all equal to the span
            x = np.full(n, 13, dtype=np.uint64)
            j = rng.integers(0, n, size=n // 12)
            x[j] = rng.integers(14, 400, size=len(j), dtype=np.uint64)
        elif kind == 2:
            x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        elif kind == 5:    # chain scores as the candidate sort sees them: ~100 values of one byte (65 .. 128 regions with holes), many equal keys
            x = (rng.integers(20, 20 + int(rng.integers(70, 128)), size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 10, size=n, dtype=np.uint64)
        elif kind in (6, 7, 8):    # backtrack candidates at human scale: the chains of one anchor (lowest score) are 30 - 60 % of the records, the rest
            # spread over tens (6), up to 200 (7: more than 64 regions) or 300 values (8: two levels) - the block-parallel walk of rh_bigsort.hip
            lone, hi = {6: (0.5, 60), 7: (0.3, 215), 8: (0.6, 315)}[kind]
            x = np.full(n, 15, dtype=np.uint64)
            m = rng.random(n) >= lone
            x[m] = rng.integers(16, hi, size=int(m.sum()), dtype=np.uint64)
            if i % 2:          # skewed scores: most chains have two anchors
                x[m & (rng.random(n) < 0.5)] = 17
        elif kind in (9, 10):    # one to three pairs / triples of equal keys among otherwise distinct ones: the any-order pass + the exact re-run that
            # takes its exact passes only on the way to the equal keys (rh_sort_job::tie_path); 9 anchor keys, 10 region keys (score << 32 | hash)
            if kind == 9:
                x = (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)) | (rng.integers(0, 24, size=n, dtype=np.uint64) << np.uint64(32)) | rng.permutation(np.unique(rng.integers(0, 1 << 27, size=2 * n + 64, dtype=np.uint64)))[:n]
            else:
                x = (rng.integers(40, 200, size=n, dtype=np.uint64) << np.uint64(32)) | rng.permutation(np.unique(rng.integers(0, 1 << 30, size=2 * n + 64, dtype=np.uint64)))[:n]
            for _ in range(int(rng.integers(1, 4))):
                j = rng.integers(0, n, size=int(rng.integers(2, 4)))
                x[j] = x[j[0]]
        elif kind == 3:    # three values of one byte, the rest equal
            x = (rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(8 * int(rng.integers(1, 8)))) | np.uint64(7)
        else:
            x = np.full(n, 0x0123456789ABCDEF, dtype=np.uint64)
        segs.append(x)
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in segs])
    a = np.zeros(int(off[-1]), dtype=MM128)
    a["x"] = np.concatenate(segs)
    a["y"] = np.arange(len(a), dtype=np.uint64)
    want = a.copy()
    assert O.lib().ro_sort128x_batch(len(sizes), ptr(want), ptr(off)) == 0
    got = ctx.sort128x(a, off)
    assert np.array_equal(got["x"], want["x"]), "not sorted like the reference"
    bad = np.nonzero(got["y"] != want["y"])[0]
    assert len(bad) == 0, f"{len(bad)} records out of the reference's order, first at {bad[:5]}"


def check_sort_packed(ctx, seed=0, sizes=None, any_order=False, lo=27, mid=5):
    """The anchor sort on ONE-WORD records (the mapping path's anchor format; rh_sort.hip: sort_fast takes the segments without equal keys, the
    general LDS path the others): anchor keys strand | target | position, segment sizes around every LDS class boundary, and per segment one of
    - distinct keys spread over a chromosome (the usual strand x target bucket), - distinct keys on ONE strand and target, - a few pairs / triples of
    equal keys, - keys crowded into a narrow window (buckets of more than 16 records: not for the tie-free path), - many equal keys, - two values only.
    Result = radix_sort_128x's permutation (oracle)."""
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = [0, 1, 2, 33, 64, 65, 100, 255, 256, 257, 511, 512, 513, 700, 1500, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 5000, 5632, 6000, 8191, 8192]
    segs = []
    for i, n in enumerate(sizes):
        kind = i % 6
        strand = rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)
        tmax = min(24, 1 << mid)
        tgt = rng.integers(0, tmax, size=n, dtype=np.uint64) << np.uint64(32)
        pos = rng.permutation(np.unique(rng.integers(0, 1 << lo, size=2 * n + 64, dtype=np.uint64)))[:n]
        if kind == 0:
            x = strand | tgt | pos
        elif kind == 1:    # one strand and target, and - a mapped read's anchors at its locus - up to 400 distinct positions within a window of 600 among them
            x = (np.uint64(1) << np.uint64(63)) | (np.uint64(3 % tmax) << np.uint64(32)) | pos
            nc = min(n // 3, 400)
            if nc > 20:
                w0 = int(rng.integers(0, (1 << lo) - 700))
                x = np.unique(np.concatenate([x[nc:], (np.uint64(1) << np.uint64(63)) | (np.uint64(3 % tmax) << np.uint64(32)) | (np.uint64(w0) + rng.permutation(600).astype(np.uint64)[:nc])]))
                while len(x) < n:
                    x = np.unique(np.concatenate([x, (np.uint64(1) << np.uint64(63)) | (np.uint64(3 % tmax) << np.uint64(32)) | rng.integers(0, 1 << lo, size=n - len(x), dtype=np.uint64)]))
                x = rng.permutation(x)
        elif kind == 2:
            x = (np.uint64(2 % tmax) << np.uint64(32)) | pos
            for _ in range(int(rng.integers(1, 4))):
                if n >= 2:
                    j = rng.integers(0, n, size=int(rng.integers(2, 4)))
                    x[j] = x[j[0]]
        elif kind == 3:    # crowded: most keys inside a window of n / 8 positions, a few far away
            x = (np.uint64(5 % tmax) << np.uint64(32)) | (np.uint64(1 << (lo - 1)) + rng.permutation(max(n, 1) * 2).astype(np.uint64)[:n])
            if n > 40:
                x[rng.integers(0, n, size=3)] = (np.uint64(5 % tmax) << np.uint64(32)) | rng.integers(0, 1 << lo, size=3, dtype=np.uint64)
        elif kind == 4:
            x = strand | (rng.integers(0, min(3, tmax), size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, max(n // 3, 2), size=n, dtype=np.uint64)
        else:
            x = (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)) | np.uint64(77)
        segs.append(x.astype(np.uint64))
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in segs])
    a = np.zeros(int(off[-1]), dtype=MM128)
    a["x"] = np.concatenate(segs) if len(a) else np.zeros(0, dtype=np.uint64)
    a["y"] = np.arange(len(a), dtype=np.uint64)
    want = a.copy()
    assert O.lib().ro_sort128x_batch(len(sizes), ptr(want), ptr(off)) == 0
    got = ctx.sort128x_packed(a, off, lo, mid, any_order=any_order)
    assert np.array_equal(got["x"], want["x"]), "not sorted like the reference"
    bad = np.nonzero(got["y"] != want["y"])[0]
    seg_of = np.searchsorted(off, bad[:5], side="right") - 1 if len(bad) else []
    assert len(bad) == 0, f"{len(bad)} records out of the reference's order, first at {bad[:5]} (segments {list(seg_of)}, sizes {[sizes[int(q)] for q in seg_of]})"


def check_sort_any(ctx, seed=0, sizes=(20000, 33000, 9000, 500, 70000, 12000)):
    """The any-order path of the segment sorter (region keys: score << 32 | count ^ hash): long segments without equal keys come
    out in the one sorted order there is (= the reference's); long segments with equal keys are flagged for the exact re-run;
    short ones go through the LDS block sorter exactly as before."""
    rng = np.random.default_rng(seed)
    segs, tied = [], []
    for i, n in enumerate(sizes):
        score = rng.integers(40, 900 if i % 2 else 200, size=n, dtype=np.uint64)
        x = (score << np.uint64(32)) | rng.integers(0, 1 << 32, size=n, dtype=np.uint64)
        x = np.unique(x)
        while len(x) < n:
            x = np.unique(np.concatenate([x, (rng.integers(40, 200, size=n - len(x), dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, size=n - len(x), dtype=np.uint64)]))
        x = rng.permutation(x)
        t = i % 3 == 1
        if t:
            j = rng.integers(0, n, size=3)
            x[j[1:]] = x[j[0]]
            t = len(np.unique(x)) < n
        segs.append(x); tied.append(t)
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in segs])
    a = np.zeros(int(off[-1]), dtype=MM128)
    a["x"] = np.concatenate(segs)
    a["y"] = np.arange(len(a), dtype=np.uint64)
    want = a.copy()
    assert O.lib().ro_sort128x_batch(len(sizes), ptr(want), ptr(off)) == 0
    got, ties = ctx.sort128x_any(a, off)
    lds_max = 8192
    for s, n in enumerate(sizes):
        b, e = int(off[s]), int(off[s + 1])
        assert np.array_equal(got["x"][b:e], want["x"][b:e]), f"segment {s}: not sorted"
        if n > lds_max:
            assert bool(ties[s]) == tied[s], f"segment {s}: tie flag {ties[s]}, equal keys {tied[s]}"
        if not tied[s]:
            assert np.array_equal(got["y"][b:e], want["y"][b:e]), f"segment {s}: records differ from the reference's"
    return int(ties.sum())


def check_device_index(lib, tmpdir, preset="sensitive", chrom_len=60_000, n_chrom=3, with_gaps=True, seed=5):
    """Index built on the device (rh_index_build_device) = index built on the host (which tests/test_oracle.py pins to the
    reference): same keys, same occurrence counts, same position lists in the same order, same mid_occ; targets with runs of
    ambiguous bases, lower case and a target shorter than one seed."""
    import os
    from rawhash_amd.api import Context, Index, MapOptions, SynthWorkload
    rng = np.random.default_rng(seed)
    wl = SynthWorkload(chrom_len=chrom_len, n_chrom=n_chrom, n_samples=8000, lib=lib)
    fasta0, model = wl.write_reference(str(tmpdir))
    seqs = [wl.genome(c).copy() for c in range(n_chrom)]
    names = [f"chr{c + 1}" for c in range(n_chrom)]
    if with_gaps:
        s = seqs[0]
        s[:37] = ord("N")                                  # leading gap
        for _ in range(12):
            a = int(rng.integers(100, chrom_len - 400)); s[a:a + int(rng.integers(1, 300))] = ord("N")
        s[chrom_len // 2] = ord("R")                       # single ambiguity code
        s[-5:] = ord("n")
        low = seqs[1][1000:5000]; seqs[1][1000:5000] = low + 32   # lower case is valid
        seqs.append(np.frombuffer(b"ACGTACG", dtype=np.uint8).copy()); names.append("tiny")
        seqs.append(np.frombuffer(b"NNNNNNNNNNNNNNNNNNNNACGTTGCANNNNNNNN", dtype=np.uint8).copy()); names.append("mostly_gap")
    fasta = os.path.join(str(tmpdir), "ref_gaps.fa")
    with open(fasta, "w") as f:
        for nm, sq in zip(names, seqs):
            f.write(f">{nm}\n")
            t = sq.tobytes().decode()
            for i in range(0, len(t), 80):
                f.write(t[i:i + 80] + "\n")
    opts = MapOptions(preset, lib=lib)
    host = Index.build(fasta, model, opts, n_threads=4, lib=lib)
    ctx = Context(0, lib=lib)
    dev = Index.build_device_seqs(ctx, names, seqs, model, opts, n_threads=4)
    o1, o2 = MapOptions(preset, lib=lib).update(host), MapOptions(preset, lib=lib).update(dev)
    assert o1.mo.mid_occ == o2.mo.mid_occ, (o1.mo.mid_occ, o2.mo.mid_occ)
    assert dev.n_keys == host.n_keys and dev.n_positions == host.n_positions, (dev.n_keys, host.n_keys, dev.n_positions, host.n_positions)
    assert [dev.seq_name(i) for i in range(dev.n_seq)] == names and [dev.seq_len(i) for i in range(dev.n_seq)] == [len(x) for x in seqs]
    dev.download(ctx, n_threads=4)
    hk = np.ctypeslib.as_array  # noqa
    # every key of the host index, with its positions in order
    import ctypes as C2
    n = host.n_keys
    lh = lib
    checked = 0
    # walk the host keys through rh_index_get on both objects
    hashes = _index_hashes(host, lib)
    assert np.array_equal(hashes, _index_hashes(dev, lib))
    # and the .ind files of the two are the same bytes (the host writer's are byte-identical to the reference's, test_oracle.py)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        pa, pb = os.path.join(d, "host.ind"), os.path.join(d, "dev.ind")
        host.write(pa); dev.write(pb)
        assert open(pa, "rb").read() == open(pb, "rb").read(), ".ind written from the device-built index differs"
    for h in hashes[:: max(1, len(hashes) // 4000)]:
        a, b = host.get(int(h)), dev.get(int(h))
        assert np.array_equal(a, b), f"positions of key {int(h):#x} differ"
        checked += 1
    # and the table the device built serves the same reads as the uploaded host index
    ctx.close()
    return checked, int(n)


def _index_hashes(index, lib):
    """sorted key hashes of a host index object, through the .ind writer/reader-independent accessor rh_index_get"""
    import tempfile, os
    # the C ABI has no key iterator: write the .ind and parse key words back (layout: SURVEY App. A.8)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.ind")
        index.write(p)
        return _ind_hashes(p)


def _ind_hashes(path):
    import struct
    with open(path, "rb") as f:
        buf = f.read()
    off = 2
    w, e, n, q, k, n_seq, flag = struct.unpack_from("<7I", buf, off); off += 28 + 16
    n_pore = struct.unpack_from("<I", buf, off + 16)[0]; off += 32 + n_pore * 4 + n_pore * 12
    for _ in range(n_seq):
        l = buf[off]; off += 1 + l + 4
    out = []
    for b in range(1 << 14):
        npos = struct.unpack_from("<i", buf, off)[0]; off += 4 + npos * 8
        size = struct.unpack_from("<I", buf, off)[0]; off += 4
        kv = np.frombuffer(buf, dtype="<u8", count=size * 2, offset=off); off += size * 16
        out.append(((kv[0::2] >> np.uint64(1)) << np.uint64(14)) | np.uint64(b))
    return np.sort(np.concatenate(out))


def check_e2e(ctx, wl, reads=None):
    reads = reads or wl.reads
    recs = ctx.map_batch(wl.opts, reads)
    got = [strip_mt(x) for x in paf_lines(wl.index, recs, reads.names, lib=ctx._l)]
    want = wl.oracle_paf(reads)
    assert len(got) == len(want) == len(reads)
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, f"{len(bad)} PAF lines differ, first: {bad[0]}"
    return recs


def check_consumed_prefix_staging(ctx, wl, pinned=False, seed=0):
    """rh_read_batch_t::n_filtered + page-locked samples: the device fetches only the stretches of signal the rounds consume (k_need / k_fetch /
    k_prefilter again).  The records must be those of the same batch uploaded whole - which check_e2e pins to the oracle - also for reads with so
    many samples outside 30 .. 200 pA that the stretch fetched ahead does not hold the round's chunk (the top-up loop), for a slice that does
    not start at offset 0, and a wrong count must fail the call."""
    from rawhash_amd import _capi
    from rawhash_amd.api import Reads, RhError
    l = ctx._l
    rng = np.random.default_rng(seed)
    smp = wl.reads.samples.copy()
    off = wl.reads.offsets
    n = len(off) - 1
    for r in range(0, n, 3):            # every third read: stretches of out-of-range samples (dropped by the filter), up to 70 % of a stretch of the read
        b, e = int(off[r]), int(off[r + 1])
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(b, e)); m = int(rng.integers(50, 9000))
            idx = np.arange(a, min(a + m, e))
            idx = idx[rng.random(len(idx)) < rng.uniform(0.2, 0.7)]
            smp[idx] = rng.choice(np.array([-32768, 32767], dtype=np.int16), size=len(idx))
    reads = Reads(smp, off, wl.reads.names, wl.reads.cal_offset, wl.reads.cal_scale)
    check_e2e(ctx, wl, reads)           # (the odd reads map like the oracle says when uploaded whole)
    want = ctx.map_batch(wl.opts, reads)
    pin = None
    try:
        if pinned:
            pin = l.rh_pinned_alloc(max(len(smp), 1) * 2 + 64)
            assert pin
            host = np.ctypeslib.as_array(C.cast(pin + 6, C.POINTER(C.c_int16)), shape=(max(len(smp), 1),))   # (+ 6 bytes: not 16-byte aligned)
            host[: len(smp)] = smp
        else:
            host = smp
        plain = _capi.make_batch(host[: len(smp)], off, reads.cal_offset, reads.cal_scale)
        nf = _capi.count_filtered(plain, n_threads=3, lib=l)
        assert nf.sum() < len(smp) and nf.sum() > 0
        counted = _capi.make_batch(host[: len(smp)], off, reads.cal_offset, reads.cal_scale, n_filtered=nf)
        got = ctx.map_batch(wl.opts, counted)
        assert got.tobytes() == want.tobytes(), "records differ between whole-batch upload and consumed-prefix staging"
        assert (got["tag_sl"] == nf.astype(np.int32)).all()
        lo = n // 3                      # a slice of the batch: absolute offsets that do not start at 0
        part = _capi.make_batch(host[: len(smp)], off[lo:], reads.cal_offset[lo:], reads.cal_scale[lo:], n_filtered=nf[lo:])
        got = ctx.map_batch(wl.opts, part)
        w2 = want[lo:].copy(); w2["read_idx"] -= lo
        assert got.tobytes() == w2.tobytes(), "slice: records differ"
        bad = nf.copy()
        unm = np.flatnonzero(want["mapped"] == 0)
        assert len(unm), "the workload has no unmappable read (one whose whole signal the device gets to see)"
        bad[unm[0]] += 1
        wrong = _capi.make_batch(host[: len(smp)], off, reads.cal_offset, reads.cal_scale, n_filtered=bad)
        try:
            ctx.map_batch(wl.opts, wrong)
            raise AssertionError("a wrong n_filtered went unnoticed")
        except RhError as e:
            assert "n_filtered" in str(e)
    finally:
        if pin:
            l.rh_pinned_free(pin)
    return int((want["mapped"] == 1).sum())


def check_ava(lib, case, directory, oracle_threads=4):
    """Rawsamble on the device path of `lib`: the signal-target index built from the reads must be the reference's .ind byte
    for byte (golden hash of the file `ref_harness sigindex` wrote), the oracle on that file and the device's all-vs-all
    mapping must both print the reference's PAF (golden)."""
    import hashlib
    import golden
    import oracle_lib as O
    from rawhash_amd.api import Context, Index, paf_lines
    os.makedirs(str(directory), exist_ok=True)
    w = golden.build_ava_case(case, directory, lib)
    c = Context(0, lib=lib)
    try:
        ix = Index.build_signals_device(c, w.reads, w.model, w.opts)
        ix.download(c)
        ind = os.path.join(str(directory), "device_built.ind")
        ix.write(ind)
        assert hashlib.sha256(O.mask_ind(open(ind, "rb").read())).hexdigest() == case["ind_sha256"], "signal-target .ind differs from the reference's"
        want = golden.expected_paf(case)
        assert w.oracle_paf(ind, n_threads=oracle_threads) == want, "oracle (all-vs-all) differs from the reference's PAF"
        w.opts.update(ix)
        recs, off = c.map_batch_multi(w.opts, w.reads, ix)
        got = [O.strip_mt(x) for x in paf_lines(ix, recs, w.reads.names, lib=lib)]
        bad = [(g, x) for g, x in zip(got, want) if g != x]
        assert len(got) == len(want) and not bad, f"{case['name']}: {len(bad)} PAF lines differ, first: {bad[:1]}"
        assert int(off[-1]) == len(recs) and all(int(r["read_idx"]) == i for i in range(len(w.reads)) for r in recs[int(off[i]):int(off[i + 1])])
        return len(recs)
    finally:
        c.close()

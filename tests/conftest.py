import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def product_lib():
    """The hipcc-built product library (host-side entry points work without a GPU)."""
    from rawhash_amd import _capi, build as rb
    if not os.path.exists(_capi.LIB_PATH):
        rb.build()
    return _capi.lib()


@pytest.fixture(scope="session")
def emu_lib():
    """Same sources compiled against the SIMT emulator (tests/emu) -- CPU-side kernel-logic checks only."""
    from rawhash_amd import _capi
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return _capi.load(build_emu.build())


@pytest.fixture(scope="session")
def emu_lib_smallcaps():
    """Emulator build with tiny LDS caps: forces the oversized-read fallback kernels on small inputs."""
    from rawhash_amd import _capi
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return _capi.load(build_emu.build(defines=build_emu.SMALL_CAPS, tag="_smallcaps"))


class Workload:
    """Synthetic reference + index + reads, generated from seeds by the product library's host code."""

    def __init__(self, directory, lib, preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=12_000, n_reads=48,
                 junk=150, noise=150_000, read_seed=3, index_lib=None, build_index=True, no_adaptive=False, fast5=False, mapopt=None, idxflag=0, r10=False):
        from rawhash_amd.api import SynthWorkload, MapOptions, Index
        self.dir, self.preset = str(directory), preset
        # r10: `--r10` (main.cpp:396-406) - 9-mers (a pore model of 4^9 levels), segmentation windows 3 / 6, thresholds 6.5 / 4.0, peak height 0.2, gap scale 1.2
        self.wl = SynthWorkload(chrom_len=chrom_len, n_chrom=n_chrom, n_samples=n_samples, junk_per_1024=junk, noise_q24=noise,
                                read_seed=read_seed, lib=lib, k=9 if r10 else 6)
        self.fasta, self.model = self.wl.write_reference(self.dir)
        self.opts = MapOptions(preset, lib=lib)
        self.no_adaptive = no_adaptive
        self.mapopt = dict(mapopt or {})        # rh_mapopt_t fields set on top of the preset ("flag" is OR-ed in): --rmq, --bw-long ...
        if r10:
            self.mapopt.update(window_length1=3, window_length2=6, threshold1=6.5, threshold2=4.0, peak_height=0.2, chain_gap_scale=1.2)
            self.opts.io.k = 9
        if no_adaptive:
            self.opts.mo.flag |= 0x20           # RH_M_NO_ADAPTIVE: one round over the whole read
        self._apply_mapopt(self.opts.mo)
        self.opts.io.flag |= idxflag            # e.g. 0x10 = RH_I_STORE_SIG (--store-sig: DTW re-scoring needs the target signals)
        self.ind = os.path.join(self.dir, f"ref_{preset}.ind")
        self.index = None
        if build_index:     # (large references: the caller builds the index on the device instead)
            self.index = Index.build(self.fasta, self.model, self.opts, out_ind=self.ind, n_threads=8, lib=lib)
            self.opts.update(self.index)
        self.reads = self.wl.reads(self.model, 0, n_reads)
        self.reads.fast5 = bool(fast5)     # the same int16 samples taken in the way the reference's FAST5 reader does (rsig.c:346-374)

    def _apply_mapopt(self, mo):
        for k, v in self.mapopt.items():
            if k == "flag":
                mo.flag |= v
            else:
                setattr(mo, k, v)

    def oracle(self):
        import oracle_lib as O
        oix = O.OracleIndex(self.ind)
        _, mo = O.preset(self.preset)
        if self.no_adaptive:
            mo.flag |= 0x20
        self._apply_mapopt(mo)
        O.lib().ro_mapopt_update(C.byref(mo), oix.h)
        return oix, mo

    def oracle_paf(self, reads=None, n_threads=4):
        import oracle_lib as O
        reads = reads or self.reads
        oix, mo = self.oracle()
        recs = O.map_batch(oix, mo, reads.batch(), n_threads=n_threads)
        return [O.strip_mt(x) for x in O.paf_lines(oix, recs, reads.names)]


class AvaWorkload:
    """Rawsamble input: synthetic reads that overlap each other (drawn from one short genome), to be indexed as signal targets
    and overlapped all-vs-all."""

    def __init__(self, directory, lib, preset="ava", chrom_len=20_000, n_samples=27_000, n_reads=60, junk=50, noise=150_000, read_seed=21, ragged=False):
        from rawhash_amd.api import SynthWorkload, MapOptions
        self.dir, self.preset = str(directory), preset
        self.wl = SynthWorkload(chrom_len=chrom_len, n_chrom=1, n_samples=n_samples, junk_per_1024=junk, noise_q24=noise, read_seed=read_seed, lib=lib)
        self.fasta, self.model = self.wl.write_reference(self.dir)
        self.opts = MapOptions(preset, lib=lib)
        self.reads = self.wl.reads(self.model, 0, n_reads)
        if ragged:      # reads of very different lengths, an empty one and two too short to give min_events events
            from rawhash_amd.api import Reads
            r = self.reads
            lens = [int(n_samples * (0.15 + 0.85 * ((i * 37) % 100) / 99.0)) for i in range(n_reads)]
            if n_reads > 7:
                lens[3], lens[5], lens[7] = 0, 90, 300
            parts = [r.samples[int(r.offsets[i]):int(r.offsets[i]) + lens[i]] for i in range(n_reads)]
            off = np.zeros(n_reads + 1, dtype=np.uint64)
            off[1:] = np.cumsum(lens)
            self.reads = Reads(np.concatenate(parts), off, r.names, r.cal_offset, r.cal_scale)
        cfg = self.wl.cfg
        self.rhr = os.path.join(self.dir, "reads.rhr")
        self.reads.write(self.rhr, cfg.digitisation, cfg.range, cfg.offset, lib=lib)

    def oracle_paf(self, ind_path, n_threads=4):
        """The oracle on an index file (signal-target indexes are built by the device path or by the reference)."""
        import oracle_lib as O
        oix = O.OracleIndex(ind_path)
        _, mo = O.preset(self.preset)
        O.lib().ro_mapopt_update(C.byref(mo), oix.h)
        recs = O.map_batch(oix, mo, self.reads.batch(), names=self.reads.names, n_threads=n_threads, max_rec_per_read=256)
        return [O.strip_mt(x) for x in O.paf_lines(oix, recs, self.reads.names)]


@pytest.fixture(scope="session")
def make_workload(tmp_path_factory, product_lib):
    cache = {}

    def make(lib=None, **kw):
        key = (id(lib),) + tuple(sorted((k, tuple(sorted(v.items())) if isinstance(v, dict) else v) for k, v in kw.items()))
        if key not in cache:
            cache[key] = Workload(tmp_path_factory.mktemp("wl"), lib or product_lib, **kw)
        return cache[key]
    return make


@pytest.fixture(scope="session")
def gpu_ctx_factory(product_lib):
    from rawhash_amd.api import Context
    made = []

    def make():
        c = Context(0, lib=product_lib)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()

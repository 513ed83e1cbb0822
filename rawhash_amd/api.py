"""Host-side mirror of the reference interface for the mapping path, on top of the C ABI.

Names follow the reference: an *index* (`.ind`, ri_idx_t), *map options* (ri_mapopt_t + `-x` presets), a *batch of
reads* (step_mt), `map_batch` (= kt_for(map_worker_for), rmap.cpp:700) returning one record per read (ri_map_t) that
`paf_lines` prints exactly like rmap.cpp:740-783.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import IdxOpt, MapOpt, MapRecord, MapStats, ReadBatch, SynthCfg, RECORD, MM128, ptr


class RhError(RuntimeError):
    pass


def _check(rc, l):
    if rc != 0:
        raise RhError(_capi.last_error(l))


class MapOptions:
    """ri_idxopt_t + ri_mapopt_t with the reference's presets (main.cpp:111-210)."""

    def __init__(self, preset=None, lib=None):
        self._l = lib or _capi.lib()
        self.io, self.mo = IdxOpt(), MapOpt()
        self._l.rh_set_preset(None, C.byref(self.io), C.byref(self.mo))
        if preset not in (None, "default"):
            _check(self._l.rh_set_preset(preset.encode(), C.byref(self.io), C.byref(self.mo)), self._l)

    def update(self, index):
        """ri_mapopt_update (rindex.c:1041): calibrates mid_occ from the index."""
        self._l.rh_mapopt_update(C.byref(self.mo), index.h)
        return self


class Index:
    """Host copy of a RawHash2 index (.ind)."""

    def __init__(self, handle, lib):
        self.h, self._l = handle, lib

    @classmethod
    def load(cls, path, lib=None):
        l = lib or _capi.lib()
        h = l.rh_index_load(os.fsencode(path))
        if not h:
            raise RhError(_capi.last_error(l))
        return cls(h, l)

    @classmethod
    def build(cls, fasta, pore_model, opts, out_ind=None, n_threads=8, lib=None):
        l = lib or _capi.lib()
        h = l.rh_index_build(os.fsencode(fasta), os.fsencode(pore_model), C.byref(opts.io),
                             os.fsencode(out_ind) if out_ind else None, n_threads)
        if not h:
            raise RhError(_capi.last_error(l))
        return cls(h, l)

    @classmethod
    def build_device(cls, ctx, fasta, pore_model, opts, n_threads=8):
        """ri_idx_gen on the GPU: the table is left resident in `ctx` (no upload needed); FASTA read on the host."""
        l = ctx._l
        h = l.rh_index_build_device_fasta(ctx.h, os.fsencode(fasta), os.fsencode(pore_model), C.byref(opts.io), n_threads)
        if not h:
            raise RhError(_capi.last_error(l))
        return cls(h, l)

    @classmethod
    def build_device_seqs(cls, ctx, names, seqs, pore_model, opts, n_threads=8):
        """Same from sequences already in host memory: seqs = list of uint8 numpy arrays (ASCII bases)."""
        l = ctx._l
        n = len(seqs)
        keep = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        np_arr = (C.c_char_p * n)(*[x.encode() for x in names])
        sp_arr = (C.c_char_p * n)(*[C.cast(k.ctypes.data, C.c_char_p) for k in keep])
        lens = np.array([len(k) for k in keep], dtype=np.uint32)
        h = l.rh_index_build_device(ctx.h, n, np_arr, sp_arr, ptr(lens), os.fsencode(pore_model), C.byref(opts.io), n_threads)
        if not h:
            raise RhError(_capi.last_error(l))
        return cls(h, l)

    @classmethod
    def build_signals_device(cls, ctx, reads, pore_model, opts):
        """ri_idx_siggen on the GPU (Rawsamble): every read becomes a target; the index is left resident in `ctx`."""
        l = ctx._l
        b = reads.batch()
        arr = (C.c_char_p * len(reads.names))(*[n.encode() for n in reads.names])
        h = l.rh_index_build_signals_device(ctx.h, C.byref(b), arr, os.fsencode(pore_model), C.byref(opts.io), C.byref(opts.mo))
        if not h:
            raise RhError(_capi.last_error(l))
        return cls(h, l)

    def name_ranks(self, names):
        """(query ranks of `names`, ranks of this index's targets): strcmp(q, t) >= 0 <=> rank(q) >= rank(t) (rmap.cpp:86)."""
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        q = np.zeros(max(len(names), 1), dtype=np.uint32)
        t = np.zeros(max(self.n_seq, 1), dtype=np.uint32)
        _check(self._l.rh_index_name_ranks(self.h, arr, len(names), ptr(q), ptr(t)), self._l)
        return q[: len(names)], t[: self.n_seq]

    def download(self, ctx, n_threads=8):
        """Fetch keys + positions of a device-built index into this host object (for get() / write())."""
        _check(self._l.rh_index_download(ctx.h, self.h, n_threads), self._l)
        return self

    def write(self, path):
        _check(self._l.rh_index_write(self.h, os.fsencode(path)), self._l)

    def close(self):
        if self.h:
            self._l.rh_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_seq(self):
        return self._l.rh_index_n_seq(self.h)

    def seq_name(self, i):
        return self._l.rh_index_seq_name(self.h, i).decode()

    def seq_len(self, i):
        return self._l.rh_index_seq_len(self.h, i)

    @property
    def n_keys(self):
        return self._l.rh_index_n_keys(self.h)

    @property
    def n_positions(self):
        return self._l.rh_index_n_positions(self.h)

    def params(self):
        io = IdxOpt()
        self._l.rh_index_params(self.h, C.byref(io))
        return io

    def get(self, hashval):
        n = C.c_int(0)
        p = self._l.rh_index_get(self.h, int(hashval), C.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()


class Reads:
    """A batch of raw reads (int16 ADC samples + calibration), SoA/CSR."""

    def __init__(self, samples, offsets, names, cal_offset, cal_scale, fast5=False):
        self.fast5 = bool(fast5)          # raw -> pA as the reference's FAST5 reader does it (rh_read_batch_t.fast5_ingest)
        self.samples = np.ascontiguousarray(samples, dtype=np.int16)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.names = list(names)
        n = len(self.offsets) - 1
        self.cal_offset = np.ascontiguousarray(np.broadcast_to(np.asarray(cal_offset, dtype=np.float64), (n,)))
        self.cal_scale = np.ascontiguousarray(np.broadcast_to(np.asarray(cal_scale, dtype=np.float32), (n,)))

    def __len__(self):
        return len(self.offsets) - 1

    def batch(self):
        return _capi.make_batch(self.samples, self.offsets, self.cal_offset, self.cal_scale, fast5=self.fast5)

    def subset(self, idx):
        idx = list(idx)
        parts = [self.samples[int(self.offsets[i]):int(self.offsets[i + 1])] for i in idx]
        off = np.zeros(len(idx) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in parts])
        return Reads(np.concatenate(parts) if parts else np.zeros(0, np.int16), off, [self.names[i] for i in idx],
                     self.cal_offset[idx], self.cal_scale[idx], fast5=self.fast5)

    @classmethod
    def load(cls, path, lib=None):
        l = lib or _capi.lib()
        h = l.rh_reads_load(os.fsencode(path))
        if not h:
            raise RhError(_capi.last_error(l))
        try:
            b = ReadBatch()
            l.rh_reads_batch(h, C.byref(b))
            n = b.n_reads
            off = np.ctypeslib.as_array(C.cast(b.offsets, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            tot = int(off[n])
            smp = np.ctypeslib.as_array(C.cast(b.samples, C.POINTER(C.c_int16)), shape=(max(tot, 1),))[:tot].copy()
            co = np.ctypeslib.as_array(C.cast(b.cal_offset, C.POINTER(C.c_double)), shape=(max(n, 1),))[:n].copy()
            cs = np.ctypeslib.as_array(C.cast(b.cal_scale, C.POINTER(C.c_float)), shape=(max(n, 1),))[:n].copy()
            names = [l.rh_reads_name(h, i).decode() for i in range(n)]
        finally:
            l.rh_reads_destroy(h)
        return cls(smp, off, names, co, cs)

    def write(self, path, digitisation, rng, offset, lib=None):
        l = lib or _capi.lib()
        arr = (C.c_char_p * len(self.names))(*[n.encode() for n in self.names])
        _check(l.rh_reads_write(os.fsencode(path), len(self.names), arr, ptr(self.samples), ptr(self.offsets),
                                float(digitisation), float(rng), float(offset)), l)


class ReadsFile:
    """A read file kept in the library's own staging buffer (page-locked when a GPU is there): batch() is a view into it, so
    rh_map_batch uploads straight from the memory the reader decoded into."""

    def __init__(self, path, lib=None):
        self._l = lib or _capi.lib()
        self.h = self._l.rh_reads_load(os.fsencode(path))
        if not self.h:
            raise RhError(_capi.last_error(self._l))
        self.names = [self._l.rh_reads_name(self.h, i).decode() for i in range(self._l.rh_reads_n(self.h))]
        self.pinned = bool(self._l.rh_reads_pinned(self.h))

    def __len__(self):
        return len(self.names)

    def batch(self):
        b = ReadBatch()
        self._l.rh_reads_batch(self.h, C.byref(b))
        b._keep = self
        return b

    def close(self):
        if self.h:
            self._l.rh_reads_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_blow5(reads, path, digitisation, rng, offset, sampling_rate=4000.0, zlib_records=False, lib=None, records=None, svb_zd=False):
    """Write a Reads batch as BLOW5 (one read group, no auxiliary fields).  records: "none" | "zlib" | "zstd" (default: zlib_records);
    svb_zd: StreamVByte zig-zag delta compression of the signal."""
    code = {"none": 0, "zlib": 1, "zstd": 2}[records] if records else int(bool(zlib_records))
    zlib_records = code | (0x100 if svb_zd else 0)
    l = lib or _capi.lib()
    arr = (C.c_char_p * len(reads.names))(*[n.encode() for n in reads.names])
    _check(l.rh_reads_write_blow5(os.fsencode(path), len(reads.names), arr, ptr(reads.samples), ptr(reads.offsets),
                                  float(digitisation), float(rng), float(offset), float(sampling_rate), int(zlib_records)), l)


class Context:
    """One GPU: HIP stream, device arenas and the HBM-resident index."""

    def __init__(self, device=0, lib=None):
        self._l = lib or _capi.lib()
        h = C.c_void_p()
        _check(self._l.rh_ctx_create(C.byref(h), device), self._l)
        self.h = h

    def close(self):
        if self.h:
            self._l.rh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, index):
        _check(self._l.rh_index_upload(self.h, index.h), self._l)
        return self

    def device_blob(self):
        p, n, hdr = C.c_void_p(), C.c_uint64(), C.create_string_buffer(256)
        _check(self._l.rh_index_device_blob(self.h, C.byref(p), C.byref(n), hdr), self._l)
        return p.value, n.value, hdr.raw

    def copy_blob_to(self, dev_ptr):
        _check(self._l.rh_index_copy_blob(self.h, C.c_void_p(dev_ptr)), self._l)

    def adopt_blob(self, dev_ptr, nbytes, header, take_ownership=False):
        _check(self._l.rh_index_adopt_blob(self.h, None, C.c_void_p(dev_ptr), nbytes, header, int(take_ownership)), self._l)

    def map_batch(self, opts, batch):
        """kt_for(map_worker_for): one record per read, in read order.  `batch` is a Reads or a ReadBatch."""
        b = batch.batch() if isinstance(batch, Reads) else batch
        cap = self._l.rh_map_max_records(C.byref(b), C.byref(opts.mo))
        out = np.zeros(max(cap, 1), dtype=RECORD)
        n = C.c_uint64(0)
        _check(self._l.rh_map_batch(self.h, C.byref(opts.mo), C.byref(b), ptr(out), cap, C.byref(n)), self._l)
        return out[: n.value]

    def map_batch_multi(self, opts, reads, index, max_records=None, device_batch=None):
        """All-vs-all overlapping (ava presets): every reported chain of a read is a record.  Returns (records, offsets):
        the records of read r are records[offsets[r]:offsets[r + 1]].  `device_batch`: the same reads already resident in HBM
        (a ReadBatch of device pointers); `reads` then only supplies the names."""
        b = device_batch if device_batch is not None else reads.batch()
        qr, _ = index.name_ranks(reads.names)
        qr = np.ascontiguousarray(qr)
        b.name_rank = qr.ctypes.data if len(qr) else None
        cap = max_records or (64 * len(reads) + 1024)
        out = np.zeros(cap, dtype=RECORD)
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        try:
            _check(self._l.rh_map_batch_multi(self.h, C.byref(opts.mo), C.byref(b), ptr(out), cap, ptr(off), C.byref(n)), self._l)
        finally:
            b.name_rank = None      # qr dies with this frame: never leave its address in the caller's batch
        return out[: n.value], off

    def set_target_ranks(self, ranks):
        r = np.ascontiguousarray(ranks, dtype=np.uint32)
        _check(self._l.rh_index_set_target_ranks(self.h, ptr(r), len(r)), self._l)

    def map_submit(self, opts, batch):
        """Start mapping a batch (rh_map_submit); returns a handle for map_wait.  Up to 2 batches may be in flight."""
        b = batch.batch() if isinstance(batch, Reads) else batch
        cap = self._l.rh_map_max_records(C.byref(b), C.byref(opts.mo))
        out = np.zeros(max(cap, 1), dtype=RECORD)
        t = _capi.Ticket()
        _check(self._l.rh_map_submit(self.h, C.byref(opts.mo), C.byref(b), ptr(out), cap, C.byref(t)), self._l)
        # the ctypes struct holds raw pointers only: the handle keeps the source object (and with it the numpy arrays the
        # background mapping thread reads) alive until map_wait
        return (t, out, b, batch)

    def map_wait(self, handle):
        t, out = handle[0], handle[1]
        n = C.c_uint64(0)
        _check(self._l.rh_map_wait(self.h, t, C.byref(n)), self._l)
        return out[: n.value]

    def stats(self):
        s = MapStats()
        self._l.rh_map_last_stats(self.h, C.byref(s))
        d = {k: getattr(s, k) for k in ("n_reads", "n_chunks", "n_samples_raw", "n_samples_used", "n_events", "n_seeds",
                                         "n_hits", "n_anchors", "n_chained", "ms_total")}
        d["n_rmq_class"] = [int(s.n_rmq_class[i]) for i in range(4)]
        d["n_dtw_device"], d["n_dtw_host"] = int(s.n_dtw_device), int(s.n_dtw_host)
        d["stages"] = {self._l.rh_stage_name(i).decode(): (s.ms_kernel[i], s.n_launch[i]) for i in range(24)
                       if self._l.rh_stage_name(i)}
        return d

    # ---- stage-level entry points (parity tests)
    def events(self, opts, reads, chunk):
        b = reads.batch()
        n = len(reads)
        cap = n * 2048 + 16
        ev = np.zeros(cap, dtype=np.float32)
        off = np.zeros(n + 1, dtype=np.uint64)
        lsig = np.zeros(max(n, 1), dtype=np.uint32)
        _check(self._l.rh_events_batch(self.h, C.byref(opts.mo), C.byref(b), chunk, ptr(ev), cap, ptr(off), ptr(lsig)), self._l)
        return ev[: int(off[n])], off, lsig[:n]

    def sketch(self, events, ev_off):
        n = len(ev_off) - 1
        cap = len(events) * 4 + 16
        sd = np.zeros(cap, dtype=MM128)
        off = np.zeros(n + 1, dtype=np.uint64)
        ev = np.ascontiguousarray(events, dtype=np.float32)
        eo = np.ascontiguousarray(ev_off, dtype=np.uint64)
        _check(self._l.rh_sketch_batch(self.h, n, ptr(ev), ptr(eo), ptr(sd), cap, ptr(off)), self._l)
        return sd[: int(off[n])], off

    def seed(self, opts, seeds, seed_off, q_offset=None, prev=None, prev_off=None, cap=None):
        n = len(seed_off) - 1
        cap = cap or (len(seeds) * 600 + (0 if prev is None else len(prev)) + 1024)
        an = np.zeros(cap, dtype=MM128)
        off = np.zeros(n + 1, dtype=np.uint64)
        rep = np.zeros(max(n, 1), dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=MM128)
        so = np.ascontiguousarray(seed_off, dtype=np.uint64)
        qo = None if q_offset is None else np.ascontiguousarray(q_offset, dtype=np.uint32)
        pv = None if prev is None else np.ascontiguousarray(prev, dtype=MM128)
        po = None if prev_off is None else np.ascontiguousarray(prev_off, dtype=np.uint64)
        _check(self._l.rh_seed_batch(self.h, C.byref(opts.mo), n, ptr(sd), ptr(so), ptr(qo), ptr(pv), ptr(po),
                                     ptr(an), cap, ptr(off), ptr(rep)), self._l)
        return an[: int(off[n])], off, rep[:n]

    def chain(self, opts, anchors, a_off):
        n = len(a_off) - 1
        cap = len(anchors) + 16
        ch = np.zeros(cap, dtype=MM128)
        pv = np.zeros(cap, dtype=MM128)
        co = np.zeros(n + 1, dtype=np.uint64)
        u = np.zeros(cap, dtype=np.uint64)
        uo = np.zeros(n + 1, dtype=np.uint64)
        an = np.ascontiguousarray(anchors, dtype=MM128)
        ao = np.ascontiguousarray(a_off, dtype=np.uint64)
        _check(self._l.rh_chain_batch(self.h, C.byref(opts.mo), n, ptr(an), ptr(ao), ptr(ch), cap, ptr(co), ptr(u), cap, ptr(uo), ptr(pv)), self._l)
        return ch[: int(co[n])], co, u[: int(uo[n])], uo, pv[: int(co[n])]

    def regions(self, opts, anchors, a_off, rep_len, qlen, all_regions=False):
        """Chains of sorted anchors -> regions: (n, 10) int32 = n_cregs, cnt, score, mapq, qs, qe, rs, re, rid, rev of creg[0];
        all_regions: also every kept region, (k, 18) int32 in the reference's field order, and their (n + 1) offsets."""
        n = len(a_off) - 1
        an = np.ascontiguousarray(anchors, dtype=MM128)
        ao = np.ascontiguousarray(a_off, dtype=np.uint64)
        rl = np.ascontiguousarray(rep_len, dtype=np.int32)
        ql = np.ascontiguousarray(qlen, dtype=np.uint32)
        out = np.zeros((max(n, 1), 10), dtype=np.int32)
        if not all_regions:
            _check(self._l.rh_regions_batch(self.h, C.byref(opts.mo), n, ptr(an), ptr(ao), ptr(rl), ptr(ql), ptr(out), None, 0, None), self._l)
            return out[:n]
        cap = max(len(an), 1)
        regs = np.zeros((cap, 18), dtype=np.int32)
        ro = np.zeros(n + 1, dtype=np.uint64)
        _check(self._l.rh_regions_batch(self.h, C.byref(opts.mo), n, ptr(an), ptr(ao), ptr(rl), ptr(ql), ptr(out), ptr(regs), cap, ptr(ro)), self._l)
        return out[:n], regs[: int(ro[n])], ro

    def sort128x(self, arr, offsets):
        a = np.ascontiguousarray(arr, dtype=MM128).copy()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(self._l.rh_sort128x_batch(self.h, len(off) - 1, ptr(a), ptr(off)), self._l)
        return a

    def sort128x_packed(self, arr, offsets, lo_bits, mid_bits, any_order=False):
        """rh_sort128x_batch on one-word records (the mapping path's anchor format)."""
        a = np.ascontiguousarray(arr, dtype=MM128).copy()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(self._l.rh_sort128x_packed_batch(self.h, len(off) - 1, ptr(a), ptr(off), lo_bits, mid_bits, 1 if any_order else 0), self._l)
        return a

    def sort128x_any(self, arr, offsets):
        """The sorter's any-order path (region keys): returns (sorted copy, has_ties per segment)."""
        a = np.ascontiguousarray(arr, dtype=MM128).copy()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ties = np.zeros(max(len(off) - 1, 1), dtype=np.uint8)
        _check(self._l.rh_sort128x_any_batch(self.h, len(off) - 1, ptr(a), ptr(off), ptr(ties)), self._l)
        return a, ties[: len(off) - 1]


def paf_lines(index, recs, names, mt_ms=0.0, lib=None):
    """PAF text of the records exactly as rmap.cpp:740-783 prints it (mt:f: filled with mt_ms)."""
    l = lib or _capi.lib()
    buf = C.create_string_buffer(4096)
    out = []
    for r in recs:
        rec = MapRecord.from_buffer_copy(r.tobytes())
        n = l.rh_paf_format(index.h, C.byref(rec), names[int(r["read_idx"])].encode(), mt_ms, buf, 4096)
        if n < 0:
            raise RhError(_capi.last_error(l))
        if n:
            out.append(buf.value.decode())
    return out


def strip_mt(line):
    """Drop the wall-clock mt:f: tag (excluded from parity)."""
    return "\t".join(f for f in line.rstrip("\n").split("\t") if not f.startswith("mt:f:"))


class SynthWorkload:
    """Deterministic synthetic genome / pore model / reads, generated by the C library."""

    def __init__(self, chrom_len=4_600_000, n_chrom=1, n_samples=40_000, junk_per_1024=0, noise_q24=0,
                 model_seed=1, genome_seed=2, read_seed=3, lib=None, k=6):
        self._l = lib or _capi.lib()
        self.k = k                          # k-mers of the pore model write_reference writes (6: R9.4; 9: R10)
        c = SynthCfg()
        self._l.rh_synth_cfg_init(C.byref(c))
        c.chrom_len, c.n_chrom, c.n_samples = chrom_len, n_chrom, n_samples
        c.junk_per_1024, c.noise_q24 = junk_per_1024, noise_q24
        c.model_seed, c.genome_seed, c.read_seed = model_seed, genome_seed, read_seed
        self.cfg = c

    def write_reference(self, directory):
        os.makedirs(directory, exist_ok=True)
        model, fasta = os.path.join(directory, "model.txt"), os.path.join(directory, "ref.fa")
        _check(self._l.rh_synth_write_model_k(C.byref(self.cfg), os.fsencode(model), self.k), self._l)
        _check(self._l.rh_synth_write_fasta(C.byref(self.cfg), os.fsencode(fasta)), self._l)
        return fasta, model

    def genome(self, chrom, n_threads=8):
        """Bases of one chromosome as a uint8 array (what write_reference puts into the FASTA)."""
        out = np.empty(self.cfg.chrom_len, dtype=np.uint8)
        _check(self._l.rh_synth_genome(C.byref(self.cfg), chrom, ptr(out), n_threads), self._l)
        return out

    def reads(self, model_path, first, n, n_threads=8, with_names=True):
        ns = self.cfg.n_samples
        smp = np.zeros(n * ns, dtype=np.int16)
        names = C.create_string_buffer(n * 64) if with_names else None
        _check(self._l.rh_synth_reads(C.byref(self.cfg), os.fsencode(model_path), first, n, ptr(smp), names, n_threads), self._l)
        nm = [names.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode() for i in range(n)] if with_names else [f"r{first + i}" for i in range(n)]
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(ns)
        return Reads(smp, off, nm, self.cfg.offset, np.float32(self.cfg.range / self.cfg.digitisation))

    def reads_device(self, ctx, model_path, first, n):
        """Generate the same reads straight into ctx's HBM; returns a ReadBatch of device pointers (owned by ctx)."""
        b = ReadBatch()
        _check(self._l.rh_synth_reads_device(ctx.h, C.byref(self.cfg), os.fsencode(model_path), first, n, C.byref(b)), self._l)
        return b

    def origin(self, idx):
        ch, pos, st, jk = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(self._l.rh_synth_origin(C.byref(self.cfg), idx, C.byref(ch), C.byref(pos), C.byref(st), C.byref(jk)), self._l)
        return ch.value, pos.value, st.value, bool(jk.value)

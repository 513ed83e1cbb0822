"""ctypes mirror of include/rawhash_amd.h (struct layouts + library loader).

The product library is `rawhash_amd/librawhash_amd.so` (built in-tree by `__graft_entry__.build()` /
`rawhash_amd/build.py`).  There is no Python or CPU fallback: if the shared object is missing this module
raises on first use.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librawhash_amd.so")

MM128 = np.dtype([("x", "<u8"), ("y", "<u8")])


class IdxOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("b", "w", "e", "n", "q", "k", "flag", "lev_col")] + \
               [(n, C.c_float) for n in ("diff", "fine_min", "fine_max", "fine_range")]


class MapOpt(C.Structure):
    _fields_ = [
        ("bp_per_sec", C.c_uint32), ("sample_rate", C.c_uint32), ("chunk_size", C.c_uint32),
        ("sample_per_base", C.c_float),
        ("mid_occ_frac", C.c_float),
        ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32),
        ("mid_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
        ("min_events", C.c_uint32),
        ("bw", C.c_int32), ("bw_long", C.c_int32), ("max_target_gap_length", C.c_int32),
        ("max_query_gap_length", C.c_int32), ("max_chain_iter", C.c_int32),
        ("max_num_skips", C.c_int32), ("min_num_anchors", C.c_int32), ("min_chaining_score", C.c_int32),
        ("min_chaining_score2", C.c_int32),
        ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
        ("w_bestq", C.c_float), ("w_bestmq", C.c_float), ("w_bestmc", C.c_float), ("w_threshold", C.c_float),
        ("mask_level", C.c_float), ("mask_len", C.c_int32),
        ("pri_ratio", C.c_float), ("best_n", C.c_int32),
        ("alt_drop", C.c_float),
        ("max_num_chunk", C.c_uint32),
        ("min_mapq", C.c_int32),
        ("flag", C.c_int64),
        ("window_length1", C.c_uint32), ("window_length2", C.c_uint32),
        ("threshold1", C.c_float), ("threshold2", C.c_float), ("peak_height", C.c_float),
        ("rmq_inner_dist", C.c_int32), ("rmq_size_cap", C.c_int32),
        ("dtw_border_constraint", C.c_uint32), ("dtw_fill_method", C.c_uint32),
        ("dtw_band_radius_frac", C.c_float), ("dtw_match_bonus", C.c_float), ("dtw_min_score", C.c_float), ("w_bestma", C.c_float),
    ]


class MapRecord(C.Structure):
    _fields_ = [
        ("read_idx", C.c_uint32), ("read_length", C.c_uint32), ("ref_id", C.c_uint32),
        ("read_start_position", C.c_uint32), ("read_end_position", C.c_uint32),
        ("fragment_start_position", C.c_uint32), ("fragment_length", C.c_uint32),
        ("mapq", C.c_uint8), ("rev", C.c_uint8), ("mapped", C.c_uint8), ("_pad", C.c_uint8),
        ("tag_ci", C.c_int32), ("tag_sl", C.c_int32), ("tag_cm", C.c_int32), ("tag_nc", C.c_int32), ("tag_s1", C.c_int32),
    ]


RECORD = np.dtype([
    ("read_idx", "<u4"), ("read_length", "<u4"), ("ref_id", "<u4"), ("read_start_position", "<u4"),
    ("read_end_position", "<u4"), ("fragment_start_position", "<u4"), ("fragment_length", "<u4"),
    ("mapq", "u1"), ("rev", "u1"), ("mapped", "u1"), ("_pad", "u1"),
    ("tag_ci", "<i4"), ("tag_sl", "<i4"), ("tag_cm", "<i4"), ("tag_nc", "<i4"), ("tag_s1", "<i4"),
])
assert RECORD.itemsize == C.sizeof(MapRecord)


class ReadBatch(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32),
        ("samples", C.c_void_p), ("offsets", C.c_void_p), ("cal_offset", C.c_void_p), ("cal_scale", C.c_void_p),
        ("name_rank", C.c_void_p),
        ("samples_on_device", C.c_int),
        ("fast5_ingest", C.c_int),
        ("n_filtered", C.c_void_p),
    ]


class MapStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "n_chunks", "n_samples_raw", "n_samples_used", "n_events",
                                          "n_seeds", "n_hits", "n_anchors", "n_chained")] + \
               [("ms_total", C.c_double), ("ms_kernel", C.c_double * 24), ("n_launch", C.c_uint32 * 24), ("n_rmq_class", C.c_uint64 * 4), ("n_dtw_device", C.c_uint64), ("n_dtw_host", C.c_uint64)]


class Ticket(C.Structure):
    _fields_ = [("slot", C.c_int32), ("serial", C.c_uint32)]


class SynthCfg(C.Structure):
    _fields_ = [
        ("model_seed", C.c_uint64), ("genome_seed", C.c_uint64), ("read_seed", C.c_uint64),
        ("n_chrom", C.c_uint32), ("chrom_len", C.c_uint32), ("n_samples", C.c_uint32),
        ("junk_per_1024", C.c_uint32), ("noise_q24", C.c_uint32),
        ("digitisation", C.c_double), ("range", C.c_double), ("offset", C.c_double),
    ]


def ptr(a):
    """void* of a numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_batch(samples, offsets, cal_offset=None, cal_scale=None, name_rank=None, keep=None, fast5=False, n_filtered=None):
    """Build a ReadBatch view over numpy arrays; the arrays are returned too so callers keep them alive."""
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    co = None if cal_offset is None else np.ascontiguousarray(cal_offset, dtype=np.float64)
    cs = None if cal_scale is None else np.ascontiguousarray(cal_scale, dtype=np.float32)
    nr = None if name_rank is None else np.ascontiguousarray(name_rank, dtype=np.uint32)
    nf = None if n_filtered is None else np.ascontiguousarray(n_filtered, dtype=np.uint32)
    b = ReadBatch(n, ptr(samples), ptr(offsets), ptr(co), ptr(cs), ptr(nr), 0, 1 if fast5 else 0, ptr(nf))
    b._keep = (samples, offsets, co, cs, nr, nf)
    return b


def count_filtered(batch, n_threads=0, lib=None):
    """rh_count_filtered: the reader's pA filter as per-read counts (what ReadBatch.n_filtered wants) of a host batch."""
    l = lib or globals()["lib"]()
    out = np.zeros(max(batch.n_reads, 1), dtype=np.uint32)
    if l.rh_count_filtered(C.byref(batch), ptr(out), n_threads) != 0:
        raise RuntimeError(last_error(l))
    return out[: batch.n_reads]


_lib = None


def _declare(lib):
    u64, u32, i32, vp, cp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_char_p
    P = C.POINTER
    sig = {
        "rh_last_error": (cp, []), "rh_version": (cp, []), "rh_device_count": (i32, []),
        "rh_idxopt_init": (None, [P(IdxOpt)]), "rh_mapopt_init": (None, [P(MapOpt)]),
        "rh_set_preset": (i32, [cp, P(IdxOpt), P(MapOpt)]),
        "rh_index_load": (vp, [cp]), "rh_index_build": (vp, [cp, cp, P(IdxOpt), cp, i32]),
        "rh_index_destroy": (None, [vp]), "rh_mapopt_update": (None, [P(MapOpt), vp]),
        "rh_index_n_seq": (u32, [vp]), "rh_index_seq_name": (cp, [vp, u32]), "rh_index_seq_len": (u32, [vp, u32]),
        "rh_index_params": (None, [vp, P(IdxOpt)]), "rh_index_n_keys": (u64, [vp]), "rh_index_n_positions": (u64, [vp]),
        "rh_index_get": (vp, [vp, u64, P(C.c_int)]),
        "rh_ctx_create": (i32, [P(vp), i32]), "rh_ctx_destroy": (None, [vp]),
        "rh_index_upload": (i32, [vp, vp]),
        "rh_index_build_device": (vp, [vp, u32, P(cp), P(cp), vp, cp, P(IdxOpt), i32]),
        "rh_index_build_device_fasta": (vp, [vp, cp, cp, P(IdxOpt), i32]),
        "rh_index_build_signals_device": (vp, [vp, P(ReadBatch), P(cp), cp, P(IdxOpt), P(MapOpt)]),
        "rh_index_name_ranks": (i32, [vp, P(cp), u32, vp, vp]), "rh_index_set_target_ranks": (i32, [vp, vp, u32]),
        "rh_map_batch_multi": (i32, [vp, P(MapOpt), P(ReadBatch), vp, u64, vp, P(u64)]),
        "rh_index_download": (i32, [vp, vp, i32]), "rh_index_write": (i32, [vp, cp]),
        "rh_synth_genome": (i32, [P(SynthCfg), u32, vp, i32]),
        "rh_index_device_blob": (i32, [vp, P(vp), P(u64), vp]),
        "rh_index_adopt_blob": (i32, [vp, vp, vp, u64, vp, i32]),
        "rh_map_max_records": (u64, [P(ReadBatch), P(MapOpt)]),
        "rh_map_batch": (i32, [vp, P(MapOpt), P(ReadBatch), vp, u64, P(u64)]),
        "rh_map_submit": (i32, [vp, P(MapOpt), P(ReadBatch), vp, u64, P(Ticket)]), "rh_map_wait": (i32, [vp, Ticket, P(u64)]),
        "rh_read_batch_to_host": (i32, [vp, P(ReadBatch), vp, vp, vp, vp]),
        "rh_pinned_alloc": (vp, [C.c_size_t]), "rh_pinned_free": (None, [vp]), "rh_index_bcast": (i32, [P(vp), i32]),
        "rh_index_bcast_path": (i32, []), "rh_rccl_selftest": (i32, [vp]),
        "rh_map_last_stats": (i32, [vp, P(MapStats)]), "rh_stage_name": (cp, [i32]),
        "rh_events_batch": (i32, [vp, P(MapOpt), P(ReadBatch), u32, vp, u64, vp, vp]),
        "rh_sketch_batch": (i32, [vp, u32, vp, vp, vp, u64, vp]),
        "rh_seed_batch": (i32, [vp, P(MapOpt), u32, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
        "rh_chain_batch": (i32, [vp, P(MapOpt), u32, vp, vp, vp, u64, vp, vp, u64, vp, vp]),
        "rh_regions_batch": (i32, [vp, P(MapOpt), u32, vp, vp, vp, vp, vp, vp, u64, vp]),
        "rh_sort128x_batch": (i32, [vp, u32, vp, vp]), "rh_sort128x_any_batch": (i32, [vp, u32, vp, vp, vp]), "rh_sort128x_packed_batch": (i32, [vp, u32, vp, vp, u32, u32, i32]),
        "rh_paf_format": (i32, [vp, P(MapRecord), cp, C.c_double, cp, C.c_size_t]),
        "rh_reads_load": (vp, [cp]), "rh_reads_destroy": (None, [vp]), "rh_reads_n": (u32, [vp]),
        "rh_reads_name": (cp, [vp, u32]), "rh_reads_batch": (i32, [vp, P(ReadBatch)]), "rh_reads_pinned": (i32, [vp]),
        "rh_count_filtered": (i32, [P(ReadBatch), vp, i32]),
        "rh_reads_write": (i32, [cp, u32, P(cp), vp, vp, C.c_double, C.c_double, C.c_double]),
        "rh_reads_write_blow5": (i32, [cp, u32, P(cp), vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, i32]),
        "rh_synth_cfg_init": (None, [P(SynthCfg)]), "rh_synth_write_model": (i32, [P(SynthCfg), cp]), "rh_synth_write_model_k": (i32, [P(SynthCfg), cp, i32]),
        "rh_synth_write_fasta": (i32, [P(SynthCfg), cp]),
        "rh_synth_reads": (i32, [P(SynthCfg), cp, u64, u32, vp, vp, i32]),
        "rh_synth_reads_device": (i32, [vp, P(SynthCfg), cp, u64, u32, P(ReadBatch)]),
        "rh_index_copy_blob": (i32, [vp, vp]),
        "rh_synth_origin": (i32, [P(SynthCfg), u64, P(u32), P(u32), P(u32), P(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the .so does not export what the header declares
        fn.restype, fn.argtypes = res, args
    return sig


EXPORTS = None


def load(path):
    """dlopen a build of the C ABI and attach the prototypes of include/rawhash_amd.h (fails on a missing symbol)."""
    global EXPORTS
    l = C.CDLL(path)
    EXPORTS = _declare(l)
    return l


def lib():
    """The product shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU/Python fallback for the mapping path)")
        _lib = load(LIB_PATH)
    return _lib


def last_error(l=None):
    return (l or lib()).rh_last_error().decode()

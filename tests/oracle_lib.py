"""ctypes loader for the CPU oracle (oracle/_build/librh_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from rawhash_amd._capi import IdxOpt, MapOpt, MapRecord, ReadBatch, RECORD, MM128, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "librh_oracle.so")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")
REF_HARNESS_V4 = os.path.join(ORACLE_DIR, "_ref", "ref_harness_v4")   # the same sources with -march=x86-64-v4 (the stock -march=native stand-in): timing only

_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def have_reference():
    """The reference harness can only be (re)built where /root/reference exists; the prebuilt binary travels."""
    return os.path.exists(REF_HARNESS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "rh_oracle.c")):
            build_oracle()
        l = C.CDLL(ORACLE_SO)
        u64, u32, i32, vp, cp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_char_p
        P = C.POINTER
        sig = {
            "ro_index_load": (vp, [cp]), "ro_index_free": (None, [vp]), "ro_index_get": (vp, [vp, u64, P(C.c_int)]),
            "ro_mapopt_update": (None, [P(MapOpt), vp]), "ro_index_n_seq": (u32, [vp]),
            "ro_index_seq_name": (cp, [vp, u32]), "ro_index_seq_len": (u32, [vp, u32]),
            "ro_index_params": (None, [vp, P(IdxOpt)]), "ro_index_n_keys": (u64, [vp]),
            "ro_index_list": (u64, [vp, vp, vp, u64]),
            "ro_idxopt_init": (None, [P(IdxOpt)]), "ro_mapopt_init": (None, [P(MapOpt)]),
            "ro_set_preset": (i32, [cp, P(IdxOpt), P(MapOpt)]),
            "ro_events_batch": (i32, [P(MapOpt), P(ReadBatch), u32, vp, u64, vp, vp]),
            "ro_sketch_batch": (i32, [vp, u32, vp, vp, vp, u64, vp]),
            "ro_seed_batch": (i32, [vp, P(MapOpt), u32, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
            "ro_chain_batch": (i32, [vp, P(MapOpt), u32, vp, vp, vp, u64, vp, vp, u64, vp, vp]),
            "ro_regions_batch": (i32, [P(MapOpt), u32, vp, vp, vp, vp, vp, vp, vp, u64, vp]),
            "ro_sort128x_batch": (i32, [u32, vp, vp]),
            "ro_map_batch": (i32, [vp, P(MapOpt), P(ReadBatch), vp, vp, u64, P(u64), i32]),
            "ro_last_counters": (None, [vp]),
            "ro_paf_format": (i32, [vp, P(MapRecord), cp, C.c_double, cp, C.c_size_t]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def preset(name):
    io, mo = IdxOpt(), MapOpt()
    lib().ro_set_preset(None, C.byref(io), C.byref(mo))
    if name not in (None, "default"):
        assert lib().ro_set_preset(name.encode(), C.byref(io), C.byref(mo)) == 0
    return io, mo


class OracleIndex:
    def __init__(self, path):
        self.h = lib().ro_index_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle cannot load {path}")

    def close(self):
        if self.h:
            lib().ro_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def names(self):
        return [lib().ro_index_seq_name(self.h, i).decode() for i in range(lib().ro_index_n_seq(self.h))]

    def listing(self):
        n = lib().ro_index_n_keys(self.h)
        hs = np.zeros(n, dtype=np.uint64)
        cs = np.zeros(n, dtype=np.uint32)
        lib().ro_index_list(self.h, ptr(hs), ptr(cs), n)
        return hs, cs

    def get(self, h):
        n = C.c_int(0)
        p = lib().ro_index_get(self.h, int(h), C.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()


def map_batch(ix, mo, batch, names=None, n_threads=1, max_rec_per_read=1):
    cap = batch.n_reads * max_rec_per_read + 16
    out = np.zeros(cap, dtype=RECORD)
    n_out = C.c_uint64(0)
    arr = None
    if names is not None:
        arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else n for n in names])
    rc = lib().ro_map_batch(ix.h, C.byref(mo), C.byref(batch), arr, ptr(out), cap, C.byref(n_out), n_threads)
    assert rc == 0
    return out[: n_out.value]


def paf_lines(ix, recs, names, mt_ms=0.0):
    buf = C.create_string_buffer(4096)
    lines = []
    for r in recs:
        rec = MapRecord.from_buffer_copy(r.tobytes())
        nm = names[int(r["read_idx"])]
        n = lib().ro_paf_format(ix.h, C.byref(rec), nm.encode() if isinstance(nm, str) else nm, mt_ms, buf, 4096)
        assert n >= 0
        if n:
            lines.append(buf.value.decode())
    return lines


def mask_ind(data):
    """.ind bytes with the 16 bytes of heap pointers the reference leaks into the header (raw fwrite of ri_pore_t,
    rindex.c:557, file offset 46..61) zeroed."""
    b = bytearray(data)
    b[46:62] = bytes(16)
    return bytes(b)


def strip_mt(line):
    """Drop the wall-clock mt:f: tag (excluded from parity)."""
    return "\t".join(f for f in line.rstrip("\n").split("\t") if not f.startswith("mt:f:"))

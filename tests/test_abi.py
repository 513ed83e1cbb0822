"""The C-ABI library loads, exports every symbol include/rawhash_amd.h declares, and has no CPU fallback."""
import ctypes as C
import os
import re

import pytest

from rawhash_amd import _capi
from rawhash_amd.api import Context, MapOptions, RhError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(product_lib):
    hdr = open(os.path.join(ROOT, "include", "rawhash_amd.h")).read()
    declared = set(re.findall(r"RH_API[^;(]*?\b(rh_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(product_lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_capi.EXPORTS), "ctypes prototypes and header out of sync"


def test_version_and_options(product_lib):
    assert b"rawhash_amd" in product_lib.rh_version()
    o = MapOptions("fast")
    assert o.mo.min_mapq == 5 and o.mo.min_chaining_score == 10 and abs(o.io.fine_range - 0.6) < 1e-6
    o = MapOptions("faster")
    assert (o.io.e, o.io.w, o.mo.max_num_chunk) == (11, 3, 5)
    o = MapOptions("ava")
    assert o.io.w == 3 and o.mo.bw == 5000 and o.mo.pri_ratio == 0.0
    with pytest.raises(RhError):
        MapOptions("no-such-preset")


def test_presets_match_oracle_tables(product_lib):
    import oracle_lib as O
    for p in [None, "sensitive", "fast", "faster", "viral", "ava", "ava-viral", "ava-sensitive", "ava-large"]:
        a = MapOptions(p)
        io, mo = O.preset(p)
        assert bytes(a.io) == bytes(io) and bytes(a.mo) == bytes(mo), p


@pytest.mark.skipif(_capi.lib().rh_device_count() > 0, reason="a GPU is visible")
def test_no_cpu_fallback(product_lib):
    """Without a GPU the compute entry points must fail loudly, not fall back."""
    with pytest.raises(RhError, match="no HIP device"):
        Context(0)


def test_blow5_reader_round_trip(tmp_path, make_workload, product_lib):
    """BLOW5 records (uncompressed, zlib, zstd) x signals (raw, svb-zd) decode into the same raw int16 batch + calibration as the
    RHR1 container; a truncated file and an unknown compression code are errors, not crashes."""
    import numpy as np
    from rawhash_amd.api import Reads, RhError, write_blow5
    w = make_workload(n_reads=24)
    cfg = w.wl.cfg
    for rec in ("none", "zlib", "zstd"):
        for svb in (False, True):
            p = str(tmp_path / f"reads_{rec}_{int(svb)}.blow5")
            write_blow5(w.reads, p, cfg.digitisation, cfg.range, cfg.offset, records=rec, svb_zd=svb, lib=product_lib)
            r = Reads.load(p, lib=product_lib)
            assert r.names == w.reads.names
            assert np.array_equal(r.samples, w.reads.samples) and np.array_equal(r.offsets, w.reads.offsets)
            assert np.array_equal(r.cal_offset, w.reads.cal_offset) and np.array_equal(r.cal_scale, w.reads.cal_scale)
    assert os.path.getsize(str(tmp_path / "reads_none_1.blow5")) < 0.75 * os.path.getsize(str(tmp_path / "reads_none_0.blow5"))   # svb-zd does compress
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.blow5")
    open(cut, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(RhError):
        Reads.load(cut, lib=product_lib)
    bad = bytearray(raw); bad[9] = 3                     # no such record compression
    open(cut, "wb").write(bytes(bad))
    with pytest.raises(RhError, match="record compression"):
        Reads.load(cut, lib=product_lib)
    bad = bytearray(raw); bad[10] = 7                    # no such signal compression
    open(cut, "wb").write(bytes(bad))
    with pytest.raises(RhError, match="signal compression"):
        Reads.load(cut, lib=product_lib)


def _svb_zd_block(x):
    """StreamVByte (Lemire, 32-bit) of the zig-zag first differences of int16 samples, written from the published format - an
    implementation independent of the library's: ceil(n/4) control bytes (2 bits per value = bytes - 1, first value lowest), then data."""
    import struct
    ctl, dat, prev = bytearray((len(x) + 3) // 4), bytearray(), 0
    for i, v in enumerate(int(t) for t in x):
        d = v - prev; prev = v
        z = ((d << 1) ^ (d >> 31)) & 0xFFFFFFFF
        nb = 1 if z < 1 << 8 else 2 if z < 1 << 16 else 3 if z < 1 << 24 else 4
        ctl[i >> 2] |= (nb - 1) << ((i & 3) * 2)
        dat += z.to_bytes(4, "little")[:nb]
    return struct.pack("<I", len(x)) + bytes(ctl) + bytes(dat)


def test_blow5_hand_assembled_fixtures(tmp_path, product_lib):
    """BLOW5 files put together byte by byte from the published format (file header, records with auxiliary fields after the signal,
    end marker) - not by the library's writer: raw and svb-zd signals (compressed byte count as u64, as u32, absent), uncompressed,
    zlib (Python's zlib) and zstd (libzstd through ctypes) records; extreme sample values, an empty read, a one-sample read."""
    import ctypes, struct, zlib
    import numpy as np
    from rawhash_amd.api import Reads
    rng = np.random.default_rng(5)
    reads = [("read-a", np.array([0, 1, -1, 32767, -32768, 300, 299, 301, -5000, 12345], dtype=np.int16), 8192.0, 6.0, 1402.882),
             ("b", rng.integers(400, 700, size=4001).astype(np.int16), 2048.0, -3.5, 748.58),
             ("empty", np.zeros(0, dtype=np.int16), 8192.0, 0.0, 1400.0),
             ("one", np.array([-7], dtype=np.int16), 8192.0, 10.0, 1467.61),
             ("walk", np.cumsum(rng.integers(-40, 41, size=1777)).astype(np.int16), 8192.0, 4.0, 1300.5)]
    try:
        zs = ctypes.CDLL("libzstd.so.1")
        zs.ZSTD_compress.restype = ctypes.c_size_t; zs.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        zs.ZSTD_compressBound.restype = ctypes.c_size_t; zs.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    except OSError:
        zs = None

    def zstd_frame(b):
        cap = zs.ZSTD_compressBound(len(b)); out = ctypes.create_string_buffer(cap)
        n = zs.ZSTD_compress(out, cap, b, len(b), 5)
        return out.raw[:n]

    aux = struct.pack("<Bdi", 7, 1234.5, -99)            # auxiliary fields follow the signal; a reader must not need them
    n_files = 0
    for rec_comp in (0, 1, 2):
        if rec_comp == 2 and zs is None:
            continue
        for sig_comp, cnt_width in ((0, 0), (1, 8), (1, 4), (1, 0)):
            text = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n"
            f = bytearray(b"BLOW5\1" + bytes([0, 2, 0, rec_comp, sig_comp]) + struct.pack("<I", 1))
            f += bytes(64 - len(f)) + struct.pack("<I", len(text)) + text
            for name, x, dig, off, ran in reads:
                body = struct.pack("<H", len(name)) + name.encode() + struct.pack("<I4dQ", 0, dig, off, ran, 4000.0, len(x))
                if sig_comp == 0:
                    body += x.tobytes()
                else:
                    blk = _svb_zd_block(x)
                    body += (struct.pack("<Q", len(blk)) if cnt_width == 8 else struct.pack("<I", len(blk)) if cnt_width == 4 else b"") + blk
                body += aux
                rec = body if rec_comp == 0 else zlib.compress(body) if rec_comp == 1 else zstd_frame(body)
                f += struct.pack("<Q", len(rec)) + rec
            f += b"5WOLB"
            p = str(tmp_path / f"fx_{rec_comp}_{sig_comp}_{cnt_width}.blow5")
            open(p, "wb").write(bytes(f))
            r = Reads.load(p, lib=product_lib)
            assert r.names == [q[0] for q in reads]
            for i, (name, x, dig, off, ran) in enumerate(reads):
                assert np.array_equal(r.samples[int(r.offsets[i]):int(r.offsets[i + 1])], x), (p, name)
                assert r.cal_offset[i] == off and r.cal_scale[i] == np.float32(ran / dig)
            n_files += 1
    assert n_files >= 8

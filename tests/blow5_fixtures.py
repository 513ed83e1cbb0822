"""BLOW5 files assembled byte by byte from the published format (SLOW5 specification 1.0.0) - independent of the library's own writer:
Python's zlib, libzstd through ctypes, and a StreamVByte encoder written here.  Used by the CPU suite (tests/test_abi.py) and by the
GPU suite (tests/test_gpu_parity.py::test_blow5_hand_assembled_file_maps)."""
import ctypes
import struct
import zlib

import numpy as np


def svb_zd_block(x):
    """StreamVByte (Lemire, 32-bit) of the zig-zag first differences of int16 samples, written from the published format - an
    implementation independent of the library's: ceil(n/4) control bytes (2 bits per value = bytes - 1, first value lowest), then data."""
    ctl, dat, prev = bytearray((len(x) + 3) // 4), bytearray(), 0
    for i, v in enumerate(int(t) for t in x):
        d = v - prev; prev = v
        z = ((d << 1) ^ (d >> 31)) & 0xFFFFFFFF
        nb = 1 if z < 1 << 8 else 2 if z < 1 << 16 else 3 if z < 1 << 24 else 4
        ctl[i >> 2] |= (nb - 1) << ((i & 3) * 2)
        dat += z.to_bytes(4, "little")[:nb]
    return struct.pack("<I", len(x)) + bytes(ctl) + bytes(dat)


def load_zstd():
    try:
        zs = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    zs.ZSTD_compress.restype = ctypes.c_size_t; zs.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    zs.ZSTD_compressBound.restype = ctypes.c_size_t; zs.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    return zs


def zstd_frame(zs, b):
    cap = zs.ZSTD_compressBound(len(b)); out = ctypes.create_string_buffer(cap)
    n = zs.ZSTD_compress(out, cap, b, len(b), 5)
    return out.raw[:n]


def header(rec_comp, sig_comp):
    text = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n"
    f = bytearray(b"BLOW5\1" + bytes([0, 2, 0, rec_comp, sig_comp]) + struct.pack("<I", 1))
    f += bytes(64 - len(f)) + struct.pack("<I", len(text)) + text
    return f


def sample_reads():
    rng = np.random.default_rng(5)
    return [("read-a", np.array([0, 1, -1, 32767, -32768, 300, 299, 301, -5000, 12345], dtype=np.int16), 8192.0, 6.0, 1402.882),
            ("b", rng.integers(400, 700, size=4001).astype(np.int16), 2048.0, -3.5, 748.58),
            ("empty", np.zeros(0, dtype=np.int16), 8192.0, 0.0, 1400.0),
            ("one", np.array([-7], dtype=np.int16), 8192.0, 10.0, 1467.61),
            ("walk", np.cumsum(rng.integers(-40, 41, size=1777)).astype(np.int16), 8192.0, 4.0, 1300.5)]


def assemble(reads, rec_comp, sig_comp, cnt_width=8, zs=None):
    """reads: (name, int16 samples, digitisation, offset, range); rec_comp 0 none / 1 zlib / 2 zstd; sig_comp 0 raw / 1 svb-zd with the
    compressed byte count written as u64 (cnt_width 8: the format), u32 (4) or left out (0) - the last two only to see them refused."""
    aux = struct.pack("<Bdi", 7, 1234.5, -99)            # auxiliary fields follow the signal; a reader must not need them
    f = header(rec_comp, sig_comp)
    for name, x, dig, off, ran in reads:
        body = struct.pack("<H", len(name)) + name.encode() + struct.pack("<I4dQ", 0, dig, off, ran, 4000.0, len(x))
        if sig_comp == 0:
            body += np.asarray(x, dtype=np.int16).tobytes()
        else:
            blk = svb_zd_block(x)
            body += (struct.pack("<Q", len(blk)) if cnt_width == 8 else struct.pack("<I", len(blk)) if cnt_width == 4 else b"") + blk
        body += aux
        rec = body if rec_comp == 0 else zlib.compress(body) if rec_comp == 1 else zstd_frame(zs, body)
        f += struct.pack("<Q", len(rec)) + rec
    f += b"5WOLB"
    return bytes(f)

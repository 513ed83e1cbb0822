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
    """BLOW5 records (uncompressed and zlib) decode into the same raw int16 batch + calibration as the RHR1 container; a
    truncated file and unsupported compression are errors, not crashes."""
    import numpy as np
    from rawhash_amd.api import Reads, RhError, write_blow5
    w = make_workload(n_reads=24)
    cfg = w.wl.cfg
    for z in (False, True):
        p = str(tmp_path / f"reads_{int(z)}.blow5")
        write_blow5(w.reads, p, cfg.digitisation, cfg.range, cfg.offset, zlib_records=z, lib=product_lib)
        r = Reads.load(p, lib=product_lib)
        assert r.names == w.reads.names
        assert np.array_equal(r.samples, w.reads.samples) and np.array_equal(r.offsets, w.reads.offsets)
        assert np.array_equal(r.cal_offset, w.reads.cal_offset) and np.array_equal(r.cal_scale, w.reads.cal_scale)
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.blow5")
    open(cut, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(RhError):
        Reads.load(cut, lib=product_lib)
    bad = bytearray(raw); bad[9] = 2                     # zstd records
    open(cut, "wb").write(bytes(bad))
    with pytest.raises(RhError, match="zstd"):
        Reads.load(cut, lib=product_lib)

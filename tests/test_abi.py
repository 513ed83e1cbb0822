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


def test_count_filtered_is_the_readers_l_sig(make_workload, product_lib):
    """rh_count_filtered (what rh_read_batch_t::n_filtered wants) = the sl:i tag the oracle prints (ri_read_sig's l_sig, rsig.c:496-503), with
    samples outside 30 .. 200 pA in the signal, in both ingest arithmetics, for any thread count; the BLOW5 / RHR loaders deliver it with the batch."""
    import numpy as np
    import oracle_lib as O
    from rawhash_amd.api import Reads
    w = make_workload(n_reads=40, n_samples=9_000)
    rng = np.random.default_rng(11)
    smp = w.reads.samples.copy()
    idx = rng.integers(0, len(smp), size=len(smp) // 7)
    smp[idx] = rng.choice(np.array([-32768, -5000, 32767], dtype=np.int16), size=len(idx))
    for fast5 in (False, True):
        reads = Reads(smp, w.reads.offsets, w.reads.names, w.reads.cal_offset, w.reads.cal_scale, fast5=fast5)
        oix, mo = w.oracle()
        recs = O.map_batch(oix, mo, reads.batch(), n_threads=2)
        want = np.array([r["tag_sl"] for r in recs], dtype=np.uint32)
        for nt in (1, 3, 0):
            got = _capi.count_filtered(reads.batch(), n_threads=nt, lib=product_lib)
            assert (got == want).all(), (fast5, nt)
        assert want.sum() < len(smp)


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


def test_blow5_hand_assembled_fixtures(tmp_path, product_lib):
    """BLOW5 files put together byte by byte from the published format (tests/blow5_fixtures.py: file header, records with auxiliary
    fields after the signal, end marker) - not by the library's writer: raw and svb-zd signals, uncompressed, zlib (Python's zlib) and
    zstd (libzstd through ctypes) records; extreme sample values, an empty read, a one-sample read.  The svb-zd signal has ONE accepted
    layout (u64 compressed byte count | u32 values | StreamVByte block): a u32 count or no count is refused, not guessed at."""
    import numpy as np
    import blow5_fixtures as B
    from rawhash_amd.api import Reads, RhError
    reads = B.sample_reads()
    zs = B.load_zstd()
    n_files = 0
    for rec_comp in (0, 1, 2):
        if rec_comp == 2 and zs is None:
            continue
        for sig_comp, cnt_width in ((0, 0), (1, 8), (1, 4), (1, 0)):
            p = str(tmp_path / f"fx_{rec_comp}_{sig_comp}_{cnt_width}.blow5")
            open(p, "wb").write(B.assemble(reads, rec_comp, sig_comp, cnt_width, zs))
            if sig_comp == 1 and cnt_width != 8:
                with pytest.raises(RhError, match="svb-zd"):
                    Reads.load(p, lib=product_lib)
                continue
            r = Reads.load(p, lib=product_lib)
            assert r.names == [q[0] for q in reads]
            for i, (name, x, dig, off, ran) in enumerate(reads):
                assert np.array_equal(r.samples[int(r.offsets[i]):int(r.offsets[i + 1])], x), (p, name)
                assert r.cal_offset[i] == off and r.cal_scale[i] == np.float32(ran / dig)
            n_files += 1
    assert n_files >= 4


def test_slow5_text_hand_assembled_fixture(tmp_path, product_lib):
    """A text .slow5 file written line by line from the published format (what `slow5tools view` prints: version / read-group header lines, the
    column types, the column names, one tab-separated line per read with raw_signal as a comma-separated list, auxiliary columns behind it) -
    slow5_open takes the text form like the binary one (rsig.c:170-207).  The same reads as the BLOW5 fixtures; columns are found by NAME (a
    second file has them in another order); malformed records are refused with the line number."""
    import numpy as np
    import blow5_fixtures as B
    from rawhash_amd.api import Reads, RhError
    reads = B.sample_reads()
    hdr = ["#slow5_version\t0.2.0", "#num_read_groups\t1", "@asic_id\t420", "@exp_start_time\t2024-01-01T00:00:00Z", "@sample_frequency\t4000"]

    def text(order, aux=True, crlf=False):
        cols = {"read_id": "char*", "read_group": "uint32_t", "digitisation": "double", "offset": "double", "range": "double", "sampling_rate": "double",
                "len_raw_signal": "uint64_t", "raw_signal": "int16_t*"}
        names = list(order) + (["channel_number", "read_number"] if aux else [])
        types = [cols[c] for c in order] + (["char*", "int32_t"] if aux else [])
        lines = hdr + ["#" + "\t".join(types), "#" + "\t".join(names)]
        for i, (name, x, dig, off, ran) in enumerate(reads):
            val = {"read_id": name, "read_group": "0", "digitisation": repr(float(dig)), "offset": repr(float(off)), "range": repr(float(ran)), "sampling_rate": "4000",
                   "len_raw_signal": str(len(x)), "raw_signal": ",".join(str(int(v)) for v in x) if len(x) else "."}
            lines.append("\t".join([val[c] for c in order] + ([str(100 + i), str(i)] if aux else [])))
        return ("\r\n" if crlf else "\n").join(lines) + "\n"

    std = ["read_id", "read_group", "digitisation", "offset", "range", "sampling_rate", "len_raw_signal", "raw_signal"]
    other = ["read_id", "len_raw_signal", "raw_signal", "range", "read_group", "offset", "sampling_rate", "digitisation"]
    for k, (order, aux, crlf) in enumerate(((std, True, False), (std, False, True), (other, True, False))):
        p = str(tmp_path / f"fx{k}.slow5")
        open(p, "w", newline="").write(text(order, aux, crlf))
        r = Reads.load(p, lib=product_lib)
        assert r.names == [q[0] for q in reads]
        for i, (name, x, dig, off, ran) in enumerate(reads):
            assert np.array_equal(r.samples[int(r.offsets[i]):int(r.offsets[i + 1])], x), (p, name)
            assert r.cal_offset[i] == off and r.cal_scale[i] == np.float32(ran / dig)
    good = text(std).splitlines()
    for bad, msg in ((good[:7] + [good[7].replace("\t" + str(len(reads[0][1])) + "\t", "\t" + str(len(reads[0][1]) + 1) + "\t", 1)] + good[8:], "len_raw_signal"),
                     (good[:6] + good[7:], "column names"),
                     (good[:7] + [good[7].rsplit("\t", 3)[0] + "\t1,2,x,4\t1\t1"] + good[8:], "comma-separated")):
        p = str(tmp_path / "bad.slow5")
        open(p, "w").write("\n".join(bad) + "\n")
        with pytest.raises(RhError, match=msg):
            Reads.load(p, lib=product_lib)
    # a record cut short BEHIND raw_signal (primary columns follow it in this order) and a calibration field that is not a number: errors, not reads
    # with offset / range 0
    og = text(other, aux=False).splitlines()
    for bad, msg in ((og[:7] + [og[7].rsplit("\t", 3)[0]] + og[8:], "fewer fields"),
                     (og[:7] + [og[7].rsplit("\t", 1)[0] + "\t8192.0x"] + og[8:], "not a number")):
        p = str(tmp_path / "bad2.slow5")
        open(p, "w").write("\n".join(bad) + "\n")
        with pytest.raises(RhError, match=msg):
            Reads.load(p, lib=product_lib)


def test_blow5_hostile_lengths_fail_before_allocating(tmp_path, product_lib):
    """Length fields are checked against what the record / file holds BEFORE a buffer grows for them: a record that declares 2^32 - 1
    samples, an svb-zd block that declares more values than it has bytes, a zstd frame that declares gigabytes - errors within
    milliseconds, not allocations of (page-locked) gigabytes."""
    import struct, time
    import blow5_fixtures as B
    from rawhash_amd.api import Reads, RhError
    zs = B.load_zstd()
    t0 = time.time()
    for sig_comp in (0, 1):
        f = B.header(0, sig_comp)
        body = struct.pack("<H", 1) + b"x" + struct.pack("<I4dQ", 0, 8192.0, 0.0, 1400.0, 4000.0, (1 << 32) - 1)
        body += (struct.pack("<QI", 8, (1 << 32) - 1) + bytes(4)) if sig_comp else bytes(16)
        f += struct.pack("<Q", len(body)) + body + b"5WOLB"
        p = str(tmp_path / f"hostile_{sig_comp}.blow5")
        open(p, "wb").write(bytes(f))
        with pytest.raises(RhError, match="signal"):
            Reads.load(p, lib=product_lib)
    if zs is not None:      # a valid frame header announcing 8 GiB of content (zstd frame format: magic, descriptor 0xE0 = single segment + 8-byte size)
        frame = struct.pack("<IB", 0xFD2FB528, 0xE0) + struct.pack("<Q", 8 << 30) + bytes(8)
        f = B.header(2, 0) + struct.pack("<Q", len(frame)) + frame + b"5WOLB"
        p = str(tmp_path / "hostile_zstd.blow5")
        open(p, "wb").write(bytes(f))
        with pytest.raises(RhError, match="zstd"):
            Reads.load(p, lib=product_lib)
    # RHR1: a name length beyond the file
    p = str(tmp_path / "hostile.rhr")
    open(p, "wb").write(b"RHR1" + struct.pack("<II", 1, 0xFFFFFFF0) + b"abc")
    with pytest.raises(RhError, match="truncated"):
        Reads.load(p, lib=product_lib)
    assert time.time() - t0 < 5.0


def _walk_isa_hazards(asm_text, func):
    """In the device assembly of `func`: for every ds_read_u8 issued by inline asm (the token walker's asynchronous read of the next
    digit), follow every path until an s_waitcnt that waits for LDS (lgkmcnt(0)) and report instructions that touch the register being loaded."""
    import re
    lines = asm_text.splitlines()
    beg = next(i for i, l in enumerate(lines) if l.startswith(func + ":"))
    end = next(i for i in range(beg, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins, labels, in_asm = [], {}, False
    for l in lines[beg + 1:end]:
        t = l.split(";")[0].strip() if not l.strip().startswith(";;#") else l.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB[0-9_]+):", t)
            if m: labels[m.group(1)] = len(ins)
            continue
        ins.append((t, in_asm))
    hazards, n_reads = [], 0
    for i, (t, in_asm) in enumerate(ins):
        m = re.match(r"ds_read_u8\s+(v\d+),", t)
        if not (m and in_asm):
            continue
        n_reads += 1
        reg = re.compile(r"\b" + m.group(1) + r"\b")
        seen, stack = set(), [i + 1]
        while stack:
            j = stack.pop()
            while j < len(ins) and j not in seen:
                seen.add(j)
                u = ins[j][0]
                if u.startswith("s_waitcnt") and ("lgkmcnt(0)" in u or u.split()[-1] in ("0", "0x0")):
                    break
                if reg.search(u):
                    hazards.append((t, u)); break
                b = re.match(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", u)
                if b and b.group(1) in labels:
                    stack.append(labels[b.group(1)])
                    if u.startswith("s_branch"): break
                if u.startswith("s_endpgm"): break
                j += 1
    return n_reads, hazards


def test_token_walker_isa_keeps_lds_read_private(tmp_path):
    """rh_tok_advance (rh_gpu.h) issues an LDS read the compiler does not know about; the value is defined only after rh_lds_wait.  The
    generated gfx950 code of the default walker must not touch the loaded register between the read and the wait on any path - checked
    on the compiler's output, so a toolchain that starts copying or spilling that register fails HERE instead of silently mis-sorting."""
    import shutil, subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this environment")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "bs.s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-x", "hip", "-I", os.path.join(root, "include"),
           "--cuda-device-only", "-S", "-w", "-o", out, os.path.join(root, "rawhash_amd", "csrc", "rh_bigsort.hip")] + os.environ.get("RH_HIPCC_EXTRA", "").split()
    subprocess.run(cmd, check=True)
    n_reads, hazards = _walk_isa_hazards(open(out).read(), "_Z13k_bs_walk_tokILi1ELi1EEv6bs_ctxjj")
    assert n_reads >= 2, "the walker's inline-asm LDS reads were not found: has the kernel been renamed?"
    assert not hazards, f"the register of an in-flight LDS read is touched before the wait: {hazards[:3]}"

"""A compiled C++ consumer of the C ABI (tests/cabi/rawhash2_step1.cpp = the INTEGRATION.md binding as a program): it must
build against include/rawhash_amd.h alone with g++, and on the GPU print the reference's golden PAF with two mini-batches in
flight (rh_map_submit / rh_map_wait)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "rawhash2_step1.cpp")


SRC_AVA = os.path.join(ROOT, "tests", "cabi", "rawhash2_ava.cpp")
SRC_MULTI = os.path.join(ROOT, "tests", "cabi", "rawhash2_multigpu.cpp")


def build_consumer(out_dir, product_lib, src=SRC):
    exe = os.path.join(str(out_dir), os.path.splitext(os.path.basename(src))[0])
    libdir = os.path.join(ROOT, "rawhash_amd")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                    "-L", libdir, "-lrawhash_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def test_consumer_builds_with_plain_gxx(tmp_path, product_lib):
    exe = build_consumer(tmp_path, product_lib)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


def test_rawsamble_consumer_builds_with_plain_gxx(tmp_path, product_lib):
    exe = build_consumer(tmp_path, product_lib, SRC_AVA)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


@pytest.mark.gpu
def test_rawsamble_consumer_prints_the_golden_paf_and_ind(tmp_path, product_lib):
    """INTEGRATION.md section 3 as a program: `-x ava -d out.ind reads` then the all-vs-all run, both through the C ABI from C++:
    the .ind is the reference's byte for byte, the overlaps are the reference's PAF."""
    import hashlib
    import golden
    import oracle_lib as O
    from rawhash_amd import strip_mt
    exe = build_consumer(tmp_path, product_lib, SRC_AVA)
    for case in golden.ava_cases():
        w = golden.build_ava_case(case, tmp_path / case["name"], product_lib)
        ind = os.path.join(str(tmp_path), case["name"] + ".ind")
        p = subprocess.run([exe, w.preset, w.model, w.rhr, ind], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert [strip_mt(x) for x in p.stdout.splitlines()] == golden.expected_paf(case)
        assert hashlib.sha256(O.mask_ind(open(ind, "rb").read())).hexdigest() == case["ind_sha256"]


@pytest.mark.gpu
def test_consumer_prints_the_golden_paf(tmp_path, product_lib):
    import golden
    from rawhash_amd import strip_mt
    exe = build_consumer(tmp_path, product_lib)
    case = [c for c in golden.cases() if c["name"] == "small_sensitive"][0]
    w = golden.build_case(case, tmp_path / "wl", product_lib)
    rhr = os.path.join(str(tmp_path), "reads.rhr")
    w.reads.write(rhr, w.wl.cfg.digitisation, w.wl.cfg.range, w.wl.cfg.offset)
    for per in ("37", "1000"):      # five mini-batches with two in flight / one batch
        p = subprocess.run([exe, "sensitive", w.ind, rhr, per], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert [strip_mt(x) for x in p.stdout.splitlines()] == golden.expected_paf(case)


def test_multigpu_consumer_builds_with_plain_gxx(tmp_path, product_lib):
    exe = build_consumer(tmp_path, product_lib, SRC_MULTI)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


@pytest.mark.gpu
def test_multigpu_consumer_prints_the_golden_paf(tmp_path, product_lib):
    """INTEGRATION.md section 4 as a program: one process, one context + host thread per GPU, rh_index_bcast, reads sharded, PAF in
    read order = the reference's golden.  One context per visible GPU (N = 1 on the test box; RCCL when N > 1), and three contexts
    sharing the devices round-robin so that the sharded flow and the replication run on a one-GPU box too."""
    import golden
    from rawhash_amd import strip_mt
    exe = build_consumer(tmp_path, product_lib, SRC_MULTI)
    case = [c for c in golden.cases() if c["name"] == "small_sensitive"][0]
    w = golden.build_case(case, tmp_path / "wl", product_lib)
    rhr = os.path.join(str(tmp_path), "reads.rhr")
    w.reads.write(rhr, w.wl.cfg.digitisation, w.wl.cfg.range, w.wl.cfg.offset)
    for n_ctx in ("0", "3"):
        p = subprocess.run([exe, "sensitive", w.ind, rhr, n_ctx], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert [strip_mt(x) for x in p.stdout.splitlines()] == golden.expected_paf(case)
        assert "context(s)" in p.stderr

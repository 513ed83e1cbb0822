"""Regenerate tests/golden/*.paf with the reference (run in the build container, where /root/reference exists):

    python tests/golden/make_golden.py

For every case: synthetic reference + reads from seeds (rawhash_amd host code) -> `ref_harness index` (the reference's
own index builder) -> `ref_harness map -t 1` (the reference's own mapper + PAF printer) -> PAF minus the mt:f: tag.
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    {"name": "small_sensitive", "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=160, junk=150, noise=150_000, read_seed=11)},
    {"name": "small_fast", "workload": dict(preset="fast", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=160, junk=150, noise=150_000, read_seed=12)},
    {"name": "small_faster_minimizers", "workload": dict(preset="faster", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=160, junk=150, noise=150_000, read_seed=13)},
    {"name": "small_viral_dense", "workload": dict(preset="viral", chrom_len=150_000, n_chrom=1, n_samples=20_000, n_reads=120, junk=100, noise=100_000, read_seed=14)},
    {"name": "clean_ecoli_like", "workload": dict(preset="sensitive", chrom_len=1_000_000, n_chrom=1, n_samples=40_000, n_reads=200, junk=102, noise=0, read_seed=15)},
    # BASELINE.json configs[0]: E. coli-sized genome, 1 k reads, the reference's own CPU path
    {"name": "config1_ecoli_4p6M_1k", "workload": dict(preset="sensitive", chrom_len=4_600_000, n_chrom=1, n_samples=40_000, n_reads=1000, junk=102, noise=0, read_seed=3)},
    # BASELINE.json configs[2] at its index size (144 Mbp in 6 targets): a sample of its read set; the index is too large for the
    # CPU suite, so only the -m gpu tests (index built on the device) check this one
    # whole-read rounds on a sequence index: `--disable-adaptive` (RI_M_NO_ADAPTIVE, main.cpp:369)
    {"name": "small_sensitive_whole_reads", "no_adaptive": True, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=16)},
    # the int16 samples taken in the way the FAST5 reader does (rsig.c:346-374: float arithmetic, kept values truncated to int16):
    # BASELINE config 1 is stated on FAST5 reads.  RH_FAST5_INGEST=1 makes the harness restate those lines (the reader itself needs HDF5)
    {"name": "small_sensitive_fast5_ingest", "env": {"RH_FAST5_INGEST": "1"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=160, junk=150, noise=150_000, read_seed=17, fast5=True)},
    # f4: RMQ chaining (--rmq, lchain.c:606) with the inner tree, without it (--rmq-inner-dist 0), with a tree small enough to hit
    # --rmq-size-cap, and the RMQ re-chaining of DP chains with a longer bandwidth (--bw-long, rmap.cpp:336)
    {"name": "small_sensitive_rmq", "env": {"RH_RMQ": "1"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=120, junk=150, noise=150_000, read_seed=31, mapopt={"flag": 2})},
    {"name": "small_fast_rmq_no_inner", "env": {"RH_RMQ": "1", "RH_RMQ_INNER_DIST": "0"}, "workload": dict(preset="fast", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=32, mapopt={"flag": 2, "rmq_inner_dist": 0})},
    {"name": "small_sensitive_rmq_cap", "env": {"RH_RMQ": "1", "RH_RMQ_SIZE_CAP": "40", "RH_RMQ_INNER_DIST": "300"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=33, mapopt={"flag": 2, "rmq_size_cap": 40, "rmq_inner_dist": 300})},
    {"name": "small_sensitive_bw_long", "env": {"RH_BW_LONG": "2000"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=120, junk=150, noise=150_000, read_seed=34, mapopt={"bw_long": 2000})},
    # f4: DTW re-scoring of chains (--dtw-evaluate-chains on a --store-sig index, rmap.cpp:128-208, dtw.cpp): the defaults (alignment between
    # consecutive anchors, slanted band), the whole chain at once (global border), full matrices instead of bands
    {"name": "small_sensitive_dtw", "env": {"RH_STORE_SIG": "1", "RH_DTW": "1"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=120, junk=150, noise=150_000, read_seed=41, idxflag=0x10, mapopt={"flag": 0x40})},
    {"name": "small_fast_dtw_global", "env": {"RH_STORE_SIG": "1", "RH_DTW": "1", "RH_DTW_BORDER": "0"}, "workload": dict(preset="fast", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=42, idxflag=0x10, mapopt={"flag": 0x40, "dtw_border_constraint": 0})},
    {"name": "small_sensitive_dtw_full", "env": {"RH_STORE_SIG": "1", "RH_DTW": "1", "RH_DTW_FILL": "0", "RH_DTW_MIN_SCORE": "5"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=43, idxflag=0x10, mapopt={"flag": 0x40, "dtw_fill_method": 0, "dtw_min_score": 5.0})},
    {"name": "config3_dmel_144M_384", "gpu_only": True, "workload": dict(preset="sensitive", chrom_len=24_000_000, n_chrom=6, n_samples=40_000, n_reads=384, junk=102, noise=0, read_seed=3)},
    # f4 at the configurations' index sizes: --rmq and --bw-long on the 144 Mbp / 6-target index of config 3 (30 k anchors per chunk), DTW re-scoring
    # on the 4.6 Mbp index of config 1 (--store-sig)
    {"name": "config3_dmel_144M_rmq", "gpu_only": True, "env": {"RH_RMQ": "1"}, "workload": dict(preset="sensitive", chrom_len=24_000_000, n_chrom=6, n_samples=40_000, n_reads=128, junk=102, noise=0, read_seed=5, mapopt={"flag": 2})},
    {"name": "config3_dmel_144M_bw_long", "gpu_only": True, "env": {"RH_BW_LONG": "2000"}, "workload": dict(preset="sensitive", chrom_len=24_000_000, n_chrom=6, n_samples=40_000, n_reads=128, junk=102, noise=0, read_seed=6, mapopt={"bw_long": 2000})},
    # option values beyond the device path's former limits: chunks of 8000 samples (the rows-in-HBM event kernels, a chunk at a time), 40 chunks
    # of 1000 samples (more than 32 chunk boundaries per read), chains of a single anchor (--min-anchors 1: 128 B of region scratch per anchor)
    {"name": "small_sensitive_chunk8000", "env": {"RH_CHUNK_SIZE": "8000"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=40_000, n_reads=96, junk=150, noise=150_000, read_seed=61, mapopt={"chunk_size": 8000})},
    {"name": "small_sensitive_40chunks", "env": {"RH_CHUNK_SIZE": "1000", "RH_MAX_CHUNKS": "40"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=40_000, n_reads=96, junk=150, noise=150_000, read_seed=62, mapopt={"chunk_size": 1000, "max_num_chunk": 40})},
    {"name": "small_sensitive_min_anchors1", "env": {"RH_MIN_ANCHORS": "1"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=96, junk=150, noise=150_000, read_seed=63, mapopt={"min_num_anchors": 1})},
    # `--r10` (main.cpp:396-406): k = 9 (a pore model of 4^9 levels, span e + 8), segmentation windows 3 / 6, thresholds 6.5 / 4.0, peak height 0.2, gap scale 1.2
    {"name": "small_r10", "env": {"RH_R10": "1"}, "workload": dict(preset="sensitive", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=160, junk=150, noise=150_000, read_seed=71, r10=True)},
    {"name": "small_r10_fast", "env": {"RH_R10": "1"}, "workload": dict(preset="fast", chrom_len=300_000, n_chrom=2, n_samples=20_000, n_reads=120, junk=150, noise=150_000, read_seed=72, r10=True)},
    {"name": "config1_ecoli_4p6M_dtw", "env": {"RH_STORE_SIG": "1", "RH_DTW": "1"}, "workload": dict(preset="sensitive", chrom_len=4_600_000, n_chrom=1, n_samples=40_000, n_reads=300, junk=102, noise=0, read_seed=7, idxflag=0x10, mapopt={"flag": 0x40})},
]


AVA_CASES = [
    # BASELINE.json configs[4] in the small: reads of 3000 bases (27 k samples) drawn from a short genome so that they overlap
    {"name": "ava_small", "workload": dict(preset="ava", chrom_len=20_000, n_samples=27_000, n_reads=60, junk=50, noise=150_000, read_seed=21)},
    {"name": "ava_ragged", "workload": dict(preset="ava", chrom_len=15_000, n_samples=27_000, n_reads=64, junk=50, noise=150_000, read_seed=24, ragged=True)},
    {"name": "ava_viral_small", "workload": dict(preset="ava-viral", chrom_len=12_000, n_samples=27_000, n_reads=40, junk=50, noise=150_000, read_seed=26)},
    {"name": "ava_large_small", "workload": dict(preset="ava-large", chrom_len=20_000, n_samples=27_000, n_reads=48, junk=50, noise=150_000, read_seed=27)},
    {"name": "ava_sensitive_small", "workload": dict(preset="ava-sensitive", chrom_len=20_000, n_samples=27_000, n_reads=48, junk=50, noise=150_000, read_seed=22)},
]


# Repeat-rich references (tests/repeat_workload.py): tandem repeats, segmental duplications with 1 % divergence, assembly gaps.
# The small one runs in the CPU suite (oracle) and on the GPU; the 52 Mbp one puts > 8192 anchors into a chunk, so the multi-workgroup
# exact sorter works on real tie-ridden keys (GPU suite only: the index is built on the device)
REPEAT_CASES = [
    {"name": "repeat_small_2M", "workload": dict(preset="sensitive", n_chrom=2, chrom_len=1_000_000, n_reads=120, genome_seed=41, read_seed=43)},
    {"name": "repeat_fast_2M", "workload": dict(preset="fast", n_chrom=2, chrom_len=1_000_000, n_reads=120, genome_seed=45, read_seed=47)},
    {"name": "repeat_faster_2M", "workload": dict(preset="faster", n_chrom=2, chrom_len=1_000_000, n_reads=120, genome_seed=55, read_seed=57)},   # minimisers (w = 3) over tandem repeats: equal minima
    {"name": "repeat_rich_52M", "gpu_only": True, "workload": dict(preset="sensitive", n_chrom=4, chrom_len=13_000_000, n_reads=200, genome_seed=51, read_seed=53)},
    # f4 on repeat-rich input, chunks of more than 8192 anchors: RMQ chaining with ties everywhere
    {"name": "repeat_rich_52M_rmq", "gpu_only": True, "env": {"RH_RMQ": "1"}, "workload": dict(preset="sensitive", n_chrom=4, chrom_len=13_000_000, n_reads=96, genome_seed=51, read_seed=59, mapopt={"flag": 2})},
]


def main():
    import oracle_lib as O
    from conftest import Workload
    from rawhash_amd import _capi
    lib = _capi.lib()
    assert O.have_reference(), "build oracle/_ref first (make -C oracle ref)"
    only = set(sys.argv[1:])                # optional: regenerate just the named cases (the case lists are always rewritten)
    for case in CASES:
        if only and case["name"] not in only:
            continue
        with tempfile.TemporaryDirectory() as d:
            w = Workload(d, lib, **case["workload"], build_index=not case.get("gpu_only"), no_adaptive=bool(case.get("no_adaptive")))
            cfg = w.wl.cfg
            rhr = os.path.join(d, "reads.rhr")
            w.reads.write(rhr, cfg.digitisation, cfg.range, cfg.offset)
            preset = case["workload"]["preset"]
            ref_ind = os.path.join(d, "refbuilt.ind")
            env = dict(os.environ, **case.get("env", {}))
            subprocess.run([O.REF_HARNESS, "index", preset, w.fasta, w.model, ref_ind, "4"], check=True, stderr=subprocess.DEVNULL, env=env)
            if case.get("no_adaptive"):
                env["RH_NO_ADAPTIVE"] = "1"
            out = subprocess.run([O.REF_HARNESS, "map", preset, ref_ind, rhr, "1"], check=True, capture_output=True, text=True, env=env).stdout
            lines = [O.strip_mt(l) for l in out.splitlines()]
            assert len(lines) == len(w.reads)
            with open(os.path.join(HERE, case["name"] + ".paf"), "w") as f:
                f.write("\n".join(lines) + "\n")
            print(case["name"], len(lines), "lines,", sum(1 for l in lines if l.split("\t")[4] != "*"), "mapped")
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(CASES, f, indent=1)
    from repeat_workload import RepeatWorkload
    for case in REPEAT_CASES:
        if only and case["name"] not in only:
            continue
        with tempfile.TemporaryDirectory() as d:
            w = RepeatWorkload(d, lib, **case["workload"])
            ref_ind = os.path.join(d, "refbuilt.ind")
            env = dict(os.environ, **case.get("env", {}))
            subprocess.run([O.REF_HARNESS, "index", w.preset, w.fasta, w.model, ref_ind, "8"], check=True, stderr=subprocess.DEVNULL, env=env)
            out = subprocess.run([O.REF_HARNESS, "map", w.preset, ref_ind, w.rhr, "8"], check=True, capture_output=True, text=True, env=env).stdout
            lines = [O.strip_mt(l) for l in out.splitlines()]
            assert len(lines) == len(w.reads)
            with open(os.path.join(HERE, case["name"] + ".paf"), "w") as f:
                f.write("\n".join(lines) + "\n")
            print(case["name"], len(lines), "lines,", sum(1 for l in lines if l.split("\t")[4] != "*"), "mapped")
    with open(os.path.join(HERE, "repeat_cases.json"), "w") as f:
        json.dump(REPEAT_CASES, f, indent=1)
    # Rawsamble (all-vs-all overlapping): `ref_harness sigindex` builds the signal-target index from the reads with the
    # reference's own functions, `ref_harness map` overlaps the same reads against it
    import hashlib
    from conftest import AvaWorkload
    old_sha = {}
    if os.path.exists(os.path.join(HERE, "ava_cases.json")):
        with open(os.path.join(HERE, "ava_cases.json")) as f:
            old_sha = {c["name"]: c.get("ind_sha256") for c in json.load(f)}
    for case in AVA_CASES:
        if only and case["name"] not in only:
            case["ind_sha256"] = old_sha.get(case["name"])
            continue
        with tempfile.TemporaryDirectory() as d:
            w = AvaWorkload(d, lib, **case["workload"])
            ref_ind = os.path.join(d, "ref.ind")
            subprocess.run([O.REF_HARNESS, "sigindex", w.preset, w.rhr, w.model, ref_ind, "4"], check=True, stderr=subprocess.DEVNULL)
            out = subprocess.run([O.REF_HARNESS, "map", w.preset, ref_ind, w.rhr, "1"], check=True, capture_output=True, text=True).stdout
            lines = [O.strip_mt(l) for l in out.splitlines()]
            with open(os.path.join(HERE, case["name"] + ".paf"), "w") as f:
                f.write("\n".join(lines) + "\n")
            case["ind_sha256"] = hashlib.sha256(O.mask_ind(open(ref_ind, "rb").read())).hexdigest()
            print(case["name"], len(lines), "lines,", sum(1 for l in lines if l.split("\t")[4] != "*"), "overlaps,", len(w.reads), "reads")
    with open(os.path.join(HERE, "ava_cases.json"), "w") as f:
        json.dump(AVA_CASES, f, indent=1)


if __name__ == "__main__":
    main()

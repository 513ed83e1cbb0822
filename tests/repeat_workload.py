"""TEST INFRASTRUCTURE: a repeat-rich reference and reads drawn from it, for the parity cases an i.i.d.-uniform genome cannot reach
(mid_occ filtering, the tandem flag of rseed.c:105-154, rep_len, heavy key ties in the anchor sort on a real workload):
tandem repeats of 2-200 bp units, duplicated blocks of 10-100 kb (segmental duplications), runs of N.  Everything is drawn from
numpy's default_rng(seed) - the same image runs here and on the GPU box, so inputs regenerate bit for bit; only the PAF the
reference prints for them is committed (tests/golden/make_golden.py)."""
import os

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(seed=41, n_chrom=4, chrom_len=13_000_000, tandem_per_chrom=60, dups_per_chrom=6, n_runs_per_chrom=5):
    rng = np.random.default_rng(seed)
    chroms = [ACGT[rng.integers(0, 4, size=chrom_len, dtype=np.uint8)] for _ in range(n_chrom)]
    for c in chroms:                                   # tandem repeats
        for _ in range(tandem_per_chrom):
            unit = ACGT[rng.integers(0, 4, size=int(rng.integers(2, 201)), dtype=np.uint8)]
            span = int(rng.integers(1_000, 20_001))
            at = int(rng.integers(0, chrom_len - span))
            c[at:at + span] = np.tile(unit, span // len(unit) + 1)[:span]
    for c in chroms:                                   # segmental duplications (copies of blocks from anywhere)
        for _ in range(dups_per_chrom):
            L = int(rng.integers(10_000, 100_001))
            src = chroms[int(rng.integers(0, n_chrom))]
            a, b = int(rng.integers(0, chrom_len - L)), int(rng.integers(0, chrom_len - L))
            blk = src[a:a + L].copy()
            mut = rng.random(L) < 0.01                 # 1 % divergence between the copies
            blk[mut] = ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
            c[b:b + L] = blk
    for c in chroms:                                   # assembly gaps
        for _ in range(n_runs_per_chrom):
            L = int(rng.integers(1_000, 50_001))
            at = int(rng.integers(0, chrom_len - L))
            c[at:at + L] = ord("N")
    return chroms


def write_fasta(path, chroms):
    with open(path, "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">rep%d\n" % (i + 1))
            for o in range(0, len(c), 1 << 20):       # one MiB per line is fine for both parsers
                f.write(c[o:o + (1 << 20)].tobytes() + b"\n")


def load_model_levels(model_path, k=6):
    """Level (pA) of every k-mer in 2-bit order, from the pore model file written by rh_synth_write_model."""
    lev = np.zeros(4 ** k, dtype=np.float64)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    with open(model_path) as f:
        for line in f:
            t = line.split("\t")
            if line.startswith("kmer") or len(t) < 2:
                continue
            v = 0
            for ch in t[0]:
                v = v * 4 + code[ch]
            lev[v] = float(t[1])
    return lev


def simulate_reads(chroms, levels, seed=43, n_reads=200, n_samples=40_000, digitisation=8192.0, rng_pa=1402.882, offset=6.0, junk_every=11, k=6):
    """R9.4-like raw reads: start uniform (spans that touch an assembly gap are redrawn), strand 50/50, dwell per base
    max(1, round(Gamma(2, 4.45))), noise N(0, 1.5 pA); every `junk_every`-th read is random sequence (unmappable)."""
    from rawhash_amd.api import Reads
    rng = np.random.default_rng(seed)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    idx = np.zeros(256, dtype=np.int64)
    for i, a in enumerate(b"ACGT"):
        idx[a] = i
    span = n_samples // 4 + 16
    samples, names = [], []
    for r in range(n_reads):
        if junk_every and r % junk_every == junk_every - 1:
            bases, name = ACGT[rng.integers(0, 4, size=span, dtype=np.uint8)], f"r{r}_junk"
        else:
            while True:
                ci = int(rng.integers(0, len(chroms))); pos = int(rng.integers(0, len(chroms[ci]) - span)); st = int(rng.integers(0, 2))
                bases = chroms[ci][pos:pos + span]
                if not (bases == ord("N")).any():
                    break
            if st:
                bases = comp[bases[::-1]]
            name = f"r{r}_rep{ci + 1}_{pos}_{'-' if st else '+'}"
        b2 = idx[bases]
        km = np.zeros(span - k + 1, dtype=np.int64)
        for j in range(k):
            km = km * 4 + b2[j:span - k + 1 + j]
        dwell = np.maximum(1, np.rint(rng.gamma(2.0, 4.45, size=len(km)))).astype(np.int64)
        pa = np.repeat(levels[km], dwell)[:n_samples]
        if len(pa) < n_samples:
            pa = np.concatenate([pa, np.full(n_samples - len(pa), pa[-1])])
        pa = pa + rng.normal(0.0, 1.5, size=n_samples)
        raw = np.rint(pa * digitisation / rng_pa - offset).clip(-32768, 32767).astype(np.int16)
        samples.append(raw); names.append(name)
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(n_samples)
    return Reads(np.concatenate(samples), off, names, offset, np.float32(rng_pa / digitisation))


class RepeatWorkload:
    """Directory with the repeat-rich FASTA, the pore model and the simulated reads (+ .rhr file for the reference harness)."""

    def __init__(self, directory, lib, preset="sensitive", n_chrom=4, chrom_len=13_000_000, n_reads=200, genome_seed=41, read_seed=43, mapopt=None):
        from rawhash_amd.api import SynthWorkload, MapOptions
        self.dir, self.preset = str(directory), preset
        os.makedirs(self.dir, exist_ok=True)
        self.wl = SynthWorkload(chrom_len=1000, n_chrom=1, n_samples=40_000, lib=lib)      # (only for the pore model + calibration constants)
        _, self.model = self.wl.write_reference(self.dir)
        self.chroms = make_genome(genome_seed, n_chrom, chrom_len)
        self.fasta = os.path.join(self.dir, "repeats.fa")
        write_fasta(self.fasta, self.chroms)
        cfg = self.wl.cfg
        self.reads = simulate_reads(self.chroms, load_model_levels(self.model), read_seed, n_reads, 40_000, cfg.digitisation, cfg.range, cfg.offset)
        self.rhr = os.path.join(self.dir, "reads.rhr")
        self.reads.write(self.rhr, cfg.digitisation, cfg.range, cfg.offset)
        self.opts = MapOptions(preset, lib=lib)
        for k, v in dict(mapopt or {}).items():     # rh_mapopt_t fields on top of the preset ("flag" is OR-ed in): --rmq, --bw-long
            if k == "flag":
                self.opts.mo.flag |= v
            else:
                setattr(self.opts.mo, k, v)
        self.index = None

"""Development aid (not a test): device index build + mapping at a given genome size; prints timings and counters."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rawhash_amd import Context, Index, MapOptions, SynthWorkload

chrom, nch, preset, nreads = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
junk = int(sys.argv[5]) if len(sys.argv) > 5 else 102
wl = SynthWorkload(chrom_len=chrom, n_chrom=nch, n_samples=40000, junk_per_1024=junk)
opts = MapOptions(preset)
wd = "/tmp/scale_probe"; os.makedirs(wd, exist_ok=True)
model = os.path.join(wd, "model.txt")
wl._l.rh_synth_write_model(C.byref(wl.cfg), model.encode())
t = time.time(); seqs = [wl.genome(c, n_threads=64) for c in range(nch)]; print(f"genome {time.time() - t:.2f} s", flush=True)
ctx = Context(0)
t = time.time(); ix = Index.build_device_seqs(ctx, [f"chr{i + 1}" for i in range(nch)], seqs, model, opts, n_threads=64)
print(f"device index build {time.time() - t:.2f} s: keys {ix.n_keys} positions {ix.n_positions}", flush=True)
opts.update(ix); print("mid_occ", opts.mo.mid_occ, flush=True)
batch = wl.reads_device(ctx, model, 0, nreads)
for it in range(2):
    t = time.time(); recs = ctx.map_batch(opts, batch); dt = time.time() - t
    st = ctx.stats()
    print(f"map {nreads} reads {dt:.3f} s = {nreads / dt:.0f} reads/s; mapped {recs['mapped'].mean():.3f}; chunks/read {st['n_chunks'] / nreads:.2f}; anchors/chunk {st['n_anchors'] / max(st['n_chunks'], 1):.0f}; "
          f"seeds/chunk {st['n_seeds'] / max(st['n_chunks'], 1):.0f} hits/seed {st['n_hits'] / max(st['n_seeds'], 1):.1f} chained/chunk {st['n_chained'] / max(st['n_chunks'], 1):.0f}", flush=True)
    print({k: round(v[0], 1) for k, v in st["stages"].items() if v[1]}, flush=True)

"""Development aid: all-vs-all at scale on the GPU (python tools/ava_probe.py <genome_bp> <reads> [cpu_sample])."""
import os, sys, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rawhash_amd.api import SynthWorkload, MapOptions, Index, Context, paf_lines, strip_mt
from conftest import AvaWorkload
from rawhash_amd import _capi
lib = _capi.lib()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t = time.time()
w = AvaWorkload(d, lib, preset="ava", chrom_len=int(sys.argv[1]), n_samples=27_000, n_reads=int(sys.argv[2]), junk=50, noise=150_000, read_seed=23)
print("reads generated", round(time.time() - t, 2), "s", flush=True)
c = Context(0, lib=lib)
t = time.time(); ix = Index.build_signals_device(c, w.reads, w.model, w.opts); print("index", round(time.time() - t, 2), "s keys", ix.n_keys, "pos", ix.n_positions, flush=True)
w.opts.update(ix); print("mid_occ", w.opts.mo.mid_occ)
for it in range(2):
    t = time.time(); recs, off = c.map_batch_multi(w.opts, w.reads, ix, max_records=600 * len(w.reads)); dt = time.time() - t
    print("map", round(dt, 2), "s", round(len(w.reads) / dt, 1), "reads/s", len(recs), "records", flush=True)
st = c.stats(); print({k: round(v[0], 1) for k, v in st["stages"].items() if v[1]}, {k: v for k, v in st.items() if k != "stages"})

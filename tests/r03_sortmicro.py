"""Development aid: k_sort_block on free-standing segments (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rawhash_amd import api
n_seg, seg = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
a = np.zeros(n_seg * seg, dtype=api.MM128)
a["x"] = rng.integers(0, 1 << int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 16, n_seg * seg, dtype=np.uint64) | (np.uint64(3) << np.uint64(32))   # argv[3]: key bits (16: heavy ties; 27: a position in a chromosome, hardly any)
a["y"] = rng.integers(0, 1 << 40, n_seg * seg, dtype=np.uint64)
off = (np.arange(n_seg + 1, dtype=np.uint64) * np.uint64(seg))
ctx = api.Context(0)
for rep in range(3):
    t = time.time(); out = ctx.sort128x(a, off); print("rep", rep, time.time() - t, flush=True)
x = out["x"].reshape(n_seg, seg)
print("sorted", bool((np.diff(x.astype(np.int64), axis=1) >= 0).all()))

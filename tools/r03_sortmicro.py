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
if os.environ.get("RH_KPROF_PRINT"):   # library built with RH_HIPCC_EXTRA=-DRH_KPROF: shader-clock cycles per phase of k_sort_block, summed over workgroups
    import ctypes
    from rawhash_amd import _capi
    lib = _capi.lib()
    out = (ctypes.c_ulonglong * 32)()
    lib.rh_debug_kprof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.rh_debug_kprof(out, 1)
    names = {1: "diff / tie reduce", 2: "histogram + scan", 3: "fast scatter", 4: "apply gather", 5: "two buckets", 6: "cycle walk", 7: "children", 8: "(levels, incl. 1-7)", 9: "ranges <= 64", 10: "load keys", 11: "tie scan", 12: "write-out"}
    tot = sum(out[i] for i in names if i != 8)
    for i, nm in names.items():
        print(f"{nm:22s} {out[i] / 1e6 / 3 / n_seg:10.4f} Mcyc per workgroup-equivalent x1e0 {100 * out[i] / max(tot, 1):5.1f}%")

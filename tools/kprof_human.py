"""Development aid: phases of k_sort_block on the human-scale bench step (library built with RH_HIPCC_EXTRA=-DRH_KPROF)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rawhash_amd import _capi
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-h2d"] + sys.argv[1:]
os.environ.setdefault("RH_SUB_BATCHES", "1")
bench.main()
lib = _capi.lib()
f = lib.rh_debug_kprof
out = (C.c_ulonglong * 32)()
f(out, 0)
names = {1: "or/and", 2: "histogram", 3: "two buckets", 4: "scatter / apply", 5: "cycle walk (full)", 6: "walk, early stop", 7: "children", 8: "small ranges", 9: "insertion", 10: "load keys", 11: "tie scan", 12: "write out", 13: "fast: load + min/max", 14: "fast: rank atomics", 15: "fast: scan", 16: "fast: scatter", 17: "fast: bucket networks", 18: "fast: write out", 20: "(count) fast: done", 21: "(count) fast: equal keys", 22: "(count) fast: not applicable", 23: "(count) records offered"}
tot = sum(out[:20])
for i in range(32):
    if out[i]:
        print(f"slot {i:2d} {names.get(i, ''):30s} {out[i]:14d}" if i >= 20 else f"slot {i:2d} {names.get(i, ''):30s} {out[i] / 1e9:9.2f} Gcyc {100.0 * out[i] / tot:5.1f} %")

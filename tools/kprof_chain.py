"""Development aid: what k_chain_wave's tiles are made of on a bench step (library built with RH_HIPCC_EXTRA=-DRH_KPROF).  Usage: python tools/kprof_chain.py [bench args]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rawhash_amd import _capi
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-h2d"] + sys.argv[1:]
os.environ.setdefault("RH_SUB_BATCHES", "1")
bench.main()
out = (C.c_ulonglong * 16)()
_capi.lib().rh_debug_kprof_chain(out, 0)
names = ["tiles", "anchors", "singletons", "anchors in small clusters", "anchors in large clusters", "tiles entering the small path", "tiles entering the large path",
         "pair-score rounds (small path)", "DP steps (small path)", "anchors in clusters of two", "clocks: large path", "clocks: small path", "clocks: tile set-up", "clocks: staging + pair scores", "clocks: skip-free steps", "clocks: max_ii catch-up + generic steps"]
for i, n in enumerate(names):
    print(f"slot {i:2d} {n:34s} {out[i]:16d}")

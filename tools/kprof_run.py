import ctypes, json, subprocess, sys, os
sys.path.insert(0, "/root/repo")
os.environ["RH_SUB_BATCHES"] = "1"
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"]
from rawhash_amd import _capi
lib = _capi.lib()
bench.main()
out = (ctypes.c_ulonglong * 32)()
lib.rh_debug_kprof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
print("kprof rc", lib.rh_debug_kprof(out, 1))
names = {1: "diff/tie reduce", 2: "hist+scan", 3: "fast scatter", 4: "apply gather", 5: "two-bucket", 6: "cycle walk", 7: "children", 8: "(levels total incl 1-7)", 9: "small ranges", 10: "load keys", 11: "tie check", 12: "store"}
tot = sum(out[i] for i in names if i != 8)
for i, nm in names.items(): print(f"{nm:24s} {out[i]/1e6:12.1f} Mcyc {100*out[i]/max(tot,1):5.1f}%")


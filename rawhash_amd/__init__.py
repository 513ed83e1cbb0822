"""rawhash_amd -- MI355X-native raw-signal mapping path (drop-in for RawHash2's map_worker_for pipeline).

Thin Python host layer over the C ABI of include/rawhash_amd.h (ctypes).  All compute happens in
rawhash_amd/librawhash_amd.so (hand-written HIP for gfx950); there is no Python or CPU fallback.
"""
from .api import Context, Index, MapOptions, Reads, SynthWorkload, RhError, paf_lines, strip_mt  # noqa: F401

__all__ = ["Context", "Index", "MapOptions", "Reads", "SynthWorkload", "RhError", "paf_lines", "strip_mt"]

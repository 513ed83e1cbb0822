"""In-tree build of the product shared library rawhash_amd/librawhash_amd.so with hipcc for gfx950.

    python -m rawhash_amd.build [--force]

Every source is compiled as HIP (-x hip) with -ffp-contract=off: the reference's fp32 results depend on unfused
multiply-adds and host/device code share rh_core.h.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librawhash_amd.so")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["rh_common.cpp", "rh_options.cpp", "rh_reads.cpp", "rh_synth.cpp", "rh_index.cpp", "rh_paf.cpp", "rh_api.cpp", "rh_kernels.hip", "rh_sort.hip", "rh_bigsort.hip", "rh_index_device.hip", "rh_chain.hip", "rh_post.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-x", "hip", "-I", os.path.join(HERE, "..", "include")]


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "rawhash_amd.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            cmd = [hipcc] + FLAGS + os.environ.get("RH_HIPCC_EXTRA", "").split() + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    if procs or force or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

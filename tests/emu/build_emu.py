"""TEST INFRASTRUCTURE ONLY: compile the unchanged product sources against the SIMT emulator header (tests/emu/rh_gpu.h)
into tests/emu/_build/librawhash_emu.so with g++.  Used by the `not gpu` tests to exercise kernel logic on the CPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "rawhash_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
OUT = os.path.join(BUILD, "librawhash_emu.so")

# small LDS caps so that the oversized-read fallbacks (the *_big kernels) are exercised on small test inputs
SMALL_CAPS = ("-DRG_CAP=8", "-DRGW_CAP=48", "-DRGW_CAP0=16", "-DRGR_PRIM_CAP=5", "-DRGB_PCAP=64", "-DRGR_DIRECT=20", "-DCG_CAP=8", "-DPF_TILES=40", "-DRH_SORT_CAP0=128", "-DRH_SORT_CAP1=384", "-DRH_SORT_CAP2=512", "-DRH_SORT_CAP3=768", "-DRH_SORT_CAP4=1024", "-DRH_SORT32_CAPH=256", "-DRH_SORT32_CAP1=384", "-DRH_SORT32_CAP2=640", "-DRH_SORT32_CAP3=1024", "-DIX_L=64", "-DIX_WU=3", "-DCH_TILE=128", "-DBS_TILE_IT=1", "-DBS_WIN_BYTES=1024", "-DBS_LANES_MIN_RANGES=100000", "-DBS_MULTI_MIN_RANGES=1", "-DBS_MW_PERIOD=5", "-DBS_MW_G16_FROM=3", "-DBS_MW_G32_FROM=10",
              "-DPW_MIN_HOLES=256", "-DPW_MIN_C0=16", "-DPW_MAX_CYCLE=4096", "-DPW_WIN_CAP=400", "-DPW_BLK_HOLES=200", "-DPW_RING_BYTES=4096", "-DPW_TEST_FEW_SLOTS", "-DRQ_RING_S=4", "-DRQ_RING=8", "-DRQ_RING_BIG=32", "-DBT_LDS_ANCHORS=512", "-DBT_LDS_CLAIMS=64")


def build(force=False, defines=(), tag=""):
    import fcntl
    src_dir = os.path.join(BUILD, "src")
    os.makedirs(src_dir, exist_ok=True)
    names = [f for f in sorted(os.listdir(CSRC)) if f != "rh_gpu.h" and f.endswith((".h", ".cpp", ".hip"))]
    newest = max([os.path.getmtime(os.path.join(CSRC, f)) for f in names] +
                 [os.path.getmtime(os.path.join(HERE, f)) for f in ("rh_gpu.h", "emu_runtime.cpp")] +
                 [os.path.getmtime(os.path.join(ROOT, "include", "rawhash_amd.h"))])
    out = OUT.replace(".so", tag + ".so")
    with open(os.path.join(BUILD, ".lock"), "w") as lock:       # several test processes (2-rank gloo test) may get here together
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            return out
        srcs = []
        for f in names:
            dst = os.path.join(src_dir, f)
            if os.path.lexists(dst):
                os.remove(dst)
            os.symlink(os.path.join(CSRC, f), dst)       # same files, but rh_gpu.h now resolves to the emulator's
            if f.endswith((".cpp", ".hip")):
                srcs.append(dst)
        for f in ("rh_gpu.h", "emu_runtime.cpp"):
            shutil.copy(os.path.join(HERE, f), os.path.join(src_dir, f))
        srcs.append(os.path.join(src_dir, "emu_runtime.cpp"))
        # -fno-gnu-unique: a kernel template's `__shared__` arrays are function-local statics here, which g++ would emit as STB_GNU_UNIQUE symbols - ONE
        # instance per process even across dlopen(RTLD_LOCAL), so the production-size build and the small-caps build (same names, different array sizes)
        # would share whichever was loaded first (round 5: test_exact_sort_multi_workgroup crashed in k_bs_scatter when run on its own)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-gnu-unique", "-pthread", "-w",
               "-I", src_dir, "-I", os.path.join(ROOT, "include"), "-x", "c++"] + list(defines) + srcs + ["-o", out + ".tmp", "-lz"]
        subprocess.run(cmd, check=True)
        os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True))

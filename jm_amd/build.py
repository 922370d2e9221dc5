"""Build libjmhip.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m jm_amd.build          # -> jm_amd/libjmhip.so

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libjmhip.so")
SOURCES = ["ctx.hip", "interp.hip", "me_fullsearch.hip", "me_fast.hip", "me_subpel.hip", "tq.hip", "deblock.hip", "deblock_rows.hip"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def newest_source():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "jmhip.h")]
    return max(os.path.getmtime(f) for f in files)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest_source():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=default",
           "-Wall", "-Wno-unused-function", "-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

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
SOURCES = ["ctx.hip", "input.hip", "interp.hip", "me_fullsearch.hip", "me_fast.hip", "me_subpel.hip", "me_refine.hip", "tq.hip", "tq8.hip", "tq16.hip", "tq_chroma.hip", "mc.hip", "intra.hip", "distortion.hip", "deblock.hip", "deblock_rows.hip", "deblock_sparse.hip"]
# mbpipe.hip is compiled thirteen times: -DMBPIPE_PART=0..3, 5..11 one kernel instance each (7, 8: B slices; 9, 10: several references; 11: the six-wave form; 12: part 0's kernel with JMHIP_MB_PROF's time stamps, which part 0 itself is compiled without), 4 the host side (one unit takes six minutes, the parts side by side)
MBPIPE_PARTS = (8, 3, 7, 10, 6, 1, 9, 5, 2, 0, 12, 11, 4)                                 # the slowest first
# The macroblock pipeline's units are compiled without machine-level loop-invariant code motion: in a kernel whose one loop body is 240 KB of code every value hoisted to the
# top is a value spilled (k_mb_pipe: 304 -> 208 bytes of scratch per lane, 2 % faster; profiles/r04_kernel_resources.txt).  JMHIP_MBPIPE_FLAGS adds flags (measurement aid).
MBPIPE_FLAGS = ["-mllvm", "-disable-machine-licm"] + os.environ.get("JMHIP_MBPIPE_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def newest_source():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "jmhip.h")]
    return max(os.path.getmtime(f) for f in files)


def build(force=False, verbose=False):
    """One hipcc -c per translation unit (in parallel, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest_source():
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    import re

    def deps(path, seen=None):
        """the file and everything it #includes with quotes, recursively (deblock_sparse.hip includes deblock_rows.hip; mbpipe.hip its .inc parts)"""
        seen = set() if seen is None else seen
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return seen
        seen.add(path)
        with open(path) as f:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), re.M):
                deps(os.path.join(os.path.dirname(path), inc), seen)
        return seen
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=default", "-Wall", "-Wno-unused-function"]

    def compile_one(job):
        src, part = job
        obj = os.path.join(objdir, os.path.basename(src) + (".o" if part is None else ".part%d.o" % part))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps(src)):
            cmd = [hipcc()] + flags + ([] if part is None else ["-DMBPIPE_PART=%d" % part] + (["-DMBPIPE_PROF_ON=0"] if part == 0 else []) + MBPIPE_FLAGS) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    jobs = [(os.path.join(CSRC, "mbpipe.hip"), k) for k in MBPIPE_PARTS] + [(s, None) for s in srcs]      # the slowest first
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

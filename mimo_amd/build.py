"""Build libmimo_hip.so (gfx950) in-tree with hipcc.

The shared library is the product's only compute path; there is no CPU fallback.
`python -m mimo_amd.build` or `__graft_entry__.build()` runs this.  hipcc cross-compiles
for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
INCLUDE = os.path.join(os.path.dirname(ROOT), "include")
LIB_PATH = os.path.join(ROOT, "libmimo_hip.so")
SOURCES = ["gemm_conv.hip", "hconv.hip", "thinconv.hip", "gemm_stream.hip", "ff_fused.hip", "ff_tail4.hip", "attention.hip", "norm.hip", "elementwise.hip", "image.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
# ff_tail4.hip: one wave per SIMD — packed fp32 VALU (what the SLP vectoriser makes of adjacent scalar adds / fmas) is slow
# beside MFMAs (MI355X_MICROARCH.md), the GEGLU there is written in scalar operations and must stay scalar
EXTRA_FLAGS = {"ff_tail4.hip": ["-fno-slp-vectorize", "-Wno-inline-asm"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


TUNE_LIB_PATH = os.path.join(ROOT, "libmimo_hip_tune.so")


def build(force: bool = False, verbose: bool = False, tune: bool = False) -> str:
    """tune=False: the shipped library (no environment reads, no mutable state).  tune=True: the same sources with
    -DMIMO_TUNE -> libmimo_hip_tune.so, whose MIMO_* knobs tools/microbench.py flips for interleaved A/B timing
    (load it with MIMO_HIP_LIB=.../libmimo_hip_tune.so)."""
    objdir = os.path.join(ROOT, "_obj_tune" if tune else "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + (["-DMIMO_TUNE"] if tune else [])
    lib_path = TUNE_LIB_PATH if tune else LIB_PATH
    headers = [os.path.join(CSRC, h) for h in ("common.hip.h", "gemm_stream.hip.h", "thinconv.hip.h", "ff_fused.hip.h")] + [os.path.join(INCLUDE, "mimo_hip.h")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + flags + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tune="--tune" in sys.argv))

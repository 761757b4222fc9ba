"""Build libsfmi.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m shapeformer_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsfmi.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(
        glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(HERE, "..", "include"), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return s

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print(f"[sfmi build] compiled {os.path.basename(s)}", file=sys.stderr)
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[sfmi build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

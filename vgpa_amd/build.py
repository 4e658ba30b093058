"""Build ``lib/libvgpa_hip.so`` in-tree with hipcc for gfx950 (``python -m vgpa_amd.build``).

This module must not import the rest of the package (the package import needs
the library this builds)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = [os.path.join(CSRC, f) for f in ("common.hip", "scan.hip", "esde.hip", "esde_l96_wave.hip", "assemble.hip", "ut.hip",
                                       "scg.hip", "path.hip")]
# per-file compiler options: the Lorenz-96 D = 40 energy kernel is issue-bound straight-line code and gains 1.5-2 % from
# the scheduler strategy that maximises instruction-level parallelism (it costs the memory-bound kernels registers)
EXTRA = {"esde_l96_wave.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
OUT = os.path.join(HERE, "lib", "libvgpa_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
           [os.path.join(ROOT, "include", "vgpa.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date():
    return not _newer(OUT, SRC + _headers())


def build(force=False, verbose=True):
    """Compile each translation unit that changed (in parallel) and link."""
    if up_to_date() and not force:
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    hdr = _headers()
    jobs = []
    for src in SRC:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _newer(obj, [src] + hdr):
            jobs.append([hipcc()] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(5, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in SRC]
    run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

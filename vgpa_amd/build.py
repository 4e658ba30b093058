"""Build ``lib/libvgpa_hip.so`` in-tree with hipcc for gfx950 (``python -m vgpa_amd.build``).

This module must not import the rest of the package (the package import needs
the library this builds)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("common.hip", "scan.hip", "esde.hip", "assemble.hip")]
HDR = [os.path.join(HERE, "csrc", "common.h"), os.path.join(ROOT, "include", "vgpa.h")]
OUT = os.path.join(HERE, "lib", "libvgpa_hip.so")


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in SRC + HDR)


def build(force=False, verbose=True):
    if up_to_date() and not force:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

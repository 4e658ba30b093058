"""Build another copy of the library with extra compiler flags, for A/B measurements:
    python tools/build_variant.py NAME [-DFLAG ...]   ->  vgpa_amd/lib/libvgpa_hip.NAME.so
Select it at run time with VGPA_HIP_LIBRARY=<path>."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vgpa_amd"))
import build as b   # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(b.HERE, "lib", "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(b.HERE, "lib", f"libvgpa_hip.{name}.so")
    only = os.environ.get("VGPA_VARIANT_UNITS")          # e.g. "scan,esde": other units are taken from the main build
    objs, jobs = [], []
    for src in b.SRC:
        unit = os.path.basename(src)[:-4]
        if only and unit not in only.split(","):
            objs.append(os.path.join(b.OBJ, unit + ".o"))
            continue
        obj = os.path.join(objdir, unit + ".o")
        objs.append(obj)
        per_file = [] if os.environ.get("VGPA_VARIANT_NO_EXTRA") else b.EXTRA.get(os.path.basename(src), [])
        jobs.append([b.hipcc()] + b.FLAGS + per_file + extra + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(lambda c: subprocess.run(c, check=True), jobs))
    subprocess.run([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print(out)


if __name__ == "__main__":
    main()

"""Instruction counts per phase and class of the Lorenz-96 wavefront-per-step kernel, from an ISA listing made with
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DVGPA_PHASE_MARKS -S --cuda-device-only -o esde.s vgpa_amd/csrc/esde.hip
usage: isa_phase_counts.py esde.s [kernel-name-substring]"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "esde_l96_ut_wave40_kernelILb0"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + want + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if "_f64" in op or "_b64" in op:
            return "valu64"
        if "dpp" in op or "permlane" in op or "readlane" in op or "readfirstlane" in op:
            return "xlane32"
        return "valu32"
    return "other"


phase = "start"
counts = collections.OrderedDict()
for line in lines[start + 1:end]:
    t = line.strip()
    if not t or t.startswith("."):
        continue
    if t.startswith(";"):
        mm = re.match(r"; WVPHASE (\d+)", t)
        if mm:
            phase = mm.group(1)
        continue
    if t.endswith(":"):
        continue
    op = t.split()[0]
    counts.setdefault(phase, collections.Counter())[cls(op)] += 1
tot = collections.Counter()
keys = ["mfma", "valu64", "valu32", "xlane32", "lds", "vmem", "salu", "other"]
print("phase " + " ".join(f"{k:>8s}" for k in keys) + "    total")
for ph, c in counts.items():
    print(f"{ph:>5s} " + " ".join(f"{c.get(k, 0):8d}" for k in keys) + f" {sum(c.values()):8d}")
    tot += c
print("  all " + " ".join(f"{tot.get(k, 0):8d}" for k in keys) + f" {sum(tot.values()):8d}")

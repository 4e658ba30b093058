"""Instruction classes and s_waitcnt values of the innermost loops of one kernel in an ISA listing.
usage: isa_loops.py file.s kernel-name-regex"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^\S*" + sys.argv[2] + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
for a, b in inner:
    c, w = collections.Counter(), collections.Counter()
    for s in body[a:b]:
        t = s.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op == "s_waitcnt":
            w[" ".join(t[1:])] += 1
        k = "mfma" if op.startswith("v_mfma") else "f64" if "_f64" in op else "vmem" if op.startswith(("global", "flat", "scratch", "buffer")) else \
            "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "valu32"
        c[k] += 1
    print(f"loop lines {a}-{b}: {dict(c)}  waitcnt {dict(w)}")

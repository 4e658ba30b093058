#!/usr/bin/env python3
"""Trim a rocprofv3 ``*_kernel_stats.csv`` into a short, committable summary.

usage: summarize_prof.py <dir-with-rocprof-output> <out.csv> [header line ...]
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    base = m.group(1) if m else name[:80]
    return ("[torch] " + base[:70]) if base.startswith(("at::", "__amd")) else base


def main():
    src, dst, notes = sys.argv[1], sys.argv[2], sys.argv[3:]
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        sys.exit("no kernel_stats.csv under " + src)
    rows = list(csv.DictReader(open(files[0])))
    with open(dst, "w") as f:
        for n in notes:
            f.write("# " + n + "\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    print(open(dst).read())


if __name__ == "__main__":
    main()

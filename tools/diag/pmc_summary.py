"""Sum rocprofv3 --pmc counter rows per kernel: pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in files:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if sub in k:
            n = len(disp[k])
            print(k, "dispatches", n, {c: x / n for c, x in v.items()})

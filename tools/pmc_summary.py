"""Dev tool: per-kernel means of the counters in rocprofv3 --pmc output directories.  usage: pmc_summary.py <dir> [<dir> ...] [--filter substr]"""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    dirs = [d for d in dirs if d != flt]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and not re.search(flt, k):
                continue
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    n = max(v[1] for v in cs.values())
    print(f"{k}  (dispatches {n})")
    for c, (s, m) in sorted(cs.items()):
        print(f"    {c:32s} {s / m:16.1f}")

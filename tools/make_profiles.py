"""Turn the rocprofv3 output directories of a bench.py run into the committed summaries under profiles/.

  gpurun_out/prof_stats  <- rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- python3 bench.py ...
  gpurun_out/prof_fetch  <- rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -- python3 bench.py ...   (own pass)
  gpurun_out/prof_write  <- rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -- python3 bench.py ...   (own pass)

usage: python tools/make_profiles.py <tag>        (writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_traffic.json)
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 under-reports wide coalesced reads by 2x,
MI355X_MICROARCH.md HBM section).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def find(d, pat):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", pat), recursive=True)      # older runs' files stay in gpurun_out/
    return max(f, key=os.path.getmtime) if f else None


def main():
    tag = sys.argv[1]
    st = find("prof_stats", "*kernel_stats.csv")
    if st:
        rows = list(csv.DictReader(open(st)))
        out = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.2f}',
                            f'{float(r["MinNs"]) / 1e3:.2f}', f'{float(r["MaxNs"]) / 1e3:.2f}', r["Percentage"]])
        print("wrote", out)
    acc = defaultdict(lambda: {"fetch_kb_raw": 0.0, "write_kb": 0.0, "n_f": 0, "n_w": 0})
    for d, key, cnt in (("prof_fetch", "fetch_kb_raw", "n_f"), ("prof_write", "write_kb", "n_w")):
        f = find(d, "*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][key] += float(r["Counter_Value"]); acc[k][cnt] += 1
    if acc:
        kern = {}
        for k, v in acc.items():
            nf, nw = max(v["n_f"], 1), max(v["n_w"], 1)
            fk, wk = v["fetch_kb_raw"] / nf, v["write_kb"] / nw
            kern[k] = {"fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1), "launches_sampled": max(v["n_f"], v["n_w"]),
                       "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        out = os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json")
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, over `python3 bench.py --steps 2 --warmup 1 "
                           "--no-cpu-baseline --no-roofline`; per-launch averages. Counter unit = KB; FETCH_SIZE doubled (gfx950 reports "
                           "half of wide coalesced 16-B/lane reads, MI355X_MICROARCH.md HBM section).", "kernels": kern},
                  open(out, "w"), indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()

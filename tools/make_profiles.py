"""Turn the rocprofv3 output directories of tools/run_profiles.sh into the committed summaries under profiles/.

  gpurun_out/<tag>/<workload>_stats  <- rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload <workload> ...
  gpurun_out/<tag>/<workload>_fetch  <- rocprofv3 --pmc FETCH_SIZE ...   (own pass)
  gpurun_out/<tag>/<workload>_write  <- rocprofv3 --pmc WRITE_SIZE ...   (own pass)
  gpurun_out/<tag>/<workload>_bench.json, train31_rccl_world1.json  <- the un-profiled bench lines

usage: python tools/make_profiles.py <tag>   (writes profiles/<tag>_<workload>_kernel_stats.csv, _hbm_traffic.json, _bench.json)
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 under-reports wide coalesced reads by 2x,
MI355X_MICROARCH.md HBM section).
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """kernel name without return type, the '(anonymous namespace)::' qualifier and the trailing ARGUMENT list (template
    arguments stay, so variants of one kernel keep their own rows)"""
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):          # the argument list = the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return name.strip()


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    # (the PMC / stats directories exist before the bench lines do: run_profiles.sh calls this once in between, so that the bench lines
    # can quote the traffic of their own session)
    workloads = sorted({os.path.basename(p)[:-len("_bench.json")] for p in glob.glob(os.path.join(src, "*_bench.json"))}
                       | {os.path.basename(p)[:-len(sfx)] for sfx in ("_stats", "_fetch") for p in glob.glob(os.path.join(src, "*" + sfx)) if os.path.isdir(p)})
    for w in workloads:
        st = find(os.path.join(src, w + "_stats"), "*kernel_stats.csv")
        if st:
            rows = list(csv.DictReader(open(st)))
            out = os.path.join(ROOT, "profiles", f"{tag}_{w}_kernel_stats.csv")
            with open(out, "w", newline="") as f:
                wr = csv.writer(f)
                wr.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
                for r in rows:
                    wr.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.2f}',
                                 f'{float(r["MinNs"]) / 1e3:.2f}', f'{float(r["MaxNs"]) / 1e3:.2f}', r["Percentage"]])
            print("wrote", out)
        acc = defaultdict(lambda: {"fetch_kb_raw": 0.0, "write_kb": 0.0, "n_f": 0, "n_w": 0})
        for d, key, cnt in ((w + "_fetch", "fetch_kb_raw", "n_f"), (w + "_write", "write_kb", "n_w")):
            f = find(os.path.join(src, d), "*counter_collection.csv")
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
            out = os.path.join(ROOT, "profiles", f"{tag}_{w}_hbm_traffic.json")
            json.dump({"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, over `python3 bench.py --workload {w} "
                               "(2-3 steps) --no-cpu-baseline --no-roofline`; per-launch averages. Counter unit = KB; FETCH_SIZE doubled "
                               "(gfx950 reports half of wide coalesced 16-B/lane reads, MI355X_MICROARCH.md HBM section).", "kernels": kern},
                      open(out, "w"), indent=1)
            print("wrote", out)
        b = os.path.join(src, w + "_bench.json")
        if os.path.exists(b):
            line = [l for l in open(b).read().splitlines() if l.startswith("{")]
            if line:
                open(os.path.join(ROOT, "profiles", f"{tag}_{w}_bench.json"), "w").write(line[-1] + "\n")
    r = os.path.join(src, "train31_rccl_world1.json")
    if os.path.exists(r):
        line = [l for l in open(r).read().splitlines() if l.startswith("{")]
        if line:
            open(os.path.join(ROOT, "profiles", f"{tag}_train31_rccl_world1_bench.json"), "w").write(line[-1] + "\n")


if __name__ == "__main__":
    main()

"""Per-kernel HBM-side traffic of one traced forward: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) over
tools/profile_forward.py, grouped by (kernel name, grid size) — which launches move more bytes than their algorithm needs.

    python tools/pmc_traffic_by_kernel.py <fetch counter_collection.csv> <write counter_collection.csv>

FETCH_SIZE is doubled (gfx950: MI355X_MICROARCH.md, HBM section), both are KB; fabric-side (Infinity-Cache hits included).
Dispatches of the warm-up forwards are included: figures are means per dispatch of a (kernel, grid)."""
import collections
import csv
import re
import sys


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        key = (name, int(r.get("Grid_Size", 0) or 0))
        tot[key] += float(r["Counter_Value"])
        n[key].add(r["Dispatch_Id"])
    return tot, n


def main():
    ft, fn = load(sys.argv[1], "FETCH_SIZE")
    wt, wn = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for key in ft:
        nf = len(fn[key])
        f = 2.0 * ft[key] * 1024 / nf
        w = wt.get(key, 0.0) * 1024 / max(len(wn.get(key, ())), 1)
        rows.append((nf * (f + w), key, nf, f, w))
    rows.sort(reverse=True)
    print(f"{'kernel':60s} {'grid':>9s} {'n':>5s} {'fetch MB':>9s} {'write MB':>9s} {'total GB (all n)':>16s}")
    for tot, (name, grid), nf, f, w in rows[:70]:
        print(f"{name[:60]:60s} {grid:9d} {nf:5d} {f/1e6:9.1f} {w/1e6:9.1f} {tot/1e9:16.2f}")


if __name__ == "__main__":
    main()

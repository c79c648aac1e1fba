"""Effective shader clock per kernel family over one forward: GRBM_GUI_ACTIVE (cycles the graphics engine was busy) divided by the
dispatch's duration, from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass over tools/profile_forward.py.  The chip clocks to
its power budget (MI355X_MICROARCH.md "DVFS give-back"): a family that runs well below the 2.4 GHz nominal is power-limited.
The counter is the SUM over the 8 XCDs (each has its own GRBM): divided by 8 here.  GUI_ACTIVE also covers the dispatch overhead
around a kernel, so the ratio overshoots for launches of a few microseconds: only families with mean durations >= 100 us are a
clock measurement (marked *).
    python tools/pmc_clock.py <counter_collection.csv> [<kernel_trace.csv>]"""
import collections
import csv
import sys

from pmc_family import family


def main(counters, trace=None):
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace)):
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    cyc, ns, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(counters)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        f = family(r["Kernel_Name"])
        if not f:
            continue
        d = dur.get(r["Dispatch_Id"])
        if d is None and "End_Timestamp" in r:
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if not d:
            continue
        cyc[f] += float(r["Counter_Value"])
        ns[f] += d
        n[f] += 1
    print(f"{'family':28s} {'dispatches':>10s} {'mean us':>9s} {'GUI_ACTIVE / 8 / duration':>26s}")
    for f in sorted(cyc, key=lambda k: -ns[k]):
        us = ns[f] / n[f] / 1e3
        print(f"{f:28s} {n[f]:10d} {us:9.1f} {cyc[f] / 8 / ns[f]:22.3f} GHz {'*' if us >= 100 else ''}")


if __name__ == "__main__":
    main(*sys.argv[1:3])

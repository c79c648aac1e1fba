"""In-situ timeline of ONE denoising forward from a rocprofv3 --kernel-trace CSV of tools/profile_forward.py:
per dispatch the kernel, its grid, its duration and the idle gap in front of it; totals per kernel family; the sum of gaps.
    python tools/trace_forward.py <kernel_trace.csv> [--all]
The forward is found as the last repeating period of the kernel-name sequence (profile_forward runs the same forward
several times back to back)."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def main(path, show_all=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    n = len(names)
    period = None
    for P in range(300, n // 2):
        if names[n - P:] == names[n - 2 * P:n - P]:
            period = P
            break
    if period is None:
        raise SystemExit("no repeating forward found")
    # the second-to-last period: a plain timed forward (the last one carries the per-launch HIP events of the roofline bracket)
    fw = rows[n - 2 * period:n - period]
    t0, t1 = int(fw[0]["Start_Timestamp"]), int(fw[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
    print(f"# forward = {period} dispatches, wall {1e-6*(t1-t0):.3f} ms, kernel time {1e-6*busy:.3f} ms, idle gaps {1e-6*(t1-t0-busy):.3f} ms")
    fam = collections.defaultdict(lambda: [0, 0.0])
    prev_end = t0
    lines = []
    for r in fw:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = short(r["Kernel_Name"])
        g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        lines.append((nm, g, wg, (e - s) * 1e-3, (s - prev_end) * 1e-3))
        prev_end = max(prev_end, e)
        f = fam[nm]
        f[0] += 1
        f[1] += (e - s) * 1e-3
    print("# per kernel (calls, total us, mean us, share)")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:80]:80s} {c:5d} {t:10.1f} {t/c:9.1f} {100*t/(busy*1e-3):6.2f}%")
    # grouped by (kernel, grid): the same shape launched repeatedly
    grp = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for nm, g, wg, d, gap in lines:
        q = grp[(nm, g)]
        q[0] += 1
        q[1] += d
        q[2] = min(q[2], d)
        q[3] = max(q[3], d)
    print("# per (kernel, grid) (calls, total us, mean, min, max)")
    for (nm, g), (c, t, lo, hi) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{nm[:64]:64s} grid {g:>9s} {c:4d} {t:9.1f} {t/c:8.1f} {lo:8.1f} {hi:8.1f}")
    if show_all:
        print("# timeline (kernel, grid, wg, us, gap us)")
        for nm, g, wg, d, gap in lines:
            print(f"{nm[:64]:64s} {g:>9s} {wg:>5s} {d:9.1f} {gap:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], "--all" in sys.argv)

"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into per-kernel statistics (text).
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/<name>.txt"""
import re
import sqlite3
import sys


def main(path, top=40):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for name, n, s, avg, mn, mx in rows[:top]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        print(f"{s/1e6:10.3f} {100*s/tot:6.2f} {n:7d} {avg/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f}  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1])

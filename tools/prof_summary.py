#!/usr/bin/env python3
"""Per-kernel summary (count / avg / min / max / total) of a rocprofv3 --kernel-trace rocpd database.

    python tools/prof_summary.py gpurun_out/prof_x/p_results.db [> profiles/rNN_x.txt]
"""
import sqlite3
import sys


def summarize(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    q = (f"select s.{name_col}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"sum(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), "
         f"max(d.private_segment_size) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} "
         f"order by 6 desc")
    rows = list(cur.execute(q))
    total = sum(r[5] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>8s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s} "
          f"{'grid':>9s} {'wg':>5s} {'lds':>6s} {'scratch':>7s}")
    for name, cnt, avg, mn, mx, tot, grid, wg, lds, scr in rows:
        print(f"{name[:70]:70s} {cnt:8d} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {tot/1e6:10.3f} "
              f"{100.0*tot/total:6.1f} {grid:9d} {wg:5d} {lds:6d} {scr:7d}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("==", p)
        summarize(p)

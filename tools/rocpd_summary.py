"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table
(the same content as rocprofv3's kernel_stats.csv): name, calls, total/avg/min/max ns, %.
usage: python tools/rocpd_summary.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, grid_x, grid_y, grid_z, vgpr_count, accum_vgpr_count, count(*), sum(duration), avg(duration), "
        "min(duration), max(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
    total = sum(r[7] for r in rows) or 1
    return [dict(name=r[0], grid=f"{r[1]}x{r[2]}x{r[3]}", vgpr=r[4], agpr=r[5], calls=r[6], total_ns=r[7],
                 avg_ns=round(r[8], 1), min_ns=r[9], max_ns=r[10], pct=round(100.0 * r[7] / total, 2)) for r in rows]


if __name__ == "__main__":
    rows = summarise(sys.argv[1])
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)

"""dev: kernel timeline (all queues) of a steady-state window from a rocprofv3 rocpd db: python tools/lane_timeline.py db [t_ms_window]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
rows = db.execute("select name, start, end, queue_id, grid_x, grid_y, grid_z from kernels order by start").fetchall()
heads = [i for i, r in enumerate(rows) if 'add_matmul_kn_kernel' in r[0]]
start = heads[len(heads) * 2 // 3]
t0 = rows[start][1]
for r in rows[start:]:
    if (r[1] - t0) / 1e6 > win:
        break
    n = r[0].replace('mphip::', '').replace('void ', '')
    n = n[:n.find('(')] if '(' in n else n
    print(f"{(r[1]-t0)/1e3:8.1f} +{(r[2]-r[1])/1e3:7.1f} q{r[3]} {n[:44]} [{r[4]}x{r[5]}x{r[6]}]")

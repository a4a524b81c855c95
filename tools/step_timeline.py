"""Print the kernel timeline of the last step in a rocpd db: python tools/step_timeline.py db [max_rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = db.execute("select name, start, end, queue_id, grid_x, grid_y, grid_z from kernels order by start").fetchall()
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if 'rt_theta' in n]
# a step has 2 rt_theta launches; the last step starts a bit before the 2nd-to-last rt_theta's generator
starts = [i for i, n in enumerate(names) if 'add_matmul_kn_kernel' in n]
start = starts[-4]
t0 = rows[start][1]
end_t = max(r[2] for r in rows[start:])
print(f"step wall {(end_t - t0)/1e3:.1f} us, kernels {len(rows)-start}")
for r in rows[start:start + limit]:
    n = r[0].replace('mphip::', '').replace('void ', '')[:46]
    print(f"{(r[1]-t0)/1e3:8.1f} +{(r[2]-r[1])/1e3:7.1f} q{r[3]} {n} [{r[4]}x{r[5]}x{r[6]}]")

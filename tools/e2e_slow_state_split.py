"""Splits the kernel trace of tools/e2e_slow_state_trace.py at its erfinv markers and prints, per kernel name, ms per step in instance 1 vs
instance 2 (4 steps each), sorted by the difference.  usage: e2e_slow_state_split.py <rocpd db>"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "erfinv" in r[0]]
assert len(marks) == 4, marks
seg = [rows[marks[0] + 1:marks[1]], rows[marks[2] + 1:marks[3]]]
agg = [collections.Counter(), collections.Counter()]
cnt = [collections.Counter(), collections.Counter()]
for k in range(2):
    for name, s, e in seg[k]:
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:90]
        agg[k][n] += e - s; cnt[k][n] += 1
names = sorted(set(agg[0]) | set(agg[1]), key=lambda n: -abs(agg[0][n] - agg[1][n]))
print(f"total kernel ms/step: instance 1 {sum(agg[0].values()) / 4e6:.2f}, instance 2 {sum(agg[1].values()) / 4e6:.2f}; wall ms/step: "
      f"{(seg[0][-1][2] - seg[0][0][1]) / 4e6:.2f} vs {(seg[1][-1][2] - seg[1][0][1]) / 4e6:.2f}")
for n in names[:25]:
    print(f"{agg[0][n] / 4e6:8.3f} {agg[1][n] / 4e6:8.3f} ms/step  calls {cnt[0][n] / 4:6.1f} {cnt[1][n] / 4:6.1f}  {n}")

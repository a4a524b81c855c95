"""Aggregate a rocprofv3 rocpd database by kernel name: ms per step.  usage: agg_summary.py <db> <steps>"""
import collections, csv, re, subprocess, sys, os
db, steps = sys.argv[1], float(sys.argv[2])
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "rocpd_summary.py"), db], capture_output=True, text=True).stdout
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(out.splitlines()):
    n = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["name"]))
    agg[n][0] += int(r["total_ns"]); agg[n][1] += int(r["calls"])
tot = sum(v[0] for v in agg.values())
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{t / steps / 1e6:8.3f} ms/step {c / steps:6.1f} calls/step {100 * t / tot:5.1f}%  {n[:100]}")
print(f"total {tot / steps / 1e6:.3f} ms/step")

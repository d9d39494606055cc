"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) per kernel name.
usage: python tools/trace_summary.py <results.db> [steps] > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, start, end from kernels order by start"))
agg = {}
for name, s, e in rows:
    n = re.sub(r"\(.*", "", name.replace("frcnn::", "").replace("void ", ""))
    a = agg.setdefault(n, [0, 0.0, 1e18, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] = min(a[2], (e - s) / 1e3); a[3] = max(a[3], (e - s) / 1e3)
tot = sum(a[1] for a in agg.values())
span = (rows[-1][2] - rows[0][1]) / 1e3
print("| kernel | calls | total us | avg us | min us | max us | % of kernel time |")
print("|---|---|---|---|---|---|---|")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (n, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
print()
print("kernel time total %.1f us over %d dispatches; first-to-last span %.1f us; busy fraction %.3f" % (tot, len(rows), span, tot / span))
if steps > 1:
    print("per step (%d steps incl. warmup): %.1f us kernel time" % (steps, tot / steps))

"""Timeline of one training step from a rocprofv3 kernel_trace.csv of bench.py: per queue (HIP stream) the
launches with start offset, duration and the idle gap before them.  usage: python tools/timeline.py <dir> [step_index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rmsprop" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a, b = idx[k], idx[k + 1]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
print("step span %.1f us, %d launches" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, len(step)))
last_end = {}
busy_any = 0
ev = []
for r in step:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].replace("void ", "").replace("frcnn::", "")
    name = name[:name.index("(")] if "(" in name else name
    print("q%-2s +%8.1f us  dur %7.1f  gap %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur = 0; lastt = ev[0][0]; idle = 0
for t, d in ev:
    if cur == 0: idle += t - lastt
    cur += d; lastt = t
print("GPU idle (no kernel on any queue): %.1f us" % (idle / 1e3))

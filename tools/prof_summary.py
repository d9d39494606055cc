"""Summarise a rocprofv3 --stats kernel_stats.csv of bench.py (23 steps): per-step time per kernel."""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 23
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step: %.3f" % (tot / 1e6 / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-84s calls/step %5.1f  avg %8.1f us  per-step %7.1f us  %5.1f%%" % (r['Name'][:84], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / steps, float(r['Percentage'])))

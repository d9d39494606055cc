"""Per-kernel evidence table (north_star: "the split evidenced by rocprof HBM GB/s and MFMA-busy counters"):
MFMA-busy fraction from SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, HBM bytes from FETCH_SIZE (x2, gfx950) +
WRITE_SIZE, durations from the kernel traces of the same runs.
usage: python tools/pmc_summary.py <mfma_dir> <fetch_dir> <write_dir> <out.csv>"""
import collections, csv, glob, re, sys


def short(n):
    n = re.sub(r"^void |frcnn::", "", n)
    return n[:n.index("(")] if "(" in n else n


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(d):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])):
        acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc


mf, fe, wr = counters(sys.argv[1]), counters(sys.argv[2]), counters(sys.argv[3])
dur = durations(sys.argv[1])
rows = []
for k in mf:
    busy = sum(mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    act = sum(mf[k].get("GRBM_GUI_ACTIVE", [0]))
    n = len(mf[k].get("GRBM_GUI_ACTIVE", [])) or 1
    frac = (busy / 1024.0) / (act / 8.0) if act else 0.0   # 1024 SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs
    f = 2.0 * 1024.0 * sum(fe.get(k, {}).get("FETCH_SIZE", [0])) / max(len(fe.get(k, {}).get("FETCH_SIZE", [])), 1)
    w = 1024.0 * sum(wr.get(k, {}).get("WRITE_SIZE", [0])) / max(len(wr.get(k, {}).get("WRITE_SIZE", [])), 1)
    us = sum(dur[k]) / len(dur[k]) / 1e3 if dur.get(k) else 0.0
    rows.append((sum(dur.get(k, [0])), k, n, us, 100.0 * frac, (f + w) / 1e6, (f + w) / 1e3 / us if us else 0.0))
rows.sort(reverse=True)
with open(sys.argv[4], "w") as o:
    o.write("kernel,dispatches,avg_us,mfma_busy_percent,hbm_MB_per_launch,hbm_GB_per_s\n")
    for _, k, n, us, fr, mb, gbs in rows:
        o.write('"%s",%d,%.1f,%.1f,%.2f,%.0f\n' % (k, n, us, fr, mb, gbs))
        print("%-62s n=%4d  %8.1f us  MFMA busy %5.1f%%  HBM %8.2f MB  %6.0f GB/s" % (k[:62], n, us, fr, mb, gbs))

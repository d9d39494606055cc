"""How many ROUNDS of resident blocks each kernel of a step takes (blocks / (256 CUs x blocks that fit a CU by LDS and registers)):
a launch of 1.05 rounds runs for two block times.  usage (GPU box): python tools/rounds.py <rocprofv3 --kernel-trace output dir>"""
import collections, csv, glob, math, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.OrderedDict()
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    blocks = grid // wg
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    regs = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
    waves = (wg + 63) // 64
    per_simd = min(8, 512 // max(8, (regs + 7) // 8 * 8)) if regs else 8
    by_regs = per_simd * 4 // waves if waves <= per_simd * 4 else 0
    by_lds = (160 * 1024) // lds if lds else 99
    occ = max(1, min(by_regs, by_lds, 32 // max(1, waves) if waves else 32))
    key = (r["Kernel_Name"][:70], blocks, wg, lds, regs)
    a = acc.setdefault(key, [0, 0.0, occ])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-72s %7s %5s %7s %5s %4s %7s %6s %8s" % ("kernel", "blocks", "wg", "lds", "regs", "occ", "rounds", "n", "avg us"))
for (name, blocks, wg, lds, regs), (n, us, occ) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    rounds = blocks / (256.0 * occ)
    flag = " <--" if 1.0 < rounds < 1.25 or 2.0 < rounds < 2.15 else ""
    print("%-72s %7d %5d %7d %5d %4d %7.2f %6d %8.1f%s" % (name, blocks, wg, lds, regs, occ, rounds, n, us / n, flag))

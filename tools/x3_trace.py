"""Summary of a conv_x3 phase trace (tools/x3_trace.sh): per block t0 entry, t1 loop start, t2 loop end, t3 stores issued,
t4 stores drained (s_memrealtime, 100 MHz -> 10 ns), word 7 = XCC id << 32 | HW_ID."""
import sys
import numpy as np

path = sys.argv[1]
head = open(path).readline().strip()
a64 = np.loadtxt(path, dtype=np.uint64, comments="#").reshape(-1, 64)
a = a64[:, :32]
mt = a64[:, 32:]
t = a[:, :5].astype(np.float64) * 0.01   # microseconds
t -= t[:, 0].min()
cu = (a[:, 7] >> np.uint64(32)) * np.uint64(1 << 16) + ((a[:, 7] >> np.uint64(8)) & np.uint64(0xFF))
print(head)
print("blocks %d on %d distinct CUs; blocks per CU: %s" % (len(a), len(set(cu.tolist())), np.bincount(np.unique(cu, return_counts=True)[1]).tolist()))


def q(x):
    return "min %.1f  p10 %.1f  med %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))


print("entry           ", q(t[:, 0]))
print("prologue (us)   ", q(t[:, 1] - t[:, 0]))
print("loop (us)       ", q(t[:, 2] - t[:, 1]))
print("stores issued   ", q(t[:, 3] - t[:, 2]))
print("stores drained  ", q(t[:, 4] - t[:, 3]))
print("loop end at     ", q(t[:, 2]))
print("block end at    ", q(t[:, 4]))
late = t[:, 0] > 5.0
print("blocks entering later than 5 us: %d (their entry: %s)" % (late.sum(), q(t[late, 0]) if late.any() else "-"))
# matrix-pipe view: how many blocks are inside their loop at each instant
grid = np.linspace(0, t[:, 4].max(), 41)
inloop = [(int(((t[:, 1] <= g) & (t[:, 2] > g)).sum())) for g in grid]
print("blocks inside the loop at 40 instants:", inloop)

# progress per CU: chunk-boundary stamps (words 8..31) -> aggregate progress rate against the number of blocks still in their loop
tc = a[:, 8:32].astype(np.float64) * 0.01
t00 = (a[:, 0].astype(np.float64) * 0.01).min()
nch = int((a[0, 8:32] > 0).sum())
if nch >= 2:
    tc = tc[:, :nch] - t00
    ends = t[:, 2]
    NB = int(np.bincount(np.unique(cu, return_counts=True)[1]).argmax())
    rate = {n: [] for n in range(1, NB + 1)}
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        if len(idx) != NB:
            continue
        order = idx[np.argsort(ends[idx])]
        # events: chunk completions (chunk j complete at stamp j+1, the last at loop end)
        comp = np.concatenate([np.concatenate([tc[b, 1:], [ends[b]]]) for b in order])
        e = np.sort(ends[order])
        bounds = [tc[order, 0].max()] + e.tolist()
        for k, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            if hi - lo > 1.0:
                rate[NB - k].append(((comp > lo) & (comp <= hi)).sum() / (hi - lo))
    for n in range(NB, 0, -1):
        if rate[n]:
            print("%d block(s) in the loop on a CU: %.3f chunks/us (CU median; %d CUs), i.e. %.2f us per chunk per CU" % (n, np.median(rate[n]), len(rate[n]), 1 / np.median(rate[n])))
    # per-block chunk durations by finishing order
    for rank in range(NB):
        d = []
        for c in set(cu.tolist()):
            idx = np.where(cu == c)[0]
            if len(idx) != NB:
                continue
            b = idx[np.argsort(ends[idx])][rank]
            d.append(np.diff(np.concatenate([tc[b], [ends[b]]])))
        print("block finishing #%d on its CU, us per chunk:" % (rank + 1), np.round(np.median(np.array(d), axis=0), 1).tolist())

# shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz) per chunk of the block that finishes last on its CU
if nch >= 2:
    clk = []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        if len(idx) != NB:
            continue
        b = idx[np.argsort(ends[idx])][NB - 1]
        r = a64[b, 8:8 + nch].astype(np.float64); m = a64[b, 40:40 + nch].astype(np.float64)
        clk.append(np.diff(m) / np.diff(r) * 0.1)   # GHz
    print("shader clock (GHz) per chunk interval of the last-finishing block:", np.round(np.median(np.array(clk), axis=0), 2).tolist())
    whole = (mt[:, 2].astype(np.float64) - mt[:, 1].astype(np.float64)) / (a[:, 2].astype(np.float64) - a[:, 1].astype(np.float64)) * 0.1
    print("shader clock over each block's loop:", q(whole))

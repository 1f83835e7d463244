"""Per-item record of the LEAN forward item in a -DART_DEBUG_TIMELINE build (tools/timeline.sh): duration, stray rays, un-park
events and window size against the item's place in the queue.  usage: python tools/timeline_lean_report.py /tmp/timeline.bin"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
idx = np.nonzero(raw[:, 1] > 0)[0]
rec = raw[idx]
t0, t1, t2 = (rec[:, k].astype(np.int64) * 10 for k in (1, 2, 6))      # ns
total = (t2 - t0) / 1e3
strays = (rec[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
unparks = (rec[:, 3] >> np.uint64(32)).astype(np.int64)
npass = (rec[:, 4] >> np.uint64(40)).astype(np.int64)
tw = ((rec[:, 4] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.int64)
th = (rec[:, 4] & np.uint64(0xFFFFF)).astype(np.int64)
print(f"{len(rec)} items; span {(t2.max() - t0.min()) / 1e6:.3f} ms; mean item {total.mean():.1f} us; sum of items / 256 CUs {total.sum() / 256 / 1e3:.3f} ms")
dec = np.array_split(np.arange(len(rec)), 10)
row = lambda name, v, f="{:.0f}": print(f"  {name:34s} " + " ".join(f.format(v[d].mean()).rjust(8) for d in dec))
row("item duration (us)", total)
row("window phase (us)", (t1 - t0) / 1e3, "{:.1f}")
t7 = rec[:, 7].astype(np.int64) * 10
row("trace phase (us)", (t7 - t1) / 1e3, "{:.1f}")
row("flush phase (us)", (t2 - t7) / 1e3, "{:.1f}")
row("stray rays per item", strays)
row("un-park events per item", unparks)
row("window columns", tw)
row("window rows", th)
row("passes", npass, "{:.2f}")
if (rec[:, 5] > 0).all():
    row("shader clock (MHz)", rec[:, 5].astype(np.float64) / np.maximum(total, 1e-3), "{:.0f}")
A = np.stack([np.ones(len(rec)), strays, unparks], axis=1)
coef, *_ = np.linalg.lstsq(A, total, rcond=None)
print(f"  least squares: item = {coef[0]:.1f} us + {coef[1] * 1e3:.2f} ns per stray ray + {coef[2] * 1e3:.1f} ns per un-park event; residual std {np.std(total - A @ coef):.1f} us")
hw = rec[:, 0]
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
hwid = hw & np.uint64(0xFFFFFFFF)
cu = ((hwid >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64)
sh = ((hwid >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
se = ((hwid >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
place = (xcc << 16) | (se << 8) | (sh << 4) | cu
pairs = (xcc << 16) | (se << 8) | (sh << 4) | (cu >> 1)
print(f"  placement: {len(np.unique(place))} CUs on {len(np.unique(xcc))} XCDs, {len(np.unique(pairs))} CU pairs; items per XCD " +
      " ".join(str(int((xcc == x).sum())) for x in np.unique(xcc)))
# does an item run faster while the other CU of its pair (shared instruction / scalar caches) is idle?
order = np.argsort(t0)
busy_frac = np.zeros(len(rec))
for i in range(len(rec)):
    sib = (pairs == pairs[i]) & (place != place[i])
    if not sib.any():
        continue
    ov = np.minimum(t2[sib], t2[i]) - np.maximum(t0[sib], t0[i])
    busy_frac[i] = np.clip(ov, 0, None).sum() / max(t2[i] - t0[i], 1)
mid = (np.arange(len(rec)) >= len(rec) // 5)          # (not the first round: far heliostats, lock step)
for lo, hi in ((0.0, 0.25), (0.25, 0.75), (0.75, 1.01)):
    sel = mid & (busy_frac >= lo) & (busy_frac < hi)
    if sel.any():
        print(f"  items whose pair CU was busy {lo:.2f}-{hi:.2f} of the time: {int(sel.sum())} items, mean {total[sel].mean():.1f} us")

"""Phase medians of the forward workgroups recorded by a -DART_DEBUG_TIMELINE build (tools/timeline.sh)."""
import sys

import numpy as np

rec = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
rec = rec[rec[:, 1] > 0]
hw = rec[:, 0]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
hwid = hw & np.uint64(0xFFFFFFFF)
cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
place = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
t = rec[:, 1:7].astype(np.int64) * 10          # ns (100 MHz counter)
names = ["window (phase 1)", "zero the tile", "first group of 4 samples (load latency + trace)", "rest of the trace",
         "flush"]
print(f"{len(rec)} workgroups on {len(np.unique(place))} CUs; kernel span {(t[:, 5].max() - t[:, 0].min()) / 1e6:.3f} ms")
dur = np.diff(t, axis=1)
for k, name in enumerate(names):
    d = dur[:, k] / 1e3
    print(f"  {name:50s} median {np.median(d):8.2f} us   p10 {np.percentile(d, 10):8.2f}   p90 {np.percentile(d, 90):8.2f}")
total = (t[:, 5] - t[:, 0]) / 1e3
print(f"  {'workgroup, first stamp to last':50s} median {np.median(total):8.2f} us   p10 {np.percentile(total, 10):8.2f}   p90 {np.percentile(total, 90):8.2f}")
gaps = []
for p_ in np.unique(place):
    sel = np.argsort(t[place == p_, 0])
    tt = t[place == p_][sel]
    gaps.extend(((tt[1:, 0] - tt[:-1, 5]) / 1e3).tolist())
gaps = np.asarray(gaps)
print(f"  {'gap between two workgroups on a CU':50s} median {np.median(gaps):8.2f} us   p10 {np.percentile(gaps, 10):8.2f}   p90 {np.percentile(gaps, 90):8.2f}")
print(f"      mean {gaps.mean():.2f} us; p75 {np.percentile(gaps, 75):.2f}  p95 {np.percentile(gaps, 95):.2f}  p99 {np.percentile(gaps, 99):.2f}  max {gaps.max():.2f}; "
      f"share of gaps > 5 us: {(gaps > 5).mean():.3f}")
first = np.array([t[place == p_, 0].min() for p_ in np.unique(place)]) - t[:, 0].min()
last = t[:, 5].max() - np.array([t[place == p_, 5].max() for p_ in np.unique(place)])
print(f"      ramp: CUs start {first.mean() / 1e3:.2f} us after the first on average (max {first.max() / 1e3:.2f}); "
      f"tail: CUs finish {last.mean() / 1e3:.2f} us before the last on average (max {last.max() / 1e3:.2f})")
per_cu = np.array([(place == p_).sum() for p_ in np.unique(place)])
print(f"      workgroups per CU: min {per_cu.min()} max {per_cu.max()}")
cyc = rec[:, 7].astype(np.float64)
mhz = cyc / (total * 1e-6) / 1e6
print(f"  shader clock over a workgroup (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz   p10 {np.percentile(mhz, 10):.0f}   p90 {np.percentile(mhz, 90):.0f}")
busy = total.sum() / 1e3
span = (t[:, 5].max() - t[:, 0].min()) / 1e6
print(f"  CU occupancy by stamped time: {busy / (span * len(np.unique(place))):.3f}")
# item duration against its place in the queue (slot = item index): systematic differences would let the queue start with the long items
idx = np.nonzero(np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:, 1] > 0)[0]
if len(idx) >= 20:
    dec = np.array_split(np.argsort(idx), 10)
    print("      item duration by decile of the queue position (us): " + " ".join(f"{total[d].mean():.0f}" for d in dec))
    print(f"      item duration: mean {total.mean():.1f} us, std {total.std():.1f}, min {total.min():.1f}, max {total.max():.1f}")
    print("      shader clock by decile (MHz): " + " ".join(f"{mhz[d].mean():.0f}" for d in dec))
    start = (t[:, 0] - t[:, 0].min()) / 1e3
    print("      item start by decile (us after the first): " + " ".join(f"{start[d].mean():.0f}" for d in dec))
    trace_d = dur[:, 3] / 1e3
    print("      trace phase by decile (us): " + " ".join(f"{trace_d[d].mean():.0f}" for d in dec))
    print("      window phase by decile (us): " + " ".join(f"{(dur[:, 0] / 1e3)[d].mean():.1f}" for d in dec))
    if (rec[:, 5] > rec[:, 2]).all() and (rec[:, 5] < rec[:, 3]).all():      # lean backward item: slot 5 = end of the edge partition
        pk = (rec[:, 5].astype(np.int64) - rec[:, 2].astype(np.int64)) * 10 / 1e3
        st = (rec[:, 3].astype(np.int64) - rec[:, 5].astype(np.int64)) * 10 / 1e3
        print(f"      lean backward item: edge partition median {np.median(pk):.2f} us, staging of dL/dflux median {np.median(st):.2f} us")

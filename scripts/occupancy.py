"""Resident-wave timeline from the LC_DEBUG_FLAGS=7168 dump: s_memrealtime (100 MHz, 10 ns ticks) at wave start / end."""
import sys
import numpy as np
c = np.load(sys.argv[1]).astype(np.uint32)
ts, te = (c >> 16).astype(np.int64), (c & 0xFFFF).astype(np.int64)
ref = int(np.median(ts))
ts = ((ts - ref + 32768) % 65536) - 32768
te = ((te - ref + 32768) % 65536) - 32768
te = np.where(te < ts, te + 65536, te)
t0 = ts.min()
ts, te = ts - t0, te - t0
dur = te - ts
span = int(te.max())
print("waves %d, span %.1f us, wave duration mean %.2f us p5 %.2f p50 %.2f p95 %.2f" % (
    len(c), span / 100, dur.mean() / 100, np.percentile(dur, 5) / 100, np.median(dur) / 100, np.percentile(dur, 95) / 100))
print("mean resident waves %.0f (%.1f per CU)" % (dur.sum() / span, dur.sum() / span / 256))
for q in (0.05, 0.1, 0.2, 0.3, 0.5, 0.7, 0.9, 0.95):
    t = int(span * q)
    print("  t=%5.1f us resident %d" % (t / 100, int(((ts <= t) & (te > t)).sum())))

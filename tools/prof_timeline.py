#!/usr/bin/env python3
"""Timeline view of a rocprofv3 rocpd kernel trace of bench.py: for the graph-replayed steps (delimited by
step_advance_kernel), how much of a step's wall time is idle (no kernel running), covered by exactly one kernel
("exposed": attributed to that kernel's name) or by two or more (overlapped).  What bounds wall time is the exposed +
idle part, not the per-kernel sums of prof_summary.py.

usage: python tools/prof_timeline.py <db> [title]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "step_advance" in r[0]]
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# kernels view columns:", cols)
print("# %d kernels, %d step marks" % (len(rows), len(marks)))
if len(marks) < 4:
    sys.exit(0)
# the last 3 complete steps
lo, hi = marks[-4], marks[-1]
seg = rows[lo:hi]
nsteps = 3
t0, t1 = seg[0][1], max(r[2] for r in seg)
ev = []
for i, (n, s, e) in enumerate(seg):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
active = set()
last = t0
idle = 0
overl = 0
exposed = defaultdict(float)
gap_after = defaultdict(float)
last_ended = None
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        if not active:
            idle += dt
            if last_ended is not None:
                gap_after[seg[last_ended][0]] += dt
        elif len(active) == 1:
            exposed[seg[next(iter(active))][0]] += dt
        else:
            overl += dt
    last = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
        last_ended = i
wall = (t1 - t0) / 1e6 / nsteps
ksum = sum(e - s for _, s, e in seg) / 1e6 / nsteps
clean = lambda n: re.sub(r'\(anonymous namespace\)::', '', n)[:100]
print("per step: wall %.3f ms | kernel-time sum %.3f ms | idle (no kernel) %.3f ms | exactly one kernel %.3f ms | >=2 kernels %.3f ms | launches %d"
      % (wall, ksum, idle / 1e6 / nsteps, sum(exposed.values()) / 1e6 / nsteps, overl / 1e6 / nsteps, len(seg) // nsteps))
print("exposed time by kernel (ms/step):")
for n, v in sorted(exposed.items(), key=lambda kv: -kv[1])[:32]:
    print("  %8.3f  %s" % (v / 1e6 / nsteps, clean(n)))
print("idle gaps by preceding kernel (ms/step):")
for n, v in sorted(gap_after.items(), key=lambda kv: -kv[1])[:16]:
    print("  %8.3f  %s" % (v / 1e6 / nsteps, clean(n)))
# the launch sequence of the last complete step: start offset, duration, gap since the latest end of anything before it
lo2 = marks[-2]
seq = rows[lo2:marks[-1]]
print("sequence of the last step (t_us, dur_us, idle_before_us, name):")
base = seq[0][1]
latest = seq[0][1]
for n, s, e in seq:
    print("  %9.1f %7.1f %6.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, max(0.0, (s - latest) / 1e3), clean(n)[:70]))
    latest = max(latest, e)

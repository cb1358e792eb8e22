#!/usr/bin/env python3
"""One replayed decode token from a rocprofv3 rocpd kernel trace (tokens are delimited by decode_prepare_kernel): every
launch with its duration and the gap since the previous kernel ended, plus per-kernel totals.

usage: python tools/prof_decode.py <db> [title]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "decode_prepare" in r[0] or "dec_finish" in r[0]]
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# %d kernels, %d tokens" % (len(rows), len(marks)))
if len(marks) < 12:
    sys.exit(0)
lo, hi = marks[-11], marks[-1]
tok = 10
wall = (rows[hi][1] - rows[lo][1]) / tok / 1e3
print("# last %d tokens: %.1f us per token, %.1f launches per token" % (tok, wall, (hi - lo) / tok))
tot, cnt, gaps = defaultdict(float), defaultdict(int), defaultdict(float)
for i in range(lo, hi):
    n = rows[i][0].split("(")[0][-60:]
    tot[n] += (rows[i][2] - rows[i][1]) / 1e3
    cnt[n] += 1
    gaps[n] += max(0, rows[i][1] - rows[i - 1][2]) / 1e3
print("%10s %6s %10s %10s  kernel" % ("us/token", "calls", "avg_us", "gap_before"))
for n in sorted(tot, key=lambda k: -tot[k]):
    print("%10.1f %6.1f %10.2f %10.2f  %s" % (tot[n] / tok, cnt[n] / tok, tot[n] / cnt[n], gaps[n] / cnt[n], n))
print("# kernel time %.1f us/token, gaps %.1f us/token" % (sum(tot.values()) / tok, sum(gaps.values()) / tok))
print("# one token in launch order:")
for i in range(marks[-2], marks[-1]):
    print("%8.2f us  gap %6.2f  %s" % ((rows[i][2] - rows[i][1]) / 1e3, max(0, rows[i][1] - rows[i - 1][2]) / 1e3, rows[i][0].split("(")[0][-70:]))

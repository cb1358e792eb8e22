#!/usr/bin/env python3
"""Launch-by-launch listing of ONE graph-replayed training step from a rocprofv3 rocpd kernel trace of bench.py: position, start
offset, duration, gap to the previous kernel, grid / workgroup size and name -- the picture of where a step of ~210 dependent
launches spends its time (which launches are a few dozen workgroups on 256 CUs, which run longer in the graph than alone).

usage: python tools/prof_sequence.py <db> <out.txt> """
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    want = [k for k in ("grid_x", "grid_size_x", "grid_size", "workgroup_x", "workgroup_size_x", "workgroup_size", "lds_size", "lds_block_size") if k in cols]
    rows = c.execute("select name, start, end %s from kernels order by start" % "".join(", " + w for w in want)).fetchall()
    marks = [i for i, r in enumerate(rows) if "step_advance" in r[0]]
    if len(marks) < 3:
        raise SystemExit("fewer than 2 complete replayed steps in the trace")
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = seg[0][1]
    with open(out, "w") as f:
        f.write("# one replayed step, %d launches, %.3f ms; columns: #, start us, duration us, gap us, %s, kernel\n" %
                (len(seg), (seg[-1][2] - t0) / 1e6, " ".join(want)))
        prev = t0
        for i, r in enumerate(seg):
            f.write("%4d %9.1f %8.1f %6.1f  %s  %s\n" % (i, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev) / 1e3,
                                                       " ".join("%7s" % str(v) for v in r[3:]), r[0][:110]))
            prev = r[2]


if __name__ == "__main__":
    main()

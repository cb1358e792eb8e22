#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database:  python tools/pmc_summary.py <db> [name filter]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    kn = re.sub(r'\(anonymous namespace\)::', '', str(r[ix.get('kernel_name', ix.get('name', 0))]))
    if flt and not re.search(flt, kn):
        continue
    acc[kn[:90]][r[ix['counter_name']]].append(float(r[ix['value']]))
for kn, d in acc.items():
    print(kn)
    for cn, v in sorted(d.items()):
        print("   %-28s n=%4d avg=%16.1f" % (cn, len(v), sum(v) / len(v)))

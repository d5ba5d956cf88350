#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (--pmc ... --kernel-trace run)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, i.name, e.value, d.id from {pmc} e join {info} i on e.pmc_id=i.id "
                  f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id")
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for k, c, v, did in rows:
    if pat in k:
        acc[k[:70]][c] += v
        cnt[k[:70]].add(did)
for k, cs in acc.items():
    n = len(cnt[k])
    print(k, f"(dispatches {n})")
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {v / n:16.1f}")

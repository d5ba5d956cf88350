#!/usr/bin/env python
"""HBM bytes of one C2 train step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py --no-graph.

  python tools/step_hbm_bytes.py <fetch.db> <write.db> <steps run (warm-up + timed + sustained)> > profiles/rNN_step_pmc.json

Only kernels dispatched a multiple of <steps> times are counted (the per-step kernels; the one-off gather roofline
launches, initialisation fills and the like drop out).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 (wide coalesced reads are tallied at half their bytes), so the total is an upper bound on the read side."""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(dbfile, counter):
    db = sqlite3.connect(dbfile)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, i.name, e.value, d.id from {pmc} e join {info} i on e.pmc_id=i.id "
                      f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id")
    tot, ids = defaultdict(float), defaultdict(set)
    for k, c, v, did in rows:
        if c == counter:
            tot[k] += v
            ids[k].add(did)
    return {k: (tot[k], len(ids[k])) for k in tot}


def main():
    fetch, write, steps = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    rd = wr = 0.0
    kernels = {}
    for k, (v, n) in fetch.items():
        if n % steps == 0 and n > 0:
            # counters are reported in KB by rocprofv3's derived metric (bytes / 1024)
            b = v * 1024.0 * 2.0 / steps
            rd += b
            kernels[k[:60]] = {"read": b}
    for k, (v, n) in write.items():
        if n % steps == 0 and n > 0:
            b = v * 1024.0 / steps
            wr += b
            kernels.setdefault(k[:60], {})["write"] = b
    top = sorted(kernels.items(), key=lambda kv: -(kv[1].get("read", 0) + kv[1].get("write", 0)))[:12]
    extra = {}
    if len(sys.argv) > 4:  # repo root: stamp the kernel sources the passes ran on (bench.py quotes the file only while
        import hashlib    # they are unchanged)
        import os
        sys.path.insert(0, sys.argv[4])
        from bench import STEP_SOURCES
        h = hashlib.sha256()
        for f in STEP_SOURCES:
            h.update(open(os.path.join(sys.argv[4], "deepof_amd", "csrc", f), "rb").read())
        extra = {"source_sha": h.hexdigest()[:16], "sources": STEP_SOURCES}
    print(json.dumps({**extra, "hbm_bytes_per_step": rd + wr, "read_bytes_x2": rd, "write_bytes": wr, "steps_profiled": steps,
                      "note": "FETCH_SIZE doubled (gfx950 correction, upper bound); per-step kernels only; eager launches",
                      "top_kernels": dict(top)}, indent=1))


if __name__ == "__main__":
    main()

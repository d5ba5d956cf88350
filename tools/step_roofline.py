#!/usr/bin/env python
"""Per-kernel roofline of one C2 train step from four rocprofv3 runs of bench.py --no-graph (eager launches):
a kernel trace (durations), FETCH_SIZE, WRITE_SIZE (separate PMC passes, MI355X_MICROARCH.md) and one SQ pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_WAIT_ANY).

  python tools/step_roofline.py <trace.db> <fetch.db> <write.db> <sq.db> <steps> > profiles/rNN_step_roofline.md

HBM: (2 x FETCH_SIZE + WRITE_SIZE) / duration against 8 TB/s (FETCH doubled = upper bound).  Matrix pipe:
SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs) -- the counter sums busy cycles over the SIMDs -- with the
clock taken as GRBM-free 2.4 GHz (an upper bound on the denominator, so the fraction is a lower bound)."""
import sqlite3
import sys
from collections import defaultdict

HBM, CLK, SIMDS = 8.0e12, 2.4e9, 1024


def tabs(db):
    t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    return {k: [x for x in t if x.startswith(k)][0] for k in ("rocpd_kernel_dispatch", "rocpd_info_kernel_symbol")} | \
           {k: ([x for x in t if x.startswith(k)] or [None])[0] for k in ("rocpd_pmc_event", "rocpd_info_pmc")}


def durations(f):
    db = sqlite3.connect(f)
    t = tabs(db)
    d = defaultdict(list)
    for k, a, b in db.execute(f"select s.kernel_name, d.start, d.end from {t['rocpd_kernel_dispatch']} d join "
                              f"{t['rocpd_info_kernel_symbol']} s on d.kernel_id=s.id"):
        d[k].append(b - a)
    return d


def counters(f):
    db = sqlite3.connect(f)
    t = tabs(db)
    tot, ids = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for k, c, v, did in db.execute(f"select s.kernel_name, i.name, e.value, d.id from {t['rocpd_pmc_event']} e join "
                                   f"{t['rocpd_info_pmc']} i on e.pmc_id=i.id join {t['rocpd_kernel_dispatch']} d on "
                                   f"e.event_id=d.event_id join {t['rocpd_info_kernel_symbol']} s on d.kernel_id=s.id"):
        tot[k][c] += v
        ids[k].add(did)
    return {k: {c: v / len(ids[k]) for c, v in cs.items()} for k, cs in tot.items()}, {k: len(v) for k, v in ids.items()}


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:44]


def main():
    dur, (fe, nf), (wr, _), (sq, _), steps = durations(sys.argv[1]), counters(sys.argv[2]), counters(sys.argv[3]), counters(sys.argv[4]), int(sys.argv[5])
    rows = []
    for k, ds in dur.items():
        if len(ds) % steps or not len(ds):
            continue
        per_step = len(ds) // steps
        us = sum(ds) / len(ds) / 1e3
        rb = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2
        wb = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        s = sq.get(k, {})
        mfma = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (us * 1e-6 * CLK * SIMDS) if us else 0.0
        wave = s.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((per_step * us, short(k), per_step, us, rb / 1e6, wb / 1e6, (rb + wb) / (us * 1e-6) / HBM if us else 0,
                     mfma, s.get("SQ_ACTIVE_INST_VALU", 0.0) / wave if wave else 0, s.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else 0,
                     s.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("| kernel | launches / step | µs / launch | MB read (x2) | MB written | HBM frac of 8 TB/s | MFMA busy frac | VALU active / wave-cycles | issue stall | waitcnt / barrier |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:24]:
        print(f"| `{r[1]}` | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]:.2f} | {r[7]:.2f} | {r[8]:.2f} | {r[9]:.2f} | {r[10]:.2f} |")
    print(f"\nper-step kernel time (eager launches, these kernels): {tot:.0f} µs; HBM bytes per step (2 x FETCH + WRITE): "
          f"{sum(r[2] * (r[4] + r[5]) for r in rows) / 1e3:.2f} GB")


if __name__ == "__main__":
    main()

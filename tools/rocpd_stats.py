#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"| kernel | calls | total_ms | avg_us | min_us | max_us | pct |")
    print("|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        n = r[0]
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| {n} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/total:.1f} |")
    print(f"\ntotal kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


def timeline(path, marker, which=-2):
    """Ordered dispatches of one step: from the `which`-th occurrence of a kernel whose name contains `marker` to the
    next one; per dispatch the duration and the idle gap to the previous dispatch's end."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[which], marks[which + 1]
    print(f"| # | kernel | dur_us | gap_us | start_us |\n|---|---|---|---|---|")
    busy = gaps = 0.0
    for i in range(a, b):
        n, s, e = rows[i]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = n if len(n) < 70 else n[:67] + "..."
        gap = (s - rows[i - 1][2]) / 1e3
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        print(f"| {i - a} | {n} | {(e - s) / 1e3:.2f} | {gap:.2f} | {(s - rows[a][1]) / 1e3:.1f} |")
    print(f"\nstep span {(rows[b][1] - rows[a][1]) / 1e3:.1f} us: {b - a} dispatches, busy {busy:.1f} us, idle {gaps:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--timeline":
        timeline(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else -2)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)

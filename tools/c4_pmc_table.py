#!/usr/bin/env python
"""tools/prof_c4_pmc.sh's two counter summaries -> the per-kernel HBM byte table of one C4 step (markdown).

    python tools/c4_pmc_table.py gpurun_out/<tag>/pmc_FETCH_SIZE.txt gpurun_out/<tag>/pmc_WRITE_SIZE.txt [steps_profiled=3]

The summaries hold per-dispatch averages in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md's gfx950 note: an upper bound)."""
import re
import sys


def parse(path, ctr):
    out, name, n = {}, None, 0
    for line in open(path):
        m = re.match(r"(\S+) \(dispatches (\d+)\)", line)
        if m:
            name, n = m.group(1), int(m.group(2))
        elif ctr in line and name:
            out[name] = (n, float(line.split()[-1]) * 1024.0)
    return out


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows, rd_tot, wr_tot = [], 0.0, 0.0
    for k in fetch:
        n, f = fetch[k]
        w = write.get(k, (n, 0.0))[1]
        per_step = n / steps
        rd, wr = 2.0 * f * per_step, w * per_step
        rd_tot += rd
        wr_tot += wr
        rows.append((rd + wr, k, per_step, 2.0 * f, w))
    print(f"TCN kernels of one C4 step: {rd_tot / 1e9:.1f} GB read (x2) + {wr_tot / 1e9:.1f} GB written = {(rd_tot + wr_tot) / 1e9:.1f} GB\n")
    print("| kernel | launches / step | read MB / launch (x2) | written MB / launch | GB / step |")
    print("|---|---|---|---|---|")
    for tot, k, per_step, f, w in sorted(rows, reverse=True):
        print(f"| `{k[:64]}` | {per_step:g} | {f / 1e6:.0f} | {w / 1e6:.0f} | {tot / 1e9:.1f} |")


if __name__ == "__main__":
    main()

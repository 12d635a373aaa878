"""Summarise rocprofv3 PMC passes into per-kernel HBM bytes per launch.

    python scripts/pmc_summary.py <dir-with-FETCH_SIZE-pass> <dir-with-WRITE_SIZE-pass>

Each directory is the -d output of `rocprofv3 --pmc FETCH_SIZE ...` / `--pmc WRITE_SIZE ...` (separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so it is doubled.  Averages are per launch.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] = int(r.get("Grid_Size", 0) or 0)
    return acc


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    fe = load(fd, "FETCH_SIZE")
    wr = load(wd, "WRITE_SIZE")
    print("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), average per launch")
    rows = []
    for k in set(fe) | set(wr):
        f = fe[k][0] / max(fe[k][1], 1) if k in fe else 0.0
        w = wr[k][0] / max(wr[k][1], 1) if k in wr else 0.0
        n = fe[k][1] if k in fe else wr[k][1]
        rows.append((2 * f + w, k, f, w, n))
    for tot, k, f, w, n in sorted(rows, reverse=True):
        print(f"  {k[:60]:60s} launches={n:5d} FETCH_SIZE={f:10.1f} KB (x2 = {2 * f / 1024:7.1f} MB)  "
              f"WRITE_SIZE={w:10.1f} KB ({w / 1024:7.1f} MB)  total={tot / 1024:7.1f} MB")


if __name__ == "__main__":
    main()

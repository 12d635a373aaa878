"""Summarise rocprofv3 PMC passes into per-kernel HBM bytes per launch.

    python scripts/pmc_summary.py <dir-with-FETCH_SIZE-pass> <dir-with-WRITE_SIZE-pass> [--json profiles/pmc_traffic.json
                                  --workload B2 --n-local 256 --profile profiles/rNN_pmc_hbm_traffic.txt]

With --json the per-launch bytes of the conv1 kernels are also written as the record bench.py reads for
`roofline.traffic`, together with the hash of the kernel sources they were measured on (bench.py reports null
when the sources have changed since).

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


# bench.py's names for the conv1 kernels <- substrings of the rocprof kernel names (template arguments included)
BENCH_KEYS = (("conv1_fwd", ("conv3x3_kernel<2,", "conv3x3_kernel<0,")), ("conv1_dgrad", ("conv3x3_kernel<3,", "conv3x3_kernel<1,")),
              ("conv1_wgrad", ("wgrad3b_pair_kernel", "wgrad3b_kernel<5", "wgrad3b_kernel<10", "wgrad3r_kernel<5", "wgrad3r_kernel<10", "wgrad3_kernel")))


def write_json(rows, path, workload, n_local, profile):
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cmlpl_amd.build_ext import source_hash
    rec = {}
    for key, pats in BENCH_KEYS:
        # the conv1 launch is the biggest-traffic kernel among the matching instantiations (conv2 uses the same
        # templates on the quarter-size map)
        cands = [(tot, k) for tot, k, f, w, n in rows if any(p in k.replace(" ", "") for p in pats)]
        if cands:
            tot, k = max(cands)
            rec[key] = tot * 1024.0
    json.dump({"source_hash": source_hash(), "workload": workload, "n_local": n_local, "profile": profile,
               "unit": "bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (rocprofv3 PMC, separate passes)",
               "bytes_per_launch": rec}, open(path, "w"), indent=1)
    print(f"wrote {path}: {rec}")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--json", default=None); ap.add_argument("--workload", default="B2")
    ap.add_argument("--n-local", type=int, default=256); ap.add_argument("--profile", default=None)
    a = ap.parse_args()
    fd, wd = a.fetch_dir, a.write_dir
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
    if a.json:
        write_json(rows, a.json, a.workload, a.n_local, a.profile)


if __name__ == "__main__":
    main()

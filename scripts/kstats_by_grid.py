"""One line per (kernel, grid) from a rocprofv3 kernel-trace csv: the --stats summary averages a kernel over every
size a script calls it at, which hides the per-size time.
    python3 scripts/kstats_by_grid.py TRACE.csv [name-substring ...]
"""
import collections
import csv
import sys


def main(path, keys):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "cmlpl" not in n or (keys and not any(k in n for k in keys)):
            continue
        name = n.replace("void ", "").replace("cmlpl::", "").split("(")[0]
        grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(int(r[f"Workgroup_Size_{a}"]), 1)) for a in "XY")
        agg[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid), v in sorted(agg.items()):
        v.sort()
        print(f"  {name:28s} workgroups {grid:>9s}  median {v[len(v) // 2]:8.2f} us  mean {sum(v) / len(v):8.2f} us  x{len(v)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

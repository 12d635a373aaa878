"""Per-kernel averages of every counter found in the given rocprofv3 --pmc output directories (one line per counter for
the kernels that take most of the step).     python3 scripts/pmc_instmix_summary.py DIR [DIR ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
waves = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cmlpl" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].replace("void ", "").replace("cmlpl::", "").split("(")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            waves[k] = int(r["Grid_Size"]) // 64
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", [0, 1])[0] / max(acc[k].get("SQ_BUSY_CYCLES", [0, 1])[1], 1)):
    avg = {c: v[0] / max(v[1], 1) for c, v in acc[k].items()}
    w = waves[k]
    print(f"{k}  ({w} waves per launch)")
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    for c in sorted(avg):
        extra = ""
        if c.startswith("SQ_INSTS_") or c == "SQ_WAVES":
            extra = f"   {avg[c] / w:10.1f} per wave"
        elif wc and (c.startswith("SQ_ACTIVE_INST") or c.startswith("SQ_WAIT")):
            extra = f"   {avg[c] / wc:10.3f} of wave-cycles"
        print(f"    {c:32s} {avg[c]:14.4g}{extra}")

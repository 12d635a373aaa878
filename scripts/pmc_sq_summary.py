"""Per-kernel averages of the SQ counters of rocprofv3 --pmc passes (one directory per pass).
    python scripts/pmc_sq_summary.py <dir> [<dir> ...]
SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles their matrix pipe was busy (checked: 64 per
v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32, times SQ_INSTS_MFMA); GRBM_GUI_ACTIVE is the sum over the 8 XCDs
of the cycles the dispatch was active (MI355X_MICROARCH.md), so
    MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
is the fraction of the chip's matrix-pipe cycles in use while the kernel ran.  SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* are per-wave quad-cycles, printed as fractions of the wave lifetime."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[r["Kernel_Name"]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    names = sorted({c for k in acc.values() for c in k})
    print("rocprofv3 --pmc (SQ counters), average per launch; counters:", ", ".join(names))
    rows = []
    for k, cs in acc.items():
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        rows.append((avg.get("SQ_BUSY_CYCLES", 0.0), k, avg, max(v[1] for v in cs.values())))
    for _, k, avg, n in sorted(rows, reverse=True):
        busy, mf = avg.get("SQ_BUSY_CYCLES", 0.0), avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        line = f"  {k[:64]:64s} n={n:4d}"
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            line += f"  MFMA utilisation {mf / (1024.0 * gui / 8.0):.3f}"
        if wc:
            line += (f"  wave-cycles: wait_any {avg.get('SQ_WAIT_ANY', 0) / wc:.2f} wait_inst {avg.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}"
                     f" active {avg.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
            if c in avg:
                line += f"  {c}={avg[c]:.3g}"
        print(line)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""rocprofv3 SQ counter passes (tools/capture_profiles.sh sq) -> profiles/<tag>_sq_counters_per_kernel.csv (one row per
kernel: average counter values per launch) and profiles/<tag>_sq_counters_summary.txt (one line per kernel family).

Counters (all summed over the chip by rocprofv3): SQ_INSTS_MFMA (MFMA instructions issued, per wave), SQ_VALU_MFMA_BUSY_CYCLES
(cycles a SIMD's matrix pipe is busy, summed over SIMDs; in quad-cycle units x4 -- see `mfma_cycles_per_inst`, which comes
out at 64 = the 16 passes x 4 cycles of v_mfma_f32_32x32x2_f32 when the unit is right), SQ_BUSY_CU_CYCLES (cycles a CU has
waves, summed over CUs), SQ_WAVE_CYCLES (wave-resident cycles summed over waves), SQ_WAVES, GRBM_GUI_ACTIVE (chip-active
cycles, summed over the 8 XCDs).  Derived per kernel:
    mfma_busy_of_cu_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)    matrix-pipe share of the time CUs hold waves
    mfma_busy_of_chip      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)  matrix-pipe share of the launch, whole chip
    waves_per_simd         = SQ_WAVE_CYCLES / (4 x SQ_BUSY_CU_CYCLES)                     resident waves per SIMD while busy
usage: sq_counters.py <dir with one sub-directory per pass> <out.csv> <summary.txt>"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_account import family  # noqa: E402


def main(src, out_csv, out_txt):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")) + glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in per.values() for c in v})
    derived = ["mfma_cycles_per_inst", "mfma_busy_of_cu_busy", "mfma_busy_of_chip", "waves_per_simd"]
    rows = [["kernel", "family", "launches"] + names + derived]
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, v in sorted(per.items()):
        if "ancsh::" not in k:
            continue
        m = {c: sum(x) / len(x) for c, x in v.items()}
        n = max(len(x) for x in v.values())
        g = lambda c: m.get(c, float("nan"))
        d = [g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA") if g("SQ_INSTS_MFMA") else float("nan"),
             g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") else float("nan"),
             g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8) if g("GRBM_GUI_ACTIVE") else float("nan"),
             g("SQ_WAVE_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") else float("nan")]
        rows.append([k[:110], family(k), n] + ["%.6g" % g(c) for c in names] + ["%.4f" % x for x in d])
        f = fam[family(k)]
        for c in names:
            f[c] += m.get(c, 0.0) * n
        f["_n"] += n
    csv.writer(open(out_csv, "w", newline="")).writerows(rows)
    with open(out_txt, "w") as o:
        o.write("# per kernel family, summed over the family's launches of the profiled command (tools/sq_counters.py)\n")
        o.write("%-36s %9s %16s %18s %18s %14s %14s %12s\n" % ("family", "launches", "SQ_INSTS_MFMA", "MFMA_BUSY_CYCLES", "BUSY_CU_CYCLES", "mfma/cu_busy", "mfma/chip", "waves/simd"))
        for k, f in sorted(fam.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
            cu, gui = f.get("SQ_BUSY_CU_CYCLES", 0), f.get("GRBM_GUI_ACTIVE", 0)
            o.write("%-36s %9d %16.4g %18.4g %18.4g %14s %14s %12s\n" % (
                k, f["_n"], f.get("SQ_INSTS_MFMA", 0), f.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), cu,
                "%.3f" % (f.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * cu)) if cu else "-",
                "%.3f" % (f.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui / 8)) if gui else "-",
                "%.2f" % (f.get("SQ_WAVE_CYCLES", 0) / (4 * cu)) if cu else "-"))
    print(open(out_txt).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])

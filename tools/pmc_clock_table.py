"""Per-kernel clock and matrix-pipe occupancy from one rocprofv3 pass with
`--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (counter arithmetic: tools/layer_table.py).

    python tools/pmc_clock_table.py <dir> [min_us]"""
import sys

from layer_table import load_counters, load_trace, short

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
tr = load_trace(d)
c = load_counters(d)
acc = {}
for k, r in enumerate(tr):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us < min_us or k >= len(c["GRBM_GUI_ACTIVE"]):
        continue
    gui = c["GRBM_GUI_ACTIVE"][k] / 8.0
    if gui <= 0:
        continue
    acc.setdefault(short(r["Kernel_Name"]), []).append((us, gui / us / 1e3, c["SQ_VALU_MFMA_BUSY_CYCLES"][k] / (gui * 1024.0)))
print("%-58s %6s %9s %10s %10s" % ("kernel ", "n", "us", "clock GHz", "MFMA busy"))
for n, a in sorted(acc.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    m = [sum(x[i] for x in a) / len(a) for i in range(3)]
    print("%-58s %6d %9.1f %10.3f %10.3f" % (n[:58], len(a), m[0], m[1], m[2]))

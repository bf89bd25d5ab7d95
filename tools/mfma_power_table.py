"""Clock and matrix-pipe occupancy of tools/mfma_power.hip's kernels from one rocprofv3 counter pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- ./tools/bin/mfma_power
    python tools/mfma_power_table.py <dir>

(same counter arithmetic as tools/layer_table.py: GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA_BUSY over all SIMDs)"""
import sys

from layer_table import load_counters, load_trace, short

d = sys.argv[1]
tr = load_trace(d)
c = load_counters(d)
acc = {}
for k, r in enumerate(tr):
    n = short(r["Kernel_Name"])
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us < 5000:      # the 1000-iteration warm-up launches
        continue
    gui = c["GRBM_GUI_ACTIVE"][k] / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][k] / (gui * 1024.0)
    a = acc.setdefault(n, [])
    a.append((us, gui / us / 1e3, busy))
print("%-26s %9s %10s %10s" % ("k<TYPE,RND,REFR,Z8,ORD,TRUNC>", "us", "clock GHz", "MFMA busy"))
for n, a in acc.items():
    a = a[len(a) // 3:]   # the settled launches
    m = [sum(x[i] for x in a) / len(a) for i in range(3)]
    print("%-26s %9.0f %10.3f %10.3f" % (n, m[0], m[1], m[2]))

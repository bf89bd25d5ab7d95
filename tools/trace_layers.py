"""Per-dispatch view of one bench step from a rocprofv3 --kernel-trace csv: which layer costs what.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline
    python tools/trace_layers.py gpurun_out/trace [n_steps_total]
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    conv = [i for i, r in enumerate(rows) if "f16x3_pc_kernel" in r["Kernel_Name"] or "f16x3_kernel<64, true" in r["Kernel_Name"]
            or "ELi64ELb1" in r["Kernel_Name"]]
    if len(conv) < 2:
        conv = [i for i, r in enumerate(rows) if "conv_first" in r["Kernel_Name"]]
    a, b = conv[-2], conv[-1]  # one full image: from a first fused conv to the next
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    tot = 0.0
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        name = r["Kernel_Name"].split("(")[0][-60:]
        print("%9.1f us  +%9.1f  grid %8s wg %4s  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                       r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                                                       r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), name))
    print("sum of kernel time %.1f us over %d dispatches; span %.1f us" % (
        tot, len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main()

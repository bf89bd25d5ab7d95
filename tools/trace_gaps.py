"""Where does a bench step's wall time go?  From a rocprofv3 --kernel-trace csv: the union of the intervals in
which a convolution kernel is resident, the time only small kernels run, and the time nothing runs, over the
steady-state span (first fused-pair kernel of the 3rd image .. last one).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 8 --warmup 2 --no-events --no-cpu-baseline --no-latency
    python tools/trace_gaps.py gpurun_out/trace
"""
import csv
import glob
import os
import sys


def union(iv):
    iv = sorted(iv)
    out, cur = 0, None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur:
                out += cur[1] - cur[0]
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur:
        out += cur[1] - cur[0]
    return out


def main():
    root = sys.argv[1]
    f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    is_conv = lambda r: "conv_mfma" in r["Kernel_Name"]
    is_pc = lambda r: "f16x3_pc_kernel" in r["Kernel_Name"]
    pcs = [i for i, r in enumerate(rows) if is_pc(r)]
    if len(pcs) < 5:
        print("not enough images in the trace")
        return
    a, b = pcs[2], pcs[-1]
    n_img = len(pcs) - 3
    seg = rows[a:b]
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    span = t1 - t0
    clip = lambda r: (max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1))
    conv_iv = [clip(r) for r in seg if is_conv(r)]
    all_iv = [clip(r) for r in seg]
    conv_busy, any_busy = union(conv_iv), union(all_iv)
    conv_sum = sum(b_ - a_ for a_, b_ in conv_iv)
    print("images %d, span %.3f ms/image" % (n_img, span / n_img / 1e6))
    print("  a conv kernel resident      %.3f ms/image (sum of conv kernel durations %.3f)" % (conv_busy / n_img / 1e6, conv_sum / n_img / 1e6))
    print("  only non-conv kernels       %.3f ms/image" % ((any_busy - conv_busy) / n_img / 1e6))
    print("  nothing resident            %.3f ms/image" % ((span - any_busy) / n_img / 1e6))
    # conv -> conv hand-over gaps inside the span
    cs = sorted(conv_iv)
    gaps = [(cs[i + 1][0] - max(c[1] for c in cs[:i + 1][-3:])) for i in range(len(cs) - 1)]
    gaps = [g for g in gaps if g > 0]
    print("  conv->conv gaps: n %d, mean %.1f us, total %.3f ms/image, max %.1f us" % (
        len(gaps), sum(gaps) / max(len(gaps), 1) / 1e3, sum(gaps) / n_img / 1e6, max(gaps) / 1e3 if gaps else 0))
    big = sorted(((cs[i + 1][0] - cs[i][1], i) for i in range(len(cs) - 1)), reverse=True)[:8]
    for g, i in big:
        print("    gap %.1f us at +%.3f ms" % (g / 1e3, (cs[i][1] - t0) / 1e6))


if __name__ == "__main__":
    main()

"""Hash of the kernel sources: ties a committed PMC summary (profiles/*.json) to the code it was measured on.
Comments and white space do not count (a reworded comment does not invalidate a counter run), nor does what only timing
builds compile (`#ifdef SHF_CONV_TIMING` blocks: the shipped library is built without the macro)."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["conv.hip", "conv_f16x3.hip", "conv_f16x3_types.h", "conv_f16x3_8w.h", "conv_f16x3_w4d.h", "conv_f16x3_pc.h", "conv_f16x3_k1.h", "conv_f16x3_h3.h", "conv_common.h",
         "misc.hip", "tail.hip", "merge.hip", "pre.hip", "shf_internal.h"]


def strip_timing_blocks(src):
    """Drop the lines of `#ifdef SHF_CONV_TIMING` branches (keep their `#else` branches): not part of the shipped build."""
    out, stack = [], []          # stack entries: [is_timing_block, in_else]
    for line in src.split("\n"):
        t = line.strip()
        if t.startswith("#if"):
            stack.append([t.replace(" ", "") == "#ifdefSHF_CONV_TIMING", False])
            if stack[-1][0]:
                continue
        elif t.startswith("#else") and stack and stack[-1][0]:
            stack[-1][1] = True
            continue
        elif t.startswith("#endif") and stack:
            was_timing = stack.pop()[0]
            if was_timing:
                continue
        if any(tb and not in_else for tb, in_else in stack):
            continue
        out.append(line)
    return "\n".join(out)


def kernel_source_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, "smallhardface_amd", "csrc", f), "rb") as fh:
            src = fh.read().decode("utf-8", "replace")
        src = strip_timing_blocks(src)
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"\s+", " ", src)
        h.update(f.encode() + b"\0" + src.encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())

"""Build recipe for libshf_hip.so (gfx950 only, in-tree so it travels with gpurun).

    python -m smallhardface_amd.build          # incremental
    python -m smallhardface_amd.build --force

hipcc cross-compiles without a GPU.  The box-arithmetic kernels (tail.hip, merge.hip)
and the image resize (pre.hip) are built with -ffp-contract=off so IoU / decode round like numpy and devIoU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libshf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = [
    ("conv.hip", []),
    ("conv_f16x3.hip", []),
    ("misc.hip", []),
    ("tail.hip", ["-ffp-contract=off"]),
    ("merge.hip", ["-ffp-contract=off"]),
    ("pre.hip", ["-ffp-contract=off"]),
    ("calib.hip", []),
    ("net_graph.cpp", []),
    ("net_forward.cpp", []),
    ("net_detect.cpp", []),
    ("net_api.cpp", []),
]
HEADERS = ["shf_internal.h", "conv_common.h", "conv_f16x3_types.h", "conv_f16x3_8w.h", "conv_f16x3_w4d.h", "conv_f16x3_pc.h", "conv_f16x3_k1.h", "conv_f16x3_h3.h",
           "proto_text.h", "net_internal.h", os.path.join("..", "..", "include", "shf_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o] + extra
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

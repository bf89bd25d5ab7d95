"""ctypes binding of libshf_hip.so (include/shf_hip.h).

There is NO fallback: if the HIP library is missing or no GPU is visible the
product path raises.  ``load(require_gpu=False)`` exists so CPU-only tests can
check that the library loads and exports every declared symbol.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHF_LIB") or os.path.join(_HERE, "libshf_hip.so")  # SHF_LIB: kernel-variant experiments
HEADER_PATH = os.path.join(_HERE, "..", "include", "shf_hip.h")

_lib = None


class ShfError(RuntimeError):
    pass


def declared_symbols():
    """Every function name declared in include/shf_hip.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(shf_[a-z0-9_]+)\s*\(", txt)))


def load(require_gpu=True):
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ShfError("libshf_hip.so is not built (%s): run `python -m smallhardface_amd.build`; "
                           "there is no CPU fallback" % LIB_PATH)
        try:
            # torch bundles its own libamdhip64 (same SONAME): import it first so the
            # process ends up with ONE HIP runtime when torch is used for RCCL plumbing
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _declare(lib)
        _lib = lib
    if require_gpu and _lib.shf_device_count() <= 0:
        raise ShfError("no MI355X/HIP device visible: the smallhardface_amd runtime has no CPU fallback")
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise ShfError("%s%s" % (what + ": " if what else "", last_error()))


def last_error():
    return _lib.shf_last_error().decode("utf-8", "replace") if _lib is not None else ""


def _declare(lib):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)
    sig = {
        "shf_set_mode_gpu": (ci, []),
        "shf_set_device": (ci, [ci]),
        "shf_device_count": (ci, []),
        "shf_last_error": (C.c_char_p, []),
        "shf_version": (C.c_char_p, []),
        "shf_net_create": (vp, [C.c_char_p, C.c_char_p, C.c_char_p, ci]),
        "shf_net_clone": (vp, [vp]),
        "shf_net_destroy": (None, [vp]),
        "shf_net_num_blobs": (ci, [vp]),
        "shf_net_blob_name": (C.c_char_p, [vp, ci]),
        "shf_net_num_inputs": (ci, [vp]),
        "shf_net_input_blob": (ci, [vp, ci]),
        "shf_net_num_outputs": (ci, [vp]),
        "shf_net_output_blob": (ci, [vp, ci]),
        "shf_net_num_layers": (ci, [vp]),
        "shf_net_layer_name": (C.c_char_p, [vp, ci]),
        "shf_net_layer_type": (C.c_char_p, [vp, ci]),
        "shf_net_layer_num_params": (ci, [vp, ci]),
        "shf_net_param_shape": (ci, [vp, ci, ci, ip]),
        "shf_net_param_data": (fp, [vp, ci, ci]),
        "shf_net_param_commit": (ci, [vp, ci]),
        "shf_blob_reshape": (ci, [vp, ci, ip, ci]),
        "shf_blob_shape": (ci, [vp, ci, ip]),
        "shf_blob_mutable_host_data": (fp, [vp, ci]),
        "shf_net_forward": (ci, [vp]),
        "shf_net_set_proposal_cfg": (ci, [vp, ci, cf, cf]),
        "shf_net_set_conv_mode": (ci, [vp, ci]),
        "shf_net_get_conv_mode": (ci, [vp]),
        "shf_net_range_fallbacks": (C.c_longlong, [vp]),
        "shf_alloc_counts": (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "shf_device_pci_bus_id": (ci, [C.c_char_p, ci]),
        "shf_image_blobs": (ci, [vp, ci, ci, ci, dp, dp, C.POINTER(vp), ip, ip]),
        "shf_net_set_layer_products": (ci, [vp, C.c_char_p, ci]),
        "shf_net_record_event": (ci, [vp]),
        "shf_net_wait_event": (ci, [vp, vp]),
        "shf_detect_begin": (ci, [vp]),
        "shf_detect_add_level": (ci, [vp, vp, ci, ci, ci, ci, ci, cf, ci, cf]),
        "shf_detect_add_levels": (ci, [vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ip, ip, ip, ip, fp, ip, cf, ci]),
        "shf_pyramid_level_shape": (ci, [ci, ci, C.c_double, ci, ip, ip, ip, ip]),
        "shf_make_pyramid_level": (ci, [vp, vp, ci, ci, C.c_double, ci, dp, vp, ci, ci, ci, ci]),
        "shf_net_set_predecessor": (ci, [vp, vp]),
        "shf_net_set_pipeline": (ci, [vp, ci]),
        "shf_detect_finish": (ci, [vp, ci, cf, dp, ci, ip]),
        "shf_detect_count": (ci, [vp]),
        "shf_detect_export": (ci, [vp, vp, ci, ip]),
        "shf_detect_import": (ci, [vp, vp, ci]),
        "shf_detect_export_many": (ci, [vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ip]),
        "shf_debug_merge": (ci, [vp, ci, cf, ci, vp, vp, vp, ip, vp, vp]),
        "shf_debug_proposal": (ci, [vp, fp, fp, ci, ci, fp, fp, fp, ci, ip, ip]),
        "shf_debug_append": (ci, [vp, fp, fp, ci, ci, cf, ci, cf]),
        "shf_caffemodel_read_blob": (ci, [C.c_char_p, C.c_char_p, ci, fp, ci, ip, ip]),
        "shf_nms": (ci, [fp, ci, cf, ci, C.POINTER(C.c_int32), ip]),
        "shf_bbox_vote": (ci, [fp, ci, cf, dp, ci, ip]),
        "shf_generate_anchors": (ci, [ci, dp, ci, dp, ci, dp, ci, dp, dp, ci]),
        "shf_prof_enable": (ci, [vp, ci]),
        "shf_prof_only": (ci, [vp, ci]),
        "shf_prof_num_classes": (ci, [vp]),
        "shf_prof_class_name": (C.c_char_p, [vp, ci]),
        "shf_prof_read": (ci, [vp, ci, C.POINTER(C.c_int64), dp, dp, dp]),
        "shf_prof_reset": (ci, [vp]),
        "shf_calib_matrix_pipe": (ci, [ci, ci, ci, ci, ci, dp]),
        "shf_net_sync": (ci, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args

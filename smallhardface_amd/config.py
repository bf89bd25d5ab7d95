"""Configuration system of the detection driver.

Mirrors /root/reference/lib/utils/get_config.py: a default schema loaded into an
attribute-dict (``cfg``), an experiment TOML merged over it with key-existence
and type checks (``cfg_from_file``, get_config.py:94-137) and ``--amend K V``
pairs parsed with ``literal_eval`` (``cfg_from_list``, get_config.py:140-158).

The default values restate the reference's ``configs/default.toml`` so that an
existing experiment file (e.g. ``configs/smallhardface.toml``, which also sets
TRAIN.* keys) merges without a KeyError.  Only the TEST / MODEL / top-level keys
feed the inference hot path; TRAIN / TENSORBOARD / MISC are schema-only here.
"""
import copy
import os
import os.path as osp
from ast import literal_eval
from collections import OrderedDict

import numpy as np

try:  # py3.11+
    import tomllib as _toml
except ImportError:  # this image: py3.10 + tomli
    import tomli as _toml


class EasyDict(dict):
    """Attribute-access dict (stand-in for the ``easydict`` package)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return EasyDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


_DEFAULTS = {
    "DATA_DIR": "/mnt/WIDER_FACE", "EPS": 1e-14, "EXP_DIR": "face",
    "MAX_RESOLUTION": 16, "NAME": "face",
    "PIXEL_MEANS": [[[102.9801, 115.9465, 122.7717]]],
    "RNG_SEED": 3, "USE_GPU_NMS": True, "DEBUG": False, "PDB": False,
    "MISC": {"MIMIC_EVAL_BUG": True, "ACCURACY_THRESHOLD": 0.9},
    "TENSORBOARD": {"ENABLE": False, "HOSTNAME": "example.com", "PORT": 8889},
    "MODEL": {"DIAGNOSE": "", "DIFFERENT_DILATION": {"ENABLE": False},
              "HACK": {"TRAIN": "", "TEST": ""}},
    # schema only (training is out of scope for this build)
    "TRAIN": {
        "ANCHOR_MIN_SIZE": 4, "ANCHOR_N_POST_NMS": 300, "ANCHOR_N_PRE_NMS": 1000,
        "ANCHOR_NEGATIVE_OVERLAP": 0.3, "ANCHOR_POSITIVE_OVERLAP": 0.5,
        "ANCHOR_REGRESSION_OVERLAP": 0.3, "ASPECT_GROUPING": True,
        "BBOX_INSIDE_WEIGHTS": [1, 1, 1, 1], "BG_THRESH_HI": 0.5, "BG_THRESH_LOW": 0,
        "DB": "wider_train", "IMS_PER_BATCH": 1, "ITERS": 60000, "ITERSIZE": 2,
        "LR_POLICY": "STEP", "ORIG_SIZE": False, "POSITIVE_MINING": True,
        "PRETRAINED": "/mnt/WIDER_FACE/imagenet_models/VGG16.caffemodel",
        "PROTOTXT": "models/train_template.prototxt", "SNAPSHOT": 1000,
        "SNAPSHOT_INFIX": "", "SOLVER": "models/solver_template.prototxt",
        "STEPSIZE": 46000, "STEPVALUE": [21000, 42000], "WEIGHT_DECAY": 0.00025,
        "USE_FLIPPED": True, "GPU_ID": [0, 1, 2, 3],
        "LR": {"BASELR": 0.004, "BACKBONE_MULT": 2.0, "HEAD_MULT": 1.0},
        "SCALES": {"MODE": "SHORT_SIDE", "SHORT_SIDE": [400, 800, 1200], "MAX_SIZE": 2000},
        "AUGMENT": {"ENABLE": True,
                    "BRIGHTNESS": {"PROB": 0.5, "DELTA": 32.0},
                    "CONTRAST": {"PROB": 0.5, "LOWER": 0.5, "UPPER": 1.5},
                    "SATURATION": {"PROB": 0.5, "LOWER": 0.5, "UPPER": 1.5},
                    "HUE": {"PROB": 0.5, "DELTA": 18.0},
                    "CROP": {"PROB": 0.5, "LOWER": 0.6, "UPPER": 1.0, "POSITIVE_ENFORCE": True,
                             "MAX_TRIES": 50, "KEEP_ONLY_CENTER_INSIDE": True}},
        "DISABLE_EASY_IMAGE": {"ENABLE": False, "THRESHOLD": 1.0, "PROB": 0.5, "SMOOTH": False},
        "ANCHOR_SAMPLING": {"ANCHORS_PER_BATCH": 256, "ANCHOR_FG_FRACTION": 0.25,
                            "ANCHOR_NUM_METHOD": "fixed_num", "BATCH_POS_NEG_RATIO": 0.33},
    },
    # the inference hot path reads these (reference configs/default.toml:118-141)
    "TEST": {
        "ANCHOR_MIN_SIZE": 0, "ANCHOR_N_POST_NMS": -1, "DB": "wider_val", "FLIP": True,
        "LEVEL": [], "MAX_SIZE": 2000, "MODEL": "", "NO_CACHE": False, "NMS_THRESH": 0.4,
        "NMS_METHOD": "BBOX_VOTE", "N_DETS_PER_MODULE": 10000, "ORIG_SIZE": False,
        "PYRAMID_BASE_SIZE": [800, 1200], "PROTOTXT": "models/test_template.prototxt",
        "SCALES": [100, 300, 600, 1000, 1400], "SCORE_THRESH": 0.002,
        "GPU_ID": [0, 1, 2, 3], "IOU_THRESH": 0.5,
        "DEMO": {"ENABLE": False, "IMAGE": "demo/demo.jpg"},
    },
}


def _sort_dict(d):
    res = OrderedDict(sorted(d.items()))
    for k in res:
        if isinstance(res[k], dict):
            res[k] = _sort_dict(res[k])
    return res


def _fresh():
    c = EasyDict(_sort_dict(copy.deepcopy(_DEFAULTS)))
    c["LOG"] = EasyDict()
    c.ROOT_DIR = osp.abspath(osp.join(osp.dirname(__file__), '..'))
    c.DATA_DIR = osp.join(c.ROOT_DIR, c.DATA_DIR)
    c.DEBUG = os.environ.get('DEBUG') == '1'
    return c


cfg = _fresh()


def cfg_reset():
    """Restore ``cfg`` (in place) to the defaults — for tests."""
    cfg.clear()
    cfg.update(_fresh())


def _toml_value(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_toml_value(x) for x in v) + "]"
    return '"' + str(v) + '"'


def cfg_dumps(c):
    """The configuration as TOML text: get_config.py:76-77 (`toml.dump(_sort_dict(cfg), file)`).  The `toml` package is
    not in this image, so the writer is ours: sorted keys, a table's scalars before its sub-tables, `[A.B]` headers --
    what `toml` 0.10 emits for nested dicts of scalars and lists; `tomli` reads it back to the same dictionary."""
    out = []

    def emit(d, prefix):
        d = _sort_dict(d)
        for k, v in d.items():
            if not isinstance(v, dict):
                out.append("%s = %s" % (k, _toml_value(v)))
        for k, v in d.items():
            if isinstance(v, dict):
                out.append("")
                out.append("[%s]" % (prefix + k))
                emit(v, prefix + k + ".")

    emit(dict(c), "")
    return "\n".join(out).lstrip("\n") + "\n"


def cfg_dump(c, file):
    """get_config.py:76-77."""
    file.write(cfg_dumps(c))


def get_output_dir(imdb_name, net_name=None, output_dir='output', idx=-1):
    """get_config.py:47-66."""
    outdir = osp.abspath(osp.join(cfg.ROOT_DIR, output_dir, cfg.EXP_DIR, imdb_name))
    if net_name is not None:
        outdir = osp.join(outdir, net_name)
    if idx >= 0:
        outdir = osp.join(outdir, str(idx))
    os.makedirs(outdir, exist_ok=True)
    return outdir


def _merge_a_into_b(a, b):
    """get_config.py:94-131 (same KeyError / ValueError behaviour)."""
    if type(a) is not EasyDict:
        return
    for k, v in a.items():
        if k == "LOG":
            continue
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            elif isinstance(b[k], str) and isinstance(v, str):
                pass
            else:
                raise ValueError(('Type mismatch ({} vs. {}) '
                                  'for config key: {}').format(type(b[k]), type(v), k))
        if type(v) is EasyDict:
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a TOML experiment file and merge it into the defaults."""
    with open(filename, 'rb') as f:
        amend_config = EasyDict(_toml.load(f))
    _merge_a_into_b(amend_config, cfg)


def cfg_from_list(cfg_list):
    """Set config keys via a flat [K, V, K, V, ...] list (``--amend``)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = cfg
        for subkey in key_list[:-1]:
            assert subkey in d
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d, 'Please put {} in default.toml'.format(subkey)
        try:
            value = literal_eval(v)
        except Exception:
            value = v  # a plain string literal
        d[subkey] = value

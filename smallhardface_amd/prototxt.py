"""Caffe prototxt (protobuf text format) handling for the inference graph.

Replaces, for the inference path only, what the reference does with
``caffe_pb2`` + ``google.protobuf.text_format`` in
/root/reference/lib/prototxt/manipulate.py:

  * ``parse`` / ``dumps``      -- a small text-format reader/writer (no protoc in
    this image, so no generated ``caffe_pb2``);
  * ``manipulate_test``        -- manipulate.py:63-86: pick the template, insert the
    ``conv4_fuse_final_dim_red`` 512->128 3x3 conv + ReLU in front of the heads
    (``_add_dimension_reduction``, manipulate.py:166-188), write ``test.prototxt``;
  * ``build_test_template``    -- emits the detection graph the reference ships as
    models/test_different_dilation_template.prototxt / models/test_template.prototxt
    (VGG-16 trunk :17-367, unified head :369-478, heads + proposal :479-697), so the
    repo can run without the reference tree.  ``tests/test_prototxt.py`` checks it
    is structurally identical to the reference files when they are present.
"""
import re

from .config import cfg


class Enum(str):
    """A bare identifier value (``MAX``, ``true``) as opposed to a quoted string."""


class Msg(object):
    """Ordered multi-map: a protobuf message in text form."""

    def __init__(self, fields=None):
        self.fields = list(fields or [])  # [(name, value)]

    # -- access -----------------------------------------------------------
    def getall(self, name):
        return [v for k, v in self.fields if k == name]

    def get(self, name, default=None):
        for k, v in self.fields:
            if k == name:
                return v
        return default

    def has(self, name):
        return any(k == name for k, _ in self.fields)

    def add(self, name, value):
        self.fields.append((name, value))
        return value

    def set(self, name, value):
        for i, (k, _) in enumerate(self.fields):
            if k == name:
                self.fields[i] = (name, value)
                return
        self.fields.append((name, value))

    def set_nth(self, name, n, value):
        c = 0
        for i, (k, _) in enumerate(self.fields):
            if k == name:
                if c == n:
                    self.fields[i] = (name, value)
                    return
                c += 1
        raise IndexError(name)

    def clear(self, name):
        self.fields = [(k, v) for k, v in self.fields if k != name]

    def copy(self):
        return Msg([(k, v.copy() if isinstance(v, Msg) else v) for k, v in self.fields])

    def __eq__(self, other):
        return isinstance(other, Msg) and self.fields == other.fields

    def __repr__(self):
        return "Msg(%r)" % (self.fields,)


_TOKEN = re.compile(r"""
    \s+ | \#[^\n]* |
    (?P<str>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*') |
    (?P<punct>[{}:\[\],;<>]) |
    (?P<atom>[^\s{}:\[\],;<>"'#]+)
""", re.X)


def _tokens(text):
    pos = 0
    out = []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError("prototxt: bad character at offset %d: %r" % (pos, text[pos:pos + 20]))
        pos = m.end()
        if m.group("str") is not None:
            s = m.group("str")
            body = s[1:-1]
            body = re.sub(r"\\(.)", lambda mm: {"n": "\n", "t": "\t"}.get(mm.group(1), mm.group(1)), body)
            out.append(("str", body))
        elif m.group("punct") is not None:
            out.append(("punct", m.group("punct")))
        elif m.group("atom") is not None:
            out.append(("atom", m.group("atom")))
    return out


def _scalar(kind, tok):
    if kind == "str":
        return tok
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return Enum(tok)


def parse(text):
    """Parse protobuf text format into a ``Msg`` tree."""
    toks = _tokens(text)
    i = 0

    def message(close):
        nonlocal i
        msg = Msg()
        while i < len(toks):
            kind, tok = toks[i]
            if kind == "punct" and tok in (close or ""):
                i += 1
                return msg
            if kind == "punct" and tok in ",;":
                i += 1
                continue
            if kind != "atom":
                raise ValueError("prototxt: expected field name, got %r" % (tok,))
            name = tok
            i += 1
            kind, tok = toks[i]
            if kind == "punct" and tok == ":":
                i += 1
                kind, tok = toks[i]
            if kind == "punct" and tok in "{<":
                i += 1
                msg.add(name, message("}" if tok == "{" else ">"))
            elif kind == "punct" and tok == "[":
                i += 1
                while True:
                    kind, tok = toks[i]
                    i += 1
                    if kind == "punct" and tok == "]":
                        break
                    if kind == "punct" and tok == ",":
                        continue
                    msg.add(name, _scalar(kind, tok))
            else:
                i += 1
                val = _scalar(kind, tok)
                # adjacent string literals concatenate
                while kind == "str" and i < len(toks) and toks[i][0] == "str":
                    val += toks[i][1]
                    i += 1
                msg.add(name, val)
        if close:
            raise ValueError("prototxt: unterminated message")
        return msg

    return message(None)


def _fmt_scalar(v):
    if isinstance(v, Enum):
        return str(v)
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        return '"%s"' % v.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")
    if isinstance(v, float):
        return repr(v) if v != int(v) else ("%.1f" % v if abs(v) < 1e16 else repr(v))
    return str(v)


def dumps(msg, indent=0):
    """Serialise a ``Msg`` tree the way ``str(pb_message)`` lays text format out."""
    pad = "  " * indent
    out = []
    for k, v in msg.fields:
        if isinstance(v, Msg):
            out.append("%s%s {\n%s%s}\n" % (pad, k, dumps(v, indent + 1), pad))
        else:
            out.append("%s%s: %s\n" % (pad, k, _fmt_scalar(v)))
    return "".join(out)


# --------------------------------------------------------------------------
# manipulate.py equivalents
# --------------------------------------------------------------------------
def _simple_conv_layer(name, bottom, top, num_output, kernel_size, pad, dilation=1,
                       std=0.01, bias=0.0, param_type=0):
    """manipulate.py:89-146 (same field order as the reference builds them)."""
    L = Msg()
    L.add("name", name)
    L.add("type", "Convolution")
    L.add("bottom", bottom)
    L.add("top", top)
    mults = {0: None, 1: ((1.0, 0.0), (2.0, 0.0)), 2: ((1.0, 1.0), (2.0, 0.0)),
             3: ((10.0, 1.0), (20.0, 0.0)), 4: ((1.0, 1.0), (2.0, 1.0))}[param_type]
    for j in range(2):
        p = Msg()
        if mults is not None:
            p.add("lr_mult", mults[j][0])
            p.add("decay_mult", mults[j][1])
        L.add("param", p)
    cp = Msg()
    cp.add("num_output", num_output)
    cp.add("pad", pad)
    cp.add("kernel_size", kernel_size)
    wf = Msg([("type", "gaussian"), ("std", std)])
    cp.add("weight_filler", wf)
    bf = Msg([("type", "constant"), ("value", bias)])
    cp.add("bias_filler", bf)
    cp.add("dilation", dilation)
    L.add("convolution_param", cp)
    return L


def _simple_relu_layer(name, bottom, top=None):
    """manipulate.py:149-155."""
    return Msg([("name", name), ("type", "ReLU"), ("bottom", bottom),
                ("top", top if top is not None else bottom)])


def _add_dimension_reduction(pb):
    """manipulate.py:166-188: 512->128 3x3 reduction in front of the first ``head*`` layer."""
    if not cfg.MODEL.DIFFERENT_DILATION.ENABLE:
        return pb
    layers = pb.getall("layer")
    split = min(i for i, x in enumerate(layers) if str(x.get("name", "")).startswith("head"))
    assert layers[split - 2].get("name") == "conv4_fuse_final"
    layers[split - 2].set_nth("top", 0, layers[split - 2].get("top") + "_tmp")
    layers[split - 1].set_nth("bottom", 0, layers[split - 1].get("bottom") + "_tmp")
    layers[split - 1].set_nth("top", 0, layers[split - 1].get("top") + "_tmp")
    new_layers = layers[:split] + [
        _simple_conv_layer("conv4_fuse_final_dim_red", "conv4_fuse_final_tmp",
                           "conv4_fuse_final", 128, 3, 1, param_type=4),
        _simple_relu_layer("conv4_fuse_final_dim_red_relu", "conv4_fuse_final"),
    ] + layers[split:]
    # ClearField('layer') + extend(): the layers end up after every other field
    pb.clear("layer")
    for L in new_layers:
        pb.add("layer", L)
    return pb


def manipulate_test(ori, target_test, **kwargs):
    """manipulate.py:63-86 without the cosmetic ``draw_net_to_file`` / tensorboard image.

    ``ori`` may be a path or ``None``/missing file, in which case the built-in
    template generator supplies the graph (the reference forces the template path
    when DIFFERENT_DILATION is enabled, manipulate.py:65-66).
    """
    import os
    if cfg.MODEL.DIFFERENT_DILATION.ENABLE:
        ori = 'models/test_different_dilation_template.prototxt'
    if ori and os.path.isfile(ori):
        with open(ori, 'r') as f:
            test_pb = parse(f.read())
    else:
        test_pb = build_test_template(bool(cfg.MODEL.DIFFERENT_DILATION.ENABLE))
    test_pb = _add_dimension_reduction(test_pb)
    with open(target_test, 'w') as f:
        f.write(dumps(test_pb))
    return None


# --------------------------------------------------------------------------
# the detection graph
# --------------------------------------------------------------------------
def _frozen(n):
    return [Msg([("lr_mult", 0), ("decay_mult", 0)]) for _ in range(n)]


def _conv(name, bottom, top, nout, k, pad, params, fillers=True, extra=None, flat=False):
    L = Msg([("name", name), ("type", "Convolution"), ("bottom", bottom), ("top", top)])
    for p in params:
        L.add("param", p)
    cp = Msg()
    cp.add("num_output", nout)
    if flat:  # the heads write kernel_size first, then pad/stride
        cp.add("kernel_size", k)
        cp.add("pad", pad)
        cp.add("stride", 1)
    else:
        cp.add("pad", pad)
        cp.add("kernel_size", k)
    for kk, vv in (extra or []):
        cp.add(kk, vv)
    if fillers:
        cp.add("weight_filler", Msg([("type", "gaussian"), ("std", 0.01)]))
        cp.add("bias_filler", Msg([("type", "constant"), ("value", 0)]))
    L.add("convolution_param", cp)
    return L


def _relu(name, blob):
    return Msg([("name", name), ("type", "ReLU"), ("bottom", blob), ("top", blob)])


def build_test_template(different_dilation=True):
    """The TEST-phase graph of the detector as a ``Msg`` (NetParameter)."""
    net = Msg()
    net.add("name", "face")
    net.add("input", "data")
    net.add("input_shape", Msg([("dim", 1), ("dim", 3), ("dim", 224), ("dim", 224)]))
    net.add("input", "im_info")
    net.add("input_shape", Msg([("dim", 1), ("dim", 3)]))
    # VGG-16 trunk
    prev = "data"
    cfg_vgg = [(1, 64, 2), (2, 128, 2), (3, 256, 3), (4, 512, 3), (5, 512, 3)]
    for stage, ch, n in cfg_vgg:
        for j in range(1, n + 1):
            name = "conv%d_%d" % (stage, j)
            pr = _frozen(2) if stage <= 2 else [Msg([("lr_mult", 1)]), Msg([("lr_mult", 2)])]
            net.add("layer", _conv(name, prev, name, ch, 3, 1, pr, fillers=False))
            net.add("layer", _relu("relu%d_%d" % (stage, j), name))
            prev = name
        if stage < 5:
            pool = "pool%d" % stage
            net.add("layer", Msg([("name", pool), ("type", "Pooling"), ("bottom", prev), ("top", pool),
                                  ("pooling_param", Msg([("pool", Enum("MAX")), ("kernel_size", 2),
                                                         ("stride", 2)]))]))
            prev = pool
    lr12 = lambda: [Msg([("lr_mult", 1)]), Msg([("lr_mult", 2)])]
    # unified head
    net.add("layer", _conv("conv5_256", "conv5_3", "conv5_256", 256, 1, 0, lr12()))
    net.add("layer", _relu("conv5_256_relu", "conv5_256"))
    net.add("layer", Msg([
        ("name", "conv5_256_up"), ("type", "Deconvolution"), ("bottom", "conv5_256"),
        ("top", "conv5_256_up"),
        ("convolution_param", Msg([("kernel_size", 4), ("stride", 2), ("num_output", 256),
                                   ("group", 256), ("pad", 1),
                                   ("weight_filler", Msg([("type", "bilinear")])),
                                   ("bias_term", Enum("false"))])),
        ("param", Msg([("lr_mult", 0), ("decay_mult", 0)]))]))
    net.add("layer", _conv("conv4_256", "conv4_3", "conv4_256", 256, 1, 0, lr12()))
    net.add("layer", _relu("conv4_256_relu", "conv4_256"))
    net.add("layer", Msg([("name", "conv4_fuse"), ("type", "Concat"), ("bottom", "conv5_256_up"),
                          ("bottom", "conv4_256"), ("top", "conv4_fuse"),
                          ("concat_param", Msg([("axis", 1)]))]))
    net.add("layer", _conv("conv4_fuse_final", "conv4_fuse", "conv4_fuse_final", 512, 3, 1, lr12()))
    net.add("layer", _relu("conv4_fuse_final_relu", "conv4_fuse_final"))

    wd = lambda: [Msg([("lr_mult", 1.0), ("decay_mult", 1.0)]), Msg([("lr_mult", 2.0), ("decay_mult", 0)])]
    if different_dilation:
        for d in (1, 2, 4):
            hp = [Msg([("name", "head_w"), ("lr_mult", 1.0), ("decay_mult", 1.0)]),
                  Msg([("name", "head_b"), ("lr_mult", 2.0), ("decay_mult", 0)])]
            L = Msg([("name", "head_%d" % d), ("type", "Convolution"), ("bottom", "conv4_fuse_final"),
                     ("top", "head_%d" % d)])
            for p in hp:
                L.add("param", p)
            L.add("convolution_param", Msg([
                ("num_output", 128), ("kernel_size", 3), ("pad", d), ("stride", 1), ("dilation", d),
                ("weight_filler", Msg([("type", "gaussian"), ("std", 0.01)])),
                ("bias_filler", Msg([("type", "constant"), ("value", 0)]))]))
            net.add("layer", L)
            net.add("layer", _relu("head_%d_relu" % d, "head_%d" % d))
        for d in (1, 2, 4):
            net.add("layer", _conv("cls_score_%d" % d, "head_%d" % d, "cls_score_%d_output" % d,
                                   2, 1, 0, wd(), flat=True))
            net.add("layer", _conv("bbox_pred_%d" % d, "head_%d" % d, "bbox_pred_%d_output" % d,
                                   4, 1, 0, wd(), flat=True))
        L = Msg([("name", "cls_score_output_concat")])
        for d in (1, 2, 4):
            L.add("bottom", "cls_score_%d_output" % d)
        L.add("top", "cls_score_reshape_output")
        L.add("type", "Concat")
        L.add("concat_param", Msg([("axis", 2)]))
        net.add("layer", L)
        L = Msg([("name", "bbox_pred_output_concat")])
        for d in (1, 2, 4):
            L.add("bottom", "bbox_pred_%d_output" % d)
        L.add("top", "bbox_pred_output")
        L.add("type", "Concat")
        L.add("concat_param", Msg([("axis", 1)]))
        net.add("layer", L)
    else:
        net.add("layer", _conv("head", "conv4_fuse_final", "head", 128, 3, 1, wd(), flat=True))
        net.add("layer", _relu("head_relu", "head"))
        net.add("layer", _conv("cls_score", "head", "cls_score_output", 6, 1, 0, wd(), flat=True))
        net.add("layer", _conv("bbox_pred", "head", "bbox_pred_output", 12, 1, 0, wd(), flat=True))
        net.add("layer", Msg([("bottom", "cls_score_output"), ("top", "cls_score_reshape_output"),
                              ("name", "cls_reshape"), ("type", "Reshape"),
                              ("reshape_param", Msg([("shape", Msg([("dim", 0), ("dim", 2), ("dim", -1),
                                                                    ("dim", 0)]))]))]))
    net.add("layer", Msg([("name", "cls_prob"), ("type", "Softmax"),
                          ("bottom", "cls_score_reshape_output"), ("top", "cls_prob_output")]))
    net.add("layer", Msg([("name", "cls_prob_reshape"), ("type", "Reshape"), ("bottom", "cls_prob_output"),
                          ("top", "cls_prob_reshape_output"),
                          ("reshape_param", Msg([("shape", Msg([("dim", 0), ("dim", 6), ("dim", -1),
                                                                ("dim", 0)]))]))]))
    net.add("layer", Msg([
        ("name", "proposal"), ("type", "Python"), ("bottom", "cls_prob_reshape_output"),
        ("bottom", "bbox_pred_output"), ("bottom", "im_info"), ("top", "boxes"), ("top", "cls_prob"),
        ("python_param", Msg([("module", "lib.layers.proposal_layer"), ("layer", "ProposalLayer"),
                              ("param_str", "{'feat_stride': [8,8,8],'scales': [1,2,4], 'ratios':[1,]}")]))]))
    return net


def write_test_prototxt(path, different_dilation=True, dim_red=True):
    """Convenience: template (+ dimension reduction) straight to a file."""
    pb = build_test_template(different_dilation)
    if dim_red and different_dilation:
        old = cfg.MODEL.DIFFERENT_DILATION.ENABLE
        cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
        try:
            pb = _add_dimension_reduction(pb)
        finally:
            cfg.MODEL.DIFFERENT_DILATION.ENABLE = old
    with open(path, "w") as f:
        f.write(dumps(pb))
    return path

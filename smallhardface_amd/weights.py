"""Synthetic, seeded parameters for the detector (no trained .caffemodel exists in the
reference tree: .gitignore:75, README links to SharePoint).  Product-side generator used by
bench.py / smoke / tests; the CPU oracle loads the very same arrays, so both sides hold
identical values."""
import numpy as np

F32 = np.float32


def _ints(msg, name, default):
    v = msg.getall(name) if msg is not None else []
    return int(v[0]) if v else default


def bilinear_filler(shape):
    """BilinearFiller::Fill (caffe/include/caffe/filler.hpp:248-258)."""
    w = np.empty(shape, dtype=F32)
    kh, kw = shape[2], shape[3]
    f = int(np.ceil(kw / 2.))
    c = F32((kw - 1) / (2. * f))
    flat = w.reshape(-1)
    for i in range(flat.size):
        x = F32(i % kw)
        yv = F32((i // kw) % kh)
        flat[i] = (F32(1) - abs(x / F32(f) - c)) * (F32(1) - abs(yv / F32(f) - c))
    return w


def synth_params(net_msg, seed=1234, cls_bias=4.0):
    """Seeded synthetic weights: He-normal conv weights (the conv
    that reads the mean-subtracted image is scaled by 1/64 so activations and
    logits are O(1) like a trained net's), zero biases except cls_score* = (+b,-b)
    (b=4: ~0.5% of anchors above 0.05 on the bench pyramid, WIDER-like), bbox_pred* x0.1, bilinear deconv; layers
    naming the same ``param {name:}`` share one tensor.  Returns
    {layer_name: [w, b]} with Caffe blob shapes.  Used by BOTH the oracle net and
    the HIP net (through Net.params) so they hold identical values."""
    rng = np.random.default_rng(seed)
    shapes = infer_channels(net_msg)
    params = {}
    shared = {}
    for L in net_msg.getall("layer"):
        t = str(L.get("type"))
        if t not in ("Convolution", "Deconvolution"):
            continue
        name = str(L.get("name"))
        cp = L.get("convolution_param")
        cin = shapes[str(L.get("bottom"))]
        cout = int(cp.get("num_output"))
        k = _ints(cp, "kernel_size", 1)
        group = _ints(cp, "group", 1)
        pnames = [str(p.get("name", "")) for p in L.getall("param")]
        bias_term = str(cp.get("bias_term", "true")) != "false"
        if t == "Deconvolution":
            w = bilinear_filler((cin, cout // group, k, k))
            params[name] = [w] + ([np.zeros(cout, F32)] if bias_term else [])
            continue
        key = pnames[0] if pnames and pnames[0] else None
        if key and key in shared:
            params[name] = shared[key]
            continue
        std = np.sqrt(2.0 / (cin // group * k * k))
        w = rng.normal(0, std, (cout, cin // group, k, k)).astype(F32)
        b = np.zeros(cout, F32)
        if name.startswith("cls_score"):
            # first half of the channels = bg (bias +b), second half = fg (bias -b)
            b[:cout // 2] = cls_bias
            b[cout // 2:] = -cls_bias
        if name.startswith("bbox_pred"):
            w *= F32(0.1)
        if str(L.get("bottom")) == "data":
            w *= F32(1.0 / 64.0)
        params[name] = [w, b] if bias_term else [w]
        if key:
            shared[key] = params[name]
    return params


def infer_channels(net_msg, in_ch=3):
    """Channel count of every blob (enough shape inference to size weights)."""
    ch = {"data": in_ch, "im_info": 1}
    for n, shp in zip(net_msg.getall("input"), net_msg.getall("input_shape")):
        dims = shp.getall("dim")
        ch[str(n)] = int(dims[1]) if len(dims) > 1 else 1
    for L in net_msg.getall("layer"):
        t = str(L.get("type"))
        bots = [str(b) for b in L.getall("bottom")]
        tops = [str(x) for x in L.getall("top")]
        if t in ("Convolution", "Deconvolution"):
            ch[tops[0]] = int(L.get("convolution_param").get("num_output"))
        elif t == "Concat":
            axis = _ints(L.get("concat_param"), "axis", 1)
            ch[tops[0]] = sum(ch[b] for b in bots) if axis == 1 else ch[bots[0]]
        elif t == "Reshape":
            dims = [int(d) for d in L.get("reshape_param").get("shape").getall("dim")]
            ch[tops[0]] = dims[1] if dims[1] > 0 else ch[bots[0]]
        elif t == "Python":
            for tp in tops:
                ch[tp] = 1
        elif t == "Input":
            for tp, shp in zip(tops, L.get("input_param").getall("shape")):
                ch[tp] = int(shp.getall("dim")[1])
        else:
            for tp in tops:
                ch[tp] = ch[bots[0]]
    return ch

"""CPU study for the next kernel design: would an UNSCALED low part with ONE accumulator keep the split-fp16
convolution inside the 1e-4 score bar?

    python tools/single_acc_study.py [side]        (default side 256; C1 is 512)

Emulates, with the oracle's conv (numpy / fp32 BLAS accumulation) at every MFMA convolution of the detector:
  scaled   -- what ships: hi = fp16(x), lo = fp16((x - hi) * 2^11), out = hi*hi + (hi*lo + lo*hi) * 2^-11, two accumulators
  unscaled -- lo = fp16(x - hi) as it is (subnormals and all: tools/mfma_denorm.hip shows the MFMA honours them), weights
              pre-scaled by a per-layer power of two so that their low parts stay normal, all three products in one sum
  fp16     -- hi only (the bottom of the reduced ladder), for scale
and reports max |score - fp32 oracle| over all anchors of one level.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from oracle import oracle as O
    from smallhardface_amd import prototxt as P
    from smallhardface_amd.config import cfg_from_file
    from tests import helpers as H
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    msg = H.detector_msg(True)
    params = O.synth_params(msg, seed=1234, cls_bias=4.0)
    data = H.synth_image_blob(side, side, seed=21)
    info = np.array([[side, side, 1.0]], np.float32)
    plain = O.convolution

    def run(conv):
        O.convolution = conv
        try:
            net = O.OracleNet(msg, params=params)
            net.blobs['data'].reshape(*data.shape)
            net.blobs['im_info'].reshape(1, 3)
            net.forward(data=data, im_info=info)
            return net.blobs["cls_prob_reshape_output"].data.copy(), net.blobs["bbox_pred_output"].data.copy()
        finally:
            O.convolution = plain

    f16, f32 = np.float16, np.float32

    def is_mfma(x, w):          # the first layer (Cin 3) and the 2- / 4-channel score / delta layers stay fp32 in the product
        return w.shape[1] >= 32 and w.shape[0] >= 64

    def scaled(x, w, b, **kw):
        if not is_mfma(x, w):
            return plain(x, w, b, **kw)
        xh = x.astype(f16); xl = ((x - xh.astype(f32)) * f32(2048)).astype(f16)
        wh = w.astype(f16); wl = ((w - wh.astype(f32)) * f32(2048)).astype(f16)
        z = None
        main = plain(xh.astype(f32), wh.astype(f32), z, **kw)
        corr = plain(xh.astype(f32), wl.astype(f32), z, **kw) + plain(xl.astype(f32), wh.astype(f32), z, **kw)
        y = main + corr * f32(1.0 / 2048)
        return y + b.reshape(1, -1, 1, 1) if b is not None else y

    def unscaled(x, w, b, **kw):
        if not is_mfma(x, w):
            return plain(x, w, b, **kw)
        s = f32(2.0 ** np.floor(np.log2(8.0 / np.abs(w).max())))      # max |w * s| in [8, 16)
        ws = w * s
        xh = x.astype(f16); xl = (x - xh.astype(f32)).astype(f16)
        wh = ws.astype(f16); wl = (ws - wh.astype(f32)).astype(f16)
        z = None
        y = plain(xh.astype(f32), wh.astype(f32), z, **kw) + plain(xh.astype(f32), wl.astype(f32), z, **kw) + \
            plain(xl.astype(f32), wh.astype(f32), z, **kw)
        y = y * (f32(1.0) / s)
        return y + b.reshape(1, -1, 1, 1) if b is not None else y

    def half(x, w, b, **kw):
        if not is_mfma(x, w):
            return plain(x, w, b, **kw)
        y = plain(x.astype(f16).astype(f32), w.astype(f16).astype(f32), None, **kw)
        return y + b.reshape(1, -1, 1, 1) if b is not None else y

    ref_s, ref_d = run(plain)
    print("level %dx%d, %d anchors; fg scores > 0.05: %d" % (side, side, ref_s.shape[2] * ref_s.shape[3] * 3,
                                                             int((ref_s[0, 3:] > 0.05).sum())))
    for name, fn in (("scaled (ships)", scaled), ("unscaled, one accumulator", unscaled), ("fp16 only", half)):
        s_, d_ = run(fn)
        print("%-28s max |dscore| %.3e   max |ddelta| %.3e" % (name, np.abs(s_ - ref_s).max(), np.abs(d_ - ref_d).max()))


if __name__ == "__main__":
    main()

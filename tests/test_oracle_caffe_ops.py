"""Caffe's own known-answer / naive-reference unit tests, restated against the
numpy oracle (the vendored Caffe cannot be built here; see oracle/oracle.py).

  test_pooling_layer.cpp:49-119      TestForwardSquare literal
  test_deconvolution_layer.cpp:91-137 TestSimpleDeconvolution overlap counts
  test_filler.cpp:241-280            BilinearFillerTest formula (n=6,7)
  test_convolution_layer.cpp:21-139  naive caffe_conv vs layer, 1e-4
        (:231 simple, :267 dilated, :443 1x1, :470 group)
  test_softmax_layer.cpp:43-75       exp/sum reference, 1e-4
  test_concat_layer.cpp / test_reshape_layer.cpp value identity
"""
import numpy as np
import pytest

from oracle import oracle as O


def naive_conv(x, w, b, pad, stride, dil, group):
    """Loop nest of caffe_conv (test_convolution_layer.cpp:21-139)."""
    N, C, H, W = x.shape
    Co, Cg, kh, kw = w.shape
    Ho = O.conv_out_size(H, kh, pad, stride, dil)
    Wo = O.conv_out_size(W, kw, pad, stride, dil)
    y = np.zeros((N, Co, Ho, Wo), dtype=np.float64)
    og = Co // group
    for n in range(N):
        for g in range(group):
            for o in range(og):
                for k in range(Cg):
                    for yy in range(Ho):
                        for xx in range(Wo):
                            for p in range(kh):
                                for q in range(kw):
                                    iy = yy * stride - pad + p * dil
                                    ix = xx * stride - pad + q * dil
                                    if 0 <= iy < H and 0 <= ix < W:
                                        y[n, o + g * og, yy, xx] += x[n, k + g * Cg, iy, ix] * w[o + g * og, k, p, q]
    return y + b[None, :, None, None]


def test_max_pool_known_answer():
    row = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], dtype=np.float32)
    x = np.tile(row, (2, 2, 1, 1))
    y = O.max_pool(x, 2, 1, 0)  # the test leaves stride at its default 1
    assert y.shape == (2, 2, 2, 4)
    exp = np.array([[9, 5, 5, 8], [9, 5, 5, 8]], dtype=np.float32)
    np.testing.assert_array_equal(y, np.tile(exp, (2, 2, 1, 1)))


def test_max_pool_ceil_and_fast_path():
    rng = np.random.default_rng(0)
    for H, W in [(8, 10), (7, 9), (2, 2), (5, 6)]:
        x = rng.normal(size=(1, 3, H, W)).astype(np.float32)
        a = O.max_pool(x, 2, 2, 0)
        assert a.shape[2:] == ((H + 1) // 2, (W + 1) // 2)  # ceil((H-2)/2)+1
        np.testing.assert_array_equal(a, O.max_pool_2x2_fast(x))


def test_deconv_known_answer():
    x = np.ones((2, 3, 6, 4), dtype=np.float32)
    w = np.ones((3, 4, 3, 3), dtype=np.float32)
    b = np.full(4, 0.1, dtype=np.float32)
    y = O.deconvolution(x, w, b, pad=0, stride=2, group=1)
    assert y.shape == (2, 4, 13, 9)
    Ht, Wt = y.shape[2:]
    for h in range(Ht):
        for ww in range(Wt):
            e = 3.1
            ho = h % 2 == 0 and 0 < h < Ht - 1
            wo = ww % 2 == 0 and 0 < ww < Wt - 1
            if ho and wo:
                e += 9
            elif ho or wo:
                e += 3
            np.testing.assert_allclose(y[:, :, h, ww], e, atol=1e-4)


@pytest.mark.parametrize("n", [6, 7, 4])
def test_bilinear_filler(n):
    w = O.bilinear_filler((5, 2, n, n))
    f = int(np.ceil(n / 2.))
    c = (n - 1) / (2. * f)
    for j in range(n * n):
        x, y = j % n, (j // n) % n
        e = (1 - abs(x / f - c)) * (1 - abs(y / f - c))
        np.testing.assert_allclose(w.reshape(10, -1)[:, j], e, atol=0.01)
    if n == 4:  # the detector's upsampler taps
        np.testing.assert_allclose(w[0, 0, 0], np.array([0.25, 0.75, 0.75, 0.25]) * 0.25, atol=1e-7)


def test_bilinear_deconv_is_upsample():
    """k4 s2 p1 depthwise deconv with bilinear taps doubles the map (out = 2H)."""
    x = np.random.default_rng(1).normal(size=(1, 5, 7, 9)).astype(np.float32)
    y = O.deconvolution(x, O.bilinear_filler((5, 1, 4, 4)), None, pad=1, stride=2, group=5)
    assert y.shape == (1, 5, 14, 18)
    # interior samples are the 0.75/0.25 blends
    np.testing.assert_allclose(y[0, :, 3, 3], (0.75 * 0.75 * x[0, :, 1, 1] + 0.75 * 0.25 * x[0, :, 1, 2] +
                                              0.25 * 0.75 * x[0, :, 2, 1] + 0.25 * 0.25 * x[0, :, 2, 2]), atol=1e-5)


@pytest.mark.parametrize("cfg", [
    dict(C=3, Co=4, k=3, pad=0, stride=2, dil=1, group=1),   # TestSimpleConvolution
    dict(C=3, Co=4, k=3, pad=0, stride=1, dil=2, group=1),   # TestDilatedConvolution
    dict(C=3, Co=4, k=1, pad=0, stride=1, dil=1, group=1),   # Test1x1Convolution
    dict(C=6, Co=3, k=3, pad=0, stride=2, dil=1, group=3),   # TestSimpleConvolutionGroup
    dict(C=4, Co=5, k=3, pad=1, stride=1, dil=1, group=1),   # the detector's 3x3 pad 1
    dict(C=4, Co=5, k=3, pad=4, stride=1, dil=4, group=1),   # head_4
])
def test_conv_vs_naive(cfg):
    rng = np.random.default_rng(7)
    x = rng.normal(size=(2, cfg["C"], 6, 5 + 2 * cfg["dil"])).astype(np.float32)
    w = rng.normal(size=(cfg["Co"], cfg["C"] // cfg["group"], cfg["k"], cfg["k"])).astype(np.float32)
    b = rng.normal(size=cfg["Co"]).astype(np.float32)
    y = O.convolution(x, w, b, cfg["pad"], cfg["stride"], cfg["dil"], cfg["group"])
    ref = naive_conv(x, w, b, cfg["pad"], cfg["stride"], cfg["dil"], cfg["group"])
    np.testing.assert_allclose(y, ref, atol=1e-4)
    # chunked col buffer gives the same answer
    y2 = O.convolution(x, w, b, cfg["pad"], cfg["stride"], cfg["dil"], cfg["group"], col_bytes=1)
    np.testing.assert_allclose(y, y2, atol=1e-5)


def test_softmax():
    x = np.random.default_rng(3).normal(size=(2, 10, 2, 3)).astype(np.float32)
    y = O.softmax(x, 1)
    e = np.exp(x.astype(np.float64))
    np.testing.assert_allclose(y, e / e.sum(1, keepdims=True), atol=1e-4)
    np.testing.assert_allclose(y.sum(1), 1.0, atol=1e-5)


def test_reshape_rules():
    assert O.caffe_reshape((1, 2, 42, 7), [0, 6, -1, 0]) == (1, 6, 14, 7)
    assert O.caffe_reshape((1, 6, 14, 7), [0, 2, -1, 0]) == (1, 2, 42, 7)


def test_net_wiring_and_sharing():
    """Graph restatement: outputs, shared head weights, channel order of the
    (1,6,h,w) probability blob = [bg1,bg2,bg4,fg1,fg2,fg4]."""
    from smallhardface_amd import prototxt as P
    from smallhardface_amd.config import cfg
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    pb = P._add_dimension_reduction(P.build_test_template(True))
    net = O.OracleNet(pb)
    assert net.inputs == ["data", "im_info"]
    assert set(net.outputs) == {"boxes", "cls_prob"}
    assert net.params["head_1"][0] is net.params["head_2"][0] is net.params["head_4"][0]
    assert net.params["conv4_fuse_final_dim_red"][0].shape == (128, 512, 3, 3)
    assert net.params["head_1"][0].shape == (128, 128, 3, 3)
    net.blobs["data"].reshape(1, 3, 32, 48)
    net.blobs["im_info"].reshape(1, 3)
    rng = np.random.default_rng(0)
    out = net.forward(data=rng.normal(0, 60, (1, 3, 32, 48)).astype(np.float32),
                      im_info=np.array([[30, 45, 1.0]], np.float32))
    assert net.blobs["conv4_fuse"].shape == (1, 512, 4, 6)
    assert net.blobs["cls_prob_reshape_output"].shape == (1, 6, 4, 6)
    p = net.blobs["cls_prob_reshape_output"].data
    np.testing.assert_allclose(p[:, :3] + p[:, 3:], 1.0, atol=1e-6)
    assert out["boxes"].shape[1] == 5 and out["cls_prob"].shape[1] == 2
    assert out["boxes"][:, 3].max() <= 44 and out["boxes"][:, 4].max() <= 29

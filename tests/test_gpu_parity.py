"""GPU parity tests (run on an MI355X: ``pytest -m gpu``).  Everything goes through
the C ABI (libshf_hip.so) and is checked against the CPU oracle / golden vectors.

Tolerances: conv-stack activations 2e-5 relative to the blob's max (fp32 MFMA vs
OpenBLAS summation order); scores 1e-4 absolute (north-star bar); box coordinates
1e-3 px; integer indices / orders exact.
"""
import numpy as np
import pytest

from oracle import oracle as O
from smallhardface_amd import prototxt as P
from smallhardface_amd.config import cfg
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fp32", "f16x3"])
def conv_mode(request):
    """Every test runs with the exact fp32 MFMA convs and with the split-fp16 MFMA convs
    (C ABI shf_net_set_conv_mode / SHF_CONV_MODE): both must meet the same parity bars."""
    import os
    old = os.environ.get("SHF_CONV_MODE")
    os.environ["SHF_CONV_MODE"] = "1" if request.param == "f16x3" else "0"
    yield request.param
    if old is None:
        os.environ.pop("SHF_CONV_MODE", None)
    else:
        os.environ["SHF_CONV_MODE"] = old

ACT_TOL = 2e-5
SCORE_TOL = 1e-4
BOX_TOL = 1e-3


def unmatched_rows(a, b, box_tol=0.05):
    """Detections of `a` without a partner in `b` (score within SCORE_TOL, every coordinate within box_tol): rows are in
    score order, and near-ties may swap between two arithmetic paths."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    n = 0
    for row in a:
        near = np.abs(b[:, 4] - row[4]) < SCORE_TOL
        if not near.any() or np.abs(b[near, :4] - row[:4]).max(axis=1).min() >= box_tol:
            n += 1
    return n


def conv_layer(name, bottom, nout, k, pad, dil=1, relu=True):
    s = ('layer { name: "%s" type: "Convolution" bottom: "%s" top: "%s" convolution_param { num_output: %d '
         'kernel_size: %d pad: %d dilation: %d } }\n' % (name, bottom, name, nout, k, pad, dil))
    if relu:
        s += 'layer { name: "%s_relu" type: "ReLU" bottom: "%s" top: "%s" }\n' % (name, name, name)
    return s


@pytest.mark.parametrize("cin,cout,k,dil,h,w,relu", [
    (64, 64, 3, 1, 37, 53, True),      # BN=64 tile (conv1_2 shape class), ragged edges
    (64, 128, 3, 1, 32, 48, True),     # BN=128 tile
    (128, 256, 3, 1, 19, 21, False),   # no ReLU, negative outputs kept
    (512, 512, 3, 1, 16, 24, True),    # conv4/5 class, 16 K-chunks
    (128, 384, 3, 1, 45, 83, True),    # three cout tiles, an odd number of 16-row tiles (dual-tile family: dummy second tile)
    (160, 128, 3, 1, 23, 17, True),    # Cin = 10 x 16: an odd number of 32-channel chunks' worth
    (512, 256, 1, 1, 9, 13, True),     # 1x1 (conv5_256 / conv4_256)
    (512, 256, 1, 1, 23, 29, True),    # 1x1 GEMM kernel: three 256-pixel blocks, the last one ragged
    (96, 512, 1, 1, 20, 20, False),    # ... an odd number of 32-channel chunks, two cout tiles, no ReLU
    (128, 128, 3, 2, 22, 26, True),    # head_2
    (128, 128, 3, 4, 22, 26, True),    # head_4
    (128, 128, 3, 4, 5, 6, True),      # map smaller than the dilation halo
    # what the family / the GEMM kernel do not take runs on the 8-wave kernel (conv_f16x3_8w.h) -- reached by shape, no knob:
    (32, 128, 3, 2, 21, 19, True),     # dilated with Cin < 64: the 8-wave kernel's DIL = 2 form (BN = 64 tiles)
    (32, 64, 3, 4, 14, 30, False),     # ... DIL = 4, Cout 64
    (512, 128, 1, 1, 18, 22, True),    # 1x1 with Cout % 256 != 0: the 8-wave kernel's KS = 1 form, BN = 128
    (64, 64, 1, 1, 11, 12, True),      # ... BN = 64
])
def test_conv_mfma(cin, cout, k, dil, h, w, relu):
    pad = dil if k == 3 else 0
    txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, k, pad, dil, relu),
                             3, h, w)
    msg = P.parse(txt)
    gnet, onet = H.make_pair(msg, seed=5)
    # give the biases some life
    rng = np.random.default_rng(3)
    for name in ("c0", "c1"):
        b = rng.normal(0, 0.5, onet.params[name][1].shape).astype(np.float32)
        onet.params[name][1][...] = b
    H.load_params(gnet, onet.params)
    data = rng.normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    assert go["c1"].shape == oo["c1"].shape == (1, cout, h, w)
    assert H.rel_err(go["c1"], oo["c1"]) < ACT_TOL
    assert H.rel_err(gnet.blobs["c0"].data, onet.blobs["c0"].data) < ACT_TOL
    if not relu:
        assert (go["c1"] < 0).any()


def _fuzz_conv_cases(n=14, seed=2025):
    """Seeded random layer shapes over everything the split-fp16 launcher distinguishes: Cin below / at / above the family's
    64, Cout with 64- / 128- / 256-multiples, 1x1 and 3x3 at dilation 1 / 2 / 4, ragged maps down to smaller than a halo."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 3, 3]))
        dil = int(rng.choice([1, 1, 2, 4])) if k == 3 else 1
        cin = int(rng.choice([32, 64, 96, 128, 160, 256, 512]))
        cout = int(rng.choice([64, 128, 192, 256, 384, 512]))
        h, w = int(rng.integers(5, 60)), int(rng.integers(5, 70))
        out.append((cin, cout, k, dil, h, w, bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("cin,cout,k,dil,h,w,relu", _fuzz_conv_cases())
def test_conv_mfma_seeded_shapes(cin, cout, k, dil, h, w, relu):
    test_conv_mfma(cin, cout, k, dil, h, w, relu)


def shared_conv_layer(name, bottom, nout, dil):
    return ('layer { name: "%s" type: "Convolution" bottom: "%s" top: "%s" param { name: "hw" } param { name: "hb" } '
            'convolution_param { num_output: %d kernel_size: 3 pad: %d dilation: %d } }\n'
            'layer { name: "%s_relu" type: "ReLU" bottom: "%s" top: "%s" }\n' % (name, bottom, name, nout, dil, dil, name, name, name))


@pytest.mark.parametrize("cin,h,w", [
    (128, 22, 26),      # the detector's shape class, ragged tiles in both directions
    (128, 5, 6),        # a map smaller than the dilation-4 halo
    (128, 64, 48),      # whole 8 x 16 tiles only (the `interior` fast path)
    (160, 17, 33),      # Cin = 10 sixteen-channel chunks
])
def test_three_shared_weight_dilated_heads_one_launch(cin, h, w, conv_mode):
    """head_1 / head_2 / head_4 (models/test_different_dilation_template.prototxt:480-552: same bottom, shared `head_w` /
    `head_b`, dilation 1 / 2 / 4) run as ONE launch in the split-fp16 modes (conv_f16x3_h3.h): each against the oracle, and
    through the runtime's profile counters the launch really is the fused kernel."""
    txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + shared_conv_layer("h1", "c0", 128, 1) +
                             shared_conv_layer("h2", "c0", 128, 2) + shared_conv_layer("h4", "c0", 128, 4), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=9)
    rng = np.random.default_rng(6)
    for name in ("c0", "h1"):
        onet.params[name][1][...] = rng.normal(0, 0.5, onet.params[name][1].shape).astype(np.float32)
    assert onet.params["h4"][1] is onet.params["h1"][1] or np.array_equal(onet.params["h4"][1], onet.params["h1"][1])
    H.load_params(gnet, onet.params)
    data = rng.normal(0, 1, (1, 3, h, w)).astype(np.float32)
    gnet.prof_enable(True)
    gnet.prof_reset()
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    prof = gnet.prof_read()
    gnet.prof_enable(False)
    for name in ("h1", "h2", "h4"):
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert a.shape == b.shape == (1, 128, h, w), name
        assert H.rel_err(a, b) < ACT_TOL, name
    assert not np.array_equal(gnet.blobs["h1"].data, gnet.blobs["h2"].data)
    fused = prof.get("conv_mfma_f16x3_heads3_kernel<true, 3>", {}).get("launches", 0)
    assert fused == (1 if conv_mode == "f16x3" else 0), prof.keys()


def test_conv_into_an_unaligned_concat_slice():
    """A convolution whose output is a channel slice that does NOT start on a 16-byte boundary (a 66-channel neighbour in
    front of it in a Concat): the vector epilogues do not apply, the 8-wave / fp32 kernels store value by value
    (conv_store_tile) -- the path unaligned views take, reached by the graph alone."""
    h, w = 19, 27
    txt = H.single_layer_net(conv_layer("c0", "data", 64, 3, 1) + conv_layer("ca", "c0", 66, 3, 1) + conv_layer("cb", "c0", 64, 3, 1) +
                             'layer { name: "cat" type: "Concat" bottom: "ca" bottom: "cb" top: "cat" }\n' +
                             conv_layer("cc", "cat", 64, 1, 0), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=21)
    rng = np.random.default_rng(8)
    for name in ("c0", "ca", "cb"):
        onet.params[name][1][...] = rng.normal(0, 0.5, onet.params[name][1].shape).astype(np.float32)
    H.load_params(gnet, onet.params)
    data = rng.normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    for name in ("ca", "cb", "cat", "cc"):
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert a.shape == b.shape, name
        assert H.rel_err(a, b) < ACT_TOL, name


def test_conv_identity_is_transpose_detecting():
    """A = I with an ASYMMETRIC weight pattern: catches swapped rows/cols in the MFMA epilogue."""
    h, w, c = 8, 16, 64
    txt = H.single_layer_net(conv_layer("c0", "data", c, 3, 1, relu=False) +
                             conv_layer("c1", "c0", 128, 1, 0, relu=False), 3, h, w)
    msg = P.parse(txt)
    gnet, onet = H.make_pair(msg, seed=1)
    w1 = np.zeros((128, c, 1, 1), np.float32)
    for o in range(128):
        for i in range(c):
            w1[o, i, 0, 0] = (o * 3 + i * 7) % 11 - 5 + 0.25 * (o > i)
    onet.params["c1"][0][...] = w1
    H.load_params(gnet, onet.params)
    data = np.random.default_rng(0).normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    assert H.rel_err(go["c1"], oo["c1"]) < ACT_TOL


@pytest.mark.parametrize("cin,cout,h,w,relu,consumer", [
    (128, 128, 19, 21, True, True),     # conv2_2 class: pooled output only (a consumer reads the pool), ragged edges
    (128, 256, 35, 18, False, True),    # no ReLU: the pool's neutral element for outside pixels is -FLT_MAX, not 0
    (256, 128, 33, 47, True, False),    # pooled AND un-pooled output both read (the conv4_3 case)
])
def test_four_wave_kernel_fused_pool(cin, cout, h, w, relu, consumer):
    """The register epilogue of the 4-wave kernels (conv_epilogue_regs): MFMA as D[cout][pixel], half-wave quad exchange,
    2x2/2 max-pool as a DPP quad max with Caffe's clipped windows on odd maps -- against the oracle's conv + pool."""
    pool = 'layer { name: "p" type: "Pooling" bottom: "c1" top: "p" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n'
    txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, 3, 1, 1, relu) + pool +
                             conv_layer("c2", "p", 64, 1, 0) + ("" if consumer else conv_layer("c3", "c1", 64, 1, 0)), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=11)
    rng = np.random.default_rng(4)
    for name in ("c0", "c1"):
        onet.params[name][1][...] = rng.normal(0, 0.5, onet.params[name][1].shape).astype(np.float32)
    H.load_params(gnet, onet.params)
    data = rng.normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    for name in ("p", "c2") + (() if consumer else ("c1", "c3")):
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert a.shape == b.shape, name
        assert H.rel_err(a, b) < ACT_TOL, name
    if not relu:
        assert gnet.blobs["p"].data.min() < 0      # negative maxima survive


def test_pool_deconv_concat():
    h, w = 18, 26
    txt = H.single_layer_net(
        conv_layer("c0", "data", 64, 3, 1) +
        'layer { name: "p" type: "Pooling" bottom: "c0" top: "p" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n' +
        conv_layer("c1", "p", 64, 1, 0) +
        'layer { name: "up" type: "Deconvolution" bottom: "c1" top: "up" convolution_param { kernel_size: 4 stride: 2 '
        'num_output: 64 group: 64 pad: 1 weight_filler: { type: "bilinear" } bias_term: false } }\n' +
        conv_layer("c2", "c0", 64, 1, 0) +
        'layer { name: "cat" type: "Concat" bottom: "up" bottom: "c2" top: "cat" concat_param { axis: 1 } }\n' +
        conv_layer("c3", "cat", 64, 3, 1), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=9)
    data = np.random.default_rng(1).normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    np.testing.assert_array_equal(gnet.blobs["p"].data >= 0, True)
    for name in ("c0", "p", "c1", "up", "c2", "cat", "c3"):
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert a.shape == b.shape, name
        assert H.rel_err(a, b) < ACT_TOL, name
    # max-pool is a pure selection: exact given identical inputs
    np.testing.assert_array_equal(gnet.blobs["p"].data, O.max_pool_2x2_fast(gnet.blobs["c0"].data))


def test_pool_known_answer_gpu():
    """Caffe's TestForwardSquare literal (test_pooling_layer.cpp:49-119) needs stride 1 on a 3x5 map;
    the GPU pool kernel is generic in k/stride."""
    # route the literal through a 1x1 identity conv so the blob is NHWC on the device
    row = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    txt = H.single_layer_net(conv_layer("c0", "data", 64, 3, 1, relu=False) +
                             'layer { name: "p" type: "Pooling" bottom: "c0" top: "p" pooling_param { pool: MAX '
                             'kernel_size: 2 stride: 1 } }\n', 4, 3, 5)
    gnet, onet = H.make_pair(P.parse(txt), seed=2)
    wid = np.zeros((64, 4, 3, 3), np.float32)
    wid[:4, :, 1, 1] = np.eye(4)
    onet.params["c0"][0][...] = wid
    H.load_params(gnet, onet.params)
    data = np.tile(row, (1, 4, 1, 1)).astype(np.float32)
    go, _ = H.run_both(gnet, onet, data, np.array([[3, 5, 1]], np.float32))
    exp = np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32)
    np.testing.assert_array_equal(go["p"][0, :4], np.tile(exp, (4, 1, 1)))


@pytest.mark.parametrize("dd,h,w,im", [(True, 64, 80, (61, 77)), (True, 112, 112, (100, 100)),
                                       (False, 48, 64, (48, 64))])
def test_detector_end_to_end(dd, h, w, im):
    msg = H.detector_msg(dd)
    gnet, onet = H.make_pair(msg)
    data = H.synth_image_blob(h, w, seed=4)
    info = np.array([[im[0], im[1], 0.75]], np.float32)
    go, oo = H.run_both(gnet, onet, data, info)
    names = ["conv1_1", "conv1_2", "pool1", "conv2_2", "conv3_3", "conv4_3", "pool4", "conv5_3", "conv5_256",
             "conv5_256_up", "conv4_256", "conv4_fuse", "conv4_fuse_final"]
    names += ["head_1", "head_2", "head_4"] if dd else ["head"]
    for n in names:
        a, b = gnet.blobs[n].data, onet.blobs[n].data
        assert a.shape == b.shape, n
        assert H.rel_err(a, b) < 5e-5, n
    gp = gnet.blobs["cls_prob_reshape_output"].data
    gd = gnet.blobs["bbox_pred_output"].data
    assert np.abs(gp - onet.blobs["cls_prob_reshape_output"].data).max() < SCORE_TOL
    assert np.abs(gd - onet.blobs["bbox_pred_output"].data).max() < 1e-3
    # stage parity of the proposal tail on IDENTICAL inputs: order / indices exact
    pb, pp = O.proposal_forward(gp, gd, info)
    gb, gs = go["boxes"], go["cls_prob"]
    assert gb.shape == pb.shape and gs.shape == pp.shape
    np.testing.assert_array_equal(gs, pp)            # same scores in the same order
    assert np.abs(gb - pb).max() < BOX_TOL
    assert gb[:, 3].max() <= im[1] - 1 and gb[:, 4].max() <= im[0] - 1
    # end to end vs the oracle net: match rows by score rank
    ob, os_ = oo["boxes"], oo["cls_prob"]
    n = min(len(ob), len(gb))
    assert abs(len(ob) - len(gb)) <= max(2, 0.01 * len(ob))
    assert np.abs(np.sort(gs[:, 1])[::-1][:n] - np.sort(os_[:, 1])[::-1][:n]).max() < SCORE_TOL
    # blobs whose producers are folded into the tail are materialised on demand (pycaffe exposes every blob after forward(),
    # pycaffe.py:24-32): the predictors' tops against the oracle net's
    for name in (["cls_score_1_output", "bbox_pred_4_output"] if dd else ["cls_score_output"]):
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-4 * max(1.0, float(np.abs(b).max())), name


def test_net_surface():
    msg = H.detector_msg(True)
    gnet, onet = H.make_pair(msg)
    assert list(gnet.blobs.keys())[:3] == ["data", "im_info", "conv1_1"]
    assert gnet.inputs == ["data", "im_info"] and set(gnet.outputs) == {"boxes", "cls_prob"}
    assert 'boxes' in gnet.blobs
    assert list(gnet.blobs.keys()) == list(onet.blobs.keys())
    with pytest.raises(Exception, match="Input blob arguments do not match net inputs."):
        gnet.forward(data=np.zeros((1, 3, 16, 16), np.float32))
    gnet.blobs["data"].reshape(1, 3, 16, 16)
    with pytest.raises(Exception, match="Input is not batch sized"):
        gnet.forward(data=np.zeros((2, 3, 16, 16), np.float32), im_info=np.zeros((1, 3), np.float32))
    assert gnet.params["head_1"][0].shape == (128, 128, 3, 3)
    # shared head weights alias one tensor (net.cpp:421-513)
    gnet.params["head_2"][0].data[0, 0, 0, 0] = 42.0
    assert gnet.params["head_4"][0].data[0, 0, 0, 0] == 42.0
    d = gnet.blobs["data"].data
    d[...] = 1.0
    assert gnet.blobs["data"].data[0, 0, 0, 0] == 1.0  # writable zero-copy view


@pytest.mark.parametrize("dd", [True, False])
def test_every_blob_is_readable_after_forward(dd, conv_mode):
    """pycaffe exposes ALL of net.blobs after forward() (pycaffe.py:24-32, _caffe.cpp:222-242).  Here the class-score /
    bbox 1x1 convs, the score concat (plain template: reshape) and the softmax are folded into the detection tail and
    never written to HBM as blobs of their own; reading one re-orders the tail's logits workspace (resp. the softmax blob
    the proposal layer reads) into the blob's own NCHW shape on demand.  Every name in net.blobs is read -- same shape as
    the oracle net's blob -- and the tail-fused ones (six names in the dilated template, plus the per-head tops) are held
    to the oracle within 1e-4 (probabilities) resp. 1e-4 relative to the largest logit."""
    gnet, onet = H.make_pair(H.detector_msg(dd), cls_bias=1.0)
    gnet.set_conv_mode(conv_mode)
    before = gnet.blobs["cls_prob_output"].data          # before the first forward: zeros of the declared shape, no error
    assert not before.any()
    data = H.synth_image_blob(80, 112, seed=5)
    info = np.array([[75, 110, 0.7]], np.float32)
    H.run_both(gnet, onet, data, info)
    fused = (["cls_score_1_output", "cls_score_2_output", "cls_score_4_output", "bbox_pred_1_output", "bbox_pred_2_output",
              "bbox_pred_4_output", "cls_score_reshape_output", "cls_prob_output"] if dd else
             ["cls_score_output", "cls_score_reshape_output", "cls_prob_output"])
    for name in fused:
        assert name in gnet.blobs, name
    for name in gnet.blobs.keys():
        a, b = gnet.blobs[name].data, onet.blobs[name].data
        assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
        if name in fused:
            tol = 1e-4 if "prob" in name else 1e-4 * max(1.0, float(np.abs(b).max()))
            assert np.abs(a - b).max() < tol, (name, float(np.abs(a - b).max()))
    # the softmax's top is the memory of the blob the proposal layer reads, re-viewed as (1, 2, A*h, w)
    np.testing.assert_array_equal(gnet.blobs["cls_prob_output"].data.reshape(-1), gnet.blobs["cls_prob_reshape_output"].data.reshape(-1))
    # a second forward at another shape: the read-back follows it
    data2 = H.synth_image_blob(48, 64, seed=6)
    H.run_both(gnet, onet, data2, np.array([[48, 64, 1.0]], np.float32))
    name = fused[0]
    a, b = gnet.blobs[name].data, onet.blobs[name].data
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-4 * max(1.0, float(np.abs(b).max()))
    if conv_mode == "f16x3":
        # a split-fp16 forward runs the fused path's kernels and leaves the intermediates to be materialised on demand from
        # the inputs still on the device: reshaping an input in between is refused with a message, not answered with garbage
        H.run_both(gnet, onet, data2, np.array([[48, 64, 1.0]], np.float32))
        gnet.blobs['data'].reshape(1, 3, 32, 32)
        with pytest.raises(Exception, match="reshaped after the last forward"):
            gnet.blobs["conv3_3"].data
        gnet.blobs['data'].reshape(*data2.shape)
        with pytest.raises(Exception, match="reshaped after the last forward"):     # (its device copy may be gone)
            gnet.blobs["conv3_3"].data
        H.run_both(gnet, onet, data2, np.array([[48, 64, 1.0]], np.float32))
        assert H.rel_err(gnet.blobs["conv3_3"].data, onet.blobs["conv3_3"].data) < ACT_TOL


@pytest.mark.parametrize("bad,msg", [
    ("stride: 0", "stride must be"), ("kernel_size: 0", "kernel_size must be"), ("num_output: 0", "num_output must be"),
    ("num_output: 99999999999", "num_output must be"), ("dilation: 0", "dilation must be"), ("pad: -3", "pad must be"),
    ("group: 3", "group must divide"), ("kernel_size: 4000000000", "kernel_size must be"),
])
def test_hostile_layer_parameters_are_refused_by_name(bad, msg, conv_mode):
    """BaseConvolutionLayer::LayerSetUp's CHECKs (base_conv_layer.cpp:21-120) abort Caffe; here a prototxt with a zero stride,
    a zero / absurd kernel, no or 10^11 outputs ... is refused at caffe.Net() with the layer's name -- no division by zero, no
    2^31-channel allocation, the process lives on (tests/test_parser_robustness.py covers the SYNTAX of hostile files)."""
    if conv_mode != "fp32":
        pytest.skip("graph construction only")
    from smallhardface_amd import caffe
    key = bad.split(":")[0]
    base = {"num_output": "num_output: 8", "kernel_size": "kernel_size: 3", "pad": "pad: 1", "stride": "", "dilation": "", "group": ""}
    base[key] = bad
    txt = ('input: "data" input_shape { dim: 1 dim: 3 dim: 16 dim: 16 }\n'
           'layer { name: "evil" type: "Convolution" bottom: "data" top: "c" convolution_param { %s } }\n' % " ".join(v for v in base.values() if v))
    with pytest.raises(RuntimeError, match=msg):
        caffe.Net(None, prototxt_text=txt)
    for shape, m2 in (("dim: 1 dim: 3 dim: -4 dim: 16", "negative dimension"), ("dim: 65536 dim: 65536 dim: 4", "exceeds INT_MAX")):
        with pytest.raises(RuntimeError, match=m2):
            caffe.Net(None, prototxt_text='input: "data" input_shape { %s }\n' % shape)
    ok = caffe.Net(None, prototxt_text='input: "data" input_shape { dim: 1 dim: 3 dim: 16 dim: 16 }\n'
                   'layer { name: "fine" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 8 kernel_size: 3 pad: 1 } }\n')
    with pytest.raises(Exception, match="INT_MAX|negative"):
        ok.blobs["data"].reshape(70000, 70000)
    with pytest.raises(Exception, match="INT_MAX|negative"):
        ok.blobs["data"].reshape(1, 3, -1, 16)


def test_proposal_edge_cases():
    """all-below-threshold keeps the single best anchor; >10000 candidates are cut at N_DETS_PER_MODULE."""
    msg = H.detector_msg(True)
    for bias, expect in ((12.0, "one"), (-6.0, "topn")):
        gnet, onet = H.make_pair(msg, cls_bias=bias)
        data = H.synth_image_blob(512 if expect == "topn" else 64, 512 if expect == "topn" else 64, seed=1)
        h = data.shape[2]
        info = np.array([[h, h, 1.0]], np.float32)
        for net in (gnet,):
            net.blobs['data'].reshape(*data.shape)
            net.blobs['im_info'].reshape(1, 3)
        go = gnet.forward(data=data, im_info=info)
        gp = gnet.blobs["cls_prob_reshape_output"].data
        gd = gnet.blobs["bbox_pred_output"].data
        pb, pp = O.proposal_forward(gp, gd, info)
        assert go["boxes"].shape == pb.shape
        np.testing.assert_array_equal(go["cls_prob"], pp)
        assert np.abs(go["boxes"] - pb).max() < BOX_TOL
        if expect == "one":
            assert pb.shape[0] == 1 and pp[0, 1] < 0.002
        else:
            assert pb.shape[0] == 10000


VOTE_SETS = ["empty", "single", "two_overlap", "singletons", "last_singleton", "clusters_small",
             "clusters_mid", "clusters_big", "dense", "iou_exact_0p4"]


@pytest.mark.parametrize("name", VOTE_SETS)
def test_bbox_vote_golden(golden, name):
    from smallhardface_amd.nms import bbox_vote
    g = golden("vote_nms.npz")
    out = bbox_vote(g[name + "_dets"], 0.4)
    ref = g[name + "_vote"]
    assert out.shape == ref.shape and out.dtype == np.float64
    np.testing.assert_array_equal(out, ref)   # bit-exact, incl. numpy's summation order


@pytest.mark.parametrize("name", VOTE_SETS)
@pytest.mark.parametrize("thr", [0.4, 0.3, 0.7])
def test_nms_golden(golden, name, thr):
    from smallhardface_amd.nms import nms
    g = golden("vote_nms.npz")
    keep = nms(g[name + "_dets"], thr)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), g[name + "_nms_%02d" % int(thr * 100)])


def test_ties_are_canonical(golden):
    """Equal scores: lower input index first (the oracle's canonical order)."""
    from smallhardface_amd.nms import bbox_vote, nms
    d = golden("vote_nms.npz")["ties_dets"]
    np.testing.assert_array_equal(np.asarray(nms(d, 0.4)), O.nms(d, 0.4))
    np.testing.assert_array_equal(bbox_vote(d, 0.4), np.asarray(O.bbox_vote(d, 0.4), dtype=np.float64))


@pytest.mark.parametrize("n", [20000, 40000])
def test_nms_vote_large(n):
    """Multi-chunk device sort + long scans; checked against the oracle."""
    from smallhardface_amd.nms import bbox_vote, nms
    rng = np.random.default_rng(n)
    c = rng.uniform(0, 3000, (n, 2))
    s = rng.uniform(8, 80, (n, 2))
    d = np.hstack([c, c + s, rng.uniform(0.05, 1, (n, 1))]).astype(np.float32)
    keep = np.asarray(nms(d, 0.4))
    np.testing.assert_array_equal(keep, O.nms(d, 0.4))
    assert np.all(np.diff(d[keep, 4]) <= 0)  # sortedness property
    if n == 20000:
        np.testing.assert_array_equal(bbox_vote(d, 0.4), np.asarray(O.bbox_vote(d, 0.4), dtype=np.float64))
    # idempotence: NMS of the kept set keeps everything
    assert len(nms(d[keep], 0.4)) == len(keep)


def test_detect_driver_vs_fused_vs_oracle(conv_mode):
    """lib/test.py control flow on the GPU net, the fused device path, and the oracle net."""
    from smallhardface_amd import test as T
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    cfg.TEST.SCALES = [100, 300]
    msg = H.detector_msg(True)
    gnet, onet = H.make_pair(msg, cls_bias=1.0)
    im = np.random.default_rng(7).integers(0, 256, (96, 128, 3)).astype(np.uint8)
    for method in ("BBOX_VOTE", "NMS"):
        cfg.TEST.NMS_METHOD = method
        dets, _ = T.detect(gnet, None, thresh=0.05, pyramid=True, im=im)
        fused = T.detect_fused(gnet, T.pyramid_units(im), thresh=0.05)
        # one Net.forward() per unit (lib/test.py's call pattern, host blobs in and out) and the fused device path agree BIT FOR
        # BIT in both arithmetics: in a split-fp16 mode Net.forward() runs the fused path's own kernels (fused first pair,
        # pools in the epilogues, split activation format -- round 6; until then it ran conv1_1 on the vector ALUs and the
        # two paths agreed to 1e-4), and flip fix / unscale / > thresh are the same fp32 operations on the host and on the device
        assert dets[0].shape == fused[0].shape
        np.testing.assert_array_equal(np.asarray(dets[0], dtype=np.float64), fused[0])
        assert dets[0].shape[0] > 0
        # units spread over 3 execution lanes (streams): same detections, same order
        laned = T.FusedDetector(gnet, n_lanes=3).detect(list(T.pyramid_units(im)), thresh=0.05)
        np.testing.assert_array_equal(laned[0], fused[0])
        # ... and as one grouped pass (one grid per conv layer over all units)
        grouped = T.FusedDetector(gnet, n_lanes=1, mode="group").detect(list(T.pyramid_units(im)), thresh=0.05)
        np.testing.assert_array_equal(grouped[0], fused[0])
    # multi-GPU window form: one grouped pass, each lane keeps ITS unit's detections for the gather
    import torch
    units = list(T.pyramid_units(im))
    fdm = T.FusedDetector(gnet, n_lanes=len(units), mode="group")
    fdm.lanes[0].detect_add_levels(fdm.lanes[:len(units)], units, 0.05, per_member_lists=True)
    fdm.lanes[0].sync()
    buf = torch.empty((10000, 5), dtype=torch.float32, device="cuda")
    for m, u in enumerate(units):
        n = fdm.lanes[m].detect_export(buf.data_ptr(), 10000)
        got = buf[:n].cpu().numpy()
        gnet.detect_begin()
        gnet.detect_add_level(*u[:7], 0.05)
        n2 = gnet.detect_export(buf.data_ptr(), 10000)
        np.testing.assert_array_equal(got, buf[:n2].cpu().numpy())
    # the same driver over the oracle net (CPU checker) agrees within tolerance
    cfg.TEST.NMS_METHOD = "NMS"
    gd, _ = T.detect(gnet, None, thresh=0.05, pyramid=True, im=im)
    import smallhardface_amd.test as tm
    old = tm.nms
    tm.nms = lambda d, t: list(O.nms(d, t))
    try:
        od, _ = T.detect(onet, None, thresh=0.05, pyramid=True, im=im)
    finally:
        tm.nms = old
    assert abs(len(gd[0]) - len(od[0])) <= max(2, 0.02 * len(od[0]))
    n = min(len(gd[0]), len(od[0]))
    assert np.abs(gd[0][:n, 4] - od[0][:n, 4]).max() < SCORE_TOL * 5 or n == 0


def test_c1_512_level_vs_oracle(conv_mode):
    """BASELINE config 1 (512x512 level, the reference's CPU-runnable case) against the oracle net:
    every fg/bg score of the 12 288 anchors within 1e-4, deltas within 1e-3, same proposal count."""
    msg = H.detector_msg(True)
    gnet, onet = H.make_pair(msg)
    data = H.synth_image_blob(512, 512, seed=21)
    info = np.array([[512, 512, 1.0]], np.float32)
    go, oo = H.run_both(gnet, onet, data, info)
    gp, op = gnet.blobs["cls_prob_reshape_output"].data, onet.blobs["cls_prob_reshape_output"].data
    assert gp.shape == (1, 6, 64, 64)
    err = float(np.abs(gp - op).max())
    assert err < SCORE_TOL, (conv_mode, err)
    assert np.abs(gnet.blobs["bbox_pred_output"].data - onet.blobs["bbox_pred_output"].data).max() < 1e-3
    assert H.rel_err(gnet.blobs["conv5_3"].data, onet.blobs["conv5_3"].data) < 5e-5
    assert abs(len(go["boxes"]) - len(oo["boxes"])) <= max(2, 0.01 * len(oo["boxes"]))
    n = min(len(go["boxes"]), len(oo["boxes"]))
    assert np.abs(go["cls_prob"][:n, 1] - oo["cls_prob"][:n, 1]).max() < SCORE_TOL


def test_split_fp16_vs_fp32_on_the_bench_pyramid_level():
    """Property at a full-size level (1008x1008, oracle too slow): the two conv arithmetics agree
    far inside the 1e-4 score bar on every anchor."""
    import os
    msg = H.detector_msg(True)
    data = H.synth_image_blob(1008, 1008, seed=5)
    info = np.array([[1000, 1000, 0.9765625]], np.float32)
    res = []
    from smallhardface_amd import weights
    params = weights.synth_params(msg)
    for mode in ("fp32", "f16x3"):
        from smallhardface_amd import caffe
        net = caffe.Net(None, prototxt_text=P.dumps(msg))
        H.load_params(net, params)
        net.set_conv_mode(mode)
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(1, 3)
        out = net.forward(data=data, im_info=info)
        res.append((net.blobs["cls_prob_reshape_output"].data.copy(), out["boxes"].copy(), out["cls_prob"].copy()))
    d = float(np.abs(res[0][0] - res[1][0]).max())
    assert d < 2e-5, d
    assert abs(len(res[0][1]) - len(res[1][1])) <= 2


def test_image_pipeline_submit_collect():
    """Two images in flight over two head lanes: same detections as one-at-a-time."""
    from smallhardface_amd import test as T
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    cfg.TEST.SCALES = [100, 300]
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    ims = [np.random.default_rng(70 + i).integers(0, 256, (96, 128 + 16 * i, 3)).astype(np.uint8) for i in range(3)]
    fd = T.FusedDetector(gnet, n_lanes=4, mode="group")
    ref = [fd.detect(list(T.pyramid_units(im)), thresh=0.05)[0] for im in ims]
    got = []
    for im in ims:
        fd.submit(list(T.pyramid_units(im)), thresh=0.05)
        if fd.pending() > 1:
            got.append(fd.collect()[0])
    while fd.pending():
        got.append(fd.collect()[0])
    assert len(got) == 3
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)


def test_image_pipeline_without_the_shared_conv_stream(tmp_path):
    """SHF_PIPE_SHARED_CONV_STREAM=0 (each head's convolutions on its own stream, hand-over by events):
    the successor head's first convolutions start on the predecessor's `ev_convs` alone, so that event must come after
    the predecessor's tail reset kernel, which zeroes the member lanes' activation-exponent slots.  Images of
    very different magnitudes back to back: a slot zeroed late, or a stale one, changes bits or raises the range flag."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text('''
import sys, numpy as np
from smallhardface_amd.config import cfg
from smallhardface_amd import test as T
from tests import helpers as H
cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
cfg.TEST.SCALES = [100, 300, 500]
gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
gnet.set_conv_mode("f16x3")
fd = T.FusedDetector(gnet, n_lanes=6, mode="group")
units = []
for i in range(8):
    im = np.random.default_rng(40 + i).integers(0, 256, (150 + 10 * (i % 3), 200, 3)).astype(np.uint8)
    us = list(T.pyramid_units(im))
    g = np.float32([1.0, 2.0 ** -9, 2.0, 2.0 ** -4][i % 4])      # units whose exponents differ by up to 10
    units.append([(u[0] * g,) + u[1:] for u in us])
got = []
for rep in range(3):
    for us in units:
        fd.submit(us, thresh=0.05)
        if fd.pending() > 1:
            got.append(fd.collect()[0])
while fd.pending():
    got.append(fd.collect()[0])
assert fd.range_fallbacks == 0 and gnet.range_fallbacks == 0
np.savez(sys.argv[1], *got)
''')
    outs = {}
    for name, env in (("shared", {}), ("own_streams", {"SHF_PIPE_SHARED_CONV_STREAM": "0"})):
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, str(script), out], env=dict(os.environ, PYTHONPATH=root, **env), cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-1500:])
        z = np.load(out)
        outs[name] = [z[k] for k in z.files]
    assert len(outs["shared"]) == 24 and sum(len(a) for a in outs["shared"]) > 0
    for k in range(8, 24):                                   # every repetition reproduces the first
        np.testing.assert_array_equal(outs["shared"][k], outs["shared"][k - 8])
    for a, b in zip(outs["shared"], outs["own_streams"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("shape,scales,flip", [
    ((97, 131), [1.0, 0.5, 1.7, 2.25, 0.3125], True),     # identity, down, up, exact-binary factor
    ((64, 64), [0.25, 3.0], True),                          # already a multiple of MAX_RESOLUTION
    ((33, 250), [0.0625, 1.0 / 3.0, 1.28], False),          # tiny level (3 rows), non-terminating factor
    ((96, 130), [0.5], True),                               # exactly 2x down: cv::resize's INTER_AREA fast path, interior only
    ((97, 130), [0.5], True),                               # 97 -> cvRound(48.5) = 48 rows: the odd row is dropped
    ((99, 135), [0.5], True),                               # 99 -> 50 rows, 135 -> 68 columns: border loop in both axes
])
def test_device_preprocessing_bit_exact(shape, scales, flip, conv_mode):
    """shf_make_pyramid_level == _get_image_blob + flip + pad of the host mirror, bit for bit
    (mean subtraction and interpolation in float64, the blob narrowed to fp32 last)."""
    if conv_mode != "fp32":
        pytest.skip("no convolution in this test")
    import torch
    from smallhardface_amd import test as T
    cfg.TEST.FLIP = flip
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    im = np.random.default_rng(shape[0]).integers(0, 256, shape + (3,)).astype(np.uint8)
    want = list(T.pyramid_units(im, scales))
    dp = T.DevicePyramid(gnet)
    got = dp.units(im, scales)
    gnet.sync()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert tuple(g[1:5]) == tuple(w[1:5]) and g[5] == w[5] and g[6] == w[6]
        n = 3 * g[1] * g[2]
        base = dp._slots[0][0]
        off = (g[0] - base.data_ptr()) // 4
        dev = base[off:off + n].cpu().numpy().reshape(1, 3, g[1], g[2])
        np.testing.assert_array_equal(dev.view(np.uint32), w[0].view(np.uint32))


@pytest.mark.parametrize("shape,scales", [((97, 131), [1.0, 0.5, 1.7, 2.25, 0.3125]), ((99, 135), [0.5, 1.3671875]),
                                          ((33, 250), [0.0625, 1.0 / 3.0, 1.28])])
def test_get_image_blob_on_the_device_equals_the_host_mirror(shape, scales, conv_mode):
    """detect()'s pre-processing step with host blobs out (C ABI shf_image_blobs; the reference calls cv2.resize here) ==
    the numpy mirror _get_image_blob, bit for bit, unpadded and unflipped like the reference's blobs."""
    if conv_mode != "fp32":
        pytest.skip("no convolution in this test")
    from smallhardface_amd import test_utils as TU
    im = np.random.default_rng(shape[1]).integers(0, 256, shape + (3,)).astype(np.uint8)
    want = TU._get_image_blob(im, scales)
    got = TU._get_image_blob_device(im, scales)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g['data'].shape == w['data'].shape and g['data'].dtype == np.float32 and g['data'].flags.writeable
        np.testing.assert_array_equal(g['data'].view(np.uint32), w['data'].view(np.uint32))


def test_two_lane_sets_overlap_mode_gives_the_same_detections(conv_mode):
    """FusedDetector(lane_sets=2) -- bench.py's `overlapped_pipeline` measurement leg: each head lane with a lane set and a
    stream of its own, consecutive images' convolutions overlapping on the GPU -- runs the same kernels on the same data:
    a stream of alternating images comes back bit for bit as from the shipped one-lane-set pipeline."""
    from smallhardface_amd import test as T
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    cfg.TEST.SCALES = [100, 300, 500]
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    gnet.set_conv_mode(conv_mode)
    ims = [np.random.default_rng(70 + i).integers(0, 256, (110 + 9 * i, 150 - 6 * i, 3)).astype(np.uint8) for i in range(3)]
    units = [list(T.pyramid_units(im)) for im in ims]
    outs = {}
    for sets in (1, 2):
        fd = T.FusedDetector(gnet, n_lanes=6, mode="group", lane_sets=sets)
        got = []
        for k in range(7):
            fd.submit(units[k % 3], 0.05)
            if fd.pending() > 1:
                got.append(fd.collect()[0])
        while fd.pending():
            got.append(fd.collect()[0])
        outs[sets] = got
    assert len(outs[1]) == len(outs[2]) == 7 and len(outs[1][0]) > 0
    for a, b in zip(outs[1], outs[2]):
        np.testing.assert_array_equal(a, b)


def test_device_preprocessing_feeds_the_detector():
    """uint8 image -> device pyramid -> grouped detector == host pyramid -> grouped detector."""
    from smallhardface_amd import test as T
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    cfg.TEST.SCALES = [100, 300, 500]
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    fd = T.FusedDetector(gnet, n_lanes=6, mode="group")
    dp = T.DevicePyramid(gnet)
    for i in range(2):
        im = np.random.default_rng(90 + i).integers(0, 256, (120 + 7 * i, 150, 3)).astype(np.uint8)
        ref = fd.detect(list(T.pyramid_units(im)), thresh=0.05)[0]
        got = fd.detect(dp.units(im), thresh=0.05, on_device=True)[0]
        assert len(ref) > 0
        np.testing.assert_array_equal(got, ref)


def test_kernel_selection_knobs_do_not_change_results(tmp_path, conv_mode):
    """Every kernel-selection / data-format knob of the split-fp16 path is a pure performance choice: the
    detections of the fused path are bit-identical with the split activation format switched off and with single /
    two-tile blocks and 8-row tiles forced in the dual-tile family (its activation exponent is a function of the unit
    alone, so no grouping may change a bit); kernels with another accumulation scheme (8-wave two-accumulator, conv1_1
    on the vector ALUs) agree to fp32-class tolerance on the PRE-MERGE rows (a borderline IoU >= 0.4 decision of the
    greedy vote cascades through a whole pile of overlapping boxes, so voted rows are compared only between builds of
    the same arithmetic)."""
    if conv_mode != "f16x3":
        pytest.skip("split-fp16 knobs")
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text('''
import sys, numpy as np, torch
from smallhardface_amd.config import cfg
from smallhardface_amd import test as T
from tests import helpers as H
cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
cfg.TEST.SCALES = [100, 300, 500]
msg = H.detector_msg(True)
gnet, onet = H.make_pair(msg, cls_bias=1.0)
rng = np.random.default_rng(17)                      # biases with some life (the synthetic ones are zero)
for name, blobs in onet.params.items():
    if name.startswith("conv") and len(blobs) > 1:
        blobs[1][...] = rng.normal(0, 0.05, blobs[1].shape).astype(np.float32)
H.load_params(gnet, onet.params)
gnet.set_conv_mode("f16x3")
im = np.random.default_rng(5).integers(0, 256, (150, 200, 3)).astype(np.uint8)
units = list(T.pyramid_units(im))
fd = T.FusedDetector(gnet, n_lanes=6, mode="group")
voted = fd.detect(units, thresh=0.05)[0]
gnet.detect_begin()                                  # the same units once more, one by one: the rows before the merge
for u in units:
    gnet.detect_add_level(*u, thresh=0.05)
buf = torch.empty((400000, 5), dtype=torch.float32, device="cuda")
n = gnet.detect_export(buf.data_ptr(), 400000)
np.savez(sys.argv[1], voted=voted, raw=buf[:n].cpu().numpy())
''')
    outs = {}
    for name, env in (("default", {}), ("no_split_act", {"SHF_F16X3_SPLIT_ACT": "0"}), ("no_w4", {"SHF_F16X3_W4": "0"}),
                      ("single_tile", {"SHF_F16X3_W4D_NTILE": "1"}), ("dual_tile", {"SHF_F16X3_W4D_NTILE": "2"}),
                      ("rows8", {"SHF_F16X3_W4_MT": "2"}), ("rows16", {"SHF_F16X3_W4_MT": "4"}), ("no_pc", {"SHF_F16X3_PC": "0"}),
                      ("pc_no_tile_table", {"SHF_F16X3_PC_TAB": "0"}), ("three_head_launches", {"SHF_F16X3_HEADS3": "0"}),
                      ("pc_block_per_tile", {"SHF_F16X3_PC_PERSIST": "0"})):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ, PYTHONPATH=root, **env)
        r = subprocess.run([sys.executable, str(script), out], env=e, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-1500:])
        outs[name] = np.load(out)
    assert len(outs["default"]["voted"]) > 0 and len(outs["default"]["raw"]) > len(outs["default"]["voted"])
    # same arithmetic, different data path: bit-identical, merged and un-merged
    for name in ("no_split_act", "single_tile", "dual_tile", "rows8", "rows16", "pc_block_per_tile", "pc_no_tile_table", "three_head_launches"):
        for key in ("voted", "raw"):
            assert outs[name][key].shape == outs["default"][key].shape and np.array_equal(outs[name][key], outs["default"][key]), (name, key)
    # other kernels for the same layers: fp32-class agreement of the rows that go into the merge (a row may cross the
    # > 0.05 cut on one side only)
    for name in ("no_w4", "no_pc"):
        a, b = outs["default"]["raw"], outs[name]["raw"]
        slack = max(2, len(a) // 500)
        assert abs(len(a) - len(b)) <= slack, (name, len(a), len(b))
        assert unmatched_rows(a, b) <= slack and unmatched_rows(b, a) <= slack, name


def test_full_bench_pyramid_properties(conv_mode):
    """BASELINE's full-size workload (1024x1024 source, scales 100..1400 x flip = 10 units, 5 TFLOP) through the
    fused path: size-independent properties instead of an oracle run -- determinism (twice, and pipelined vs one
    at a time, and from the raw uint8 image through the device pyramid), score order and range, boxes inside the
    image, and idempotence of the merge (voting the voted boxes again changes nothing but exact duplicates)."""
    if conv_mode != "f16x3":
        pytest.skip("full-size run once, in the benchmarked arithmetic")
    from smallhardface_amd import caffe, nms, test as T, weights
    from smallhardface_amd.config import cfg_from_file
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_from_file(os.path.join(root, "configs", "smallhardface.toml"))
    msg = P._add_dimension_reduction(P.build_test_template(True))
    net = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(net, weights.synth_params(msg, seed=1234))
    net.set_conv_mode("f16x3")
    im = np.random.default_rng(1000).integers(0, 256, (1024, 1024, 3)).astype(np.uint8)
    units = list(T.pyramid_units(im))
    assert len(units) == 10 and max(u[1] for u in units) == 1408
    fd = T.FusedDetector(net, n_lanes=10, mode="group")
    a = fd.detect(units, thresh=0.05)[0]
    b = fd.detect(units, thresh=0.05)[0]
    np.testing.assert_array_equal(a, b)
    dp = T.DevicePyramid(net)
    got = []
    for _ in range(3):
        fd.submit(dp.units(im, net=fd.next_head()), thresh=0.05, on_device=True)
        if fd.pending() > 1:
            got.append(fd.collect()[0])
    while fd.pending():
        got.append(fd.collect()[0])
    for g in got:
        np.testing.assert_array_equal(g, a)
    assert len(a) > 0 and a.shape[1] == 5
    assert np.all(a[:, 4] > 0.05) and np.all(a[:, 4] <= 1.0)
    assert np.all(np.diff(a[:, 4]) <= 0)                       # bbox_vote emits clusters by falling head score
    assert np.all(a[:, 0] <= a[:, 2]) and np.all(a[:, 1] <= a[:, 3])
    assert a[:, :4].min() >= -1e-3 and a[:, [0, 2]].max() <= 1024 and a[:, [1, 3]].max() <= 1024
    again = nms.bbox_vote(a.astype(np.float32), cfg.TEST.NMS_THRESH)
    assert len(again) <= len(a) and np.all(np.diff(again[:, 4]) <= 0)

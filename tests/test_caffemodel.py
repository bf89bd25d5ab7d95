"""``.caffemodel`` wire format: python writer <-> the runtime's C++ reader."""
import numpy as np
import pytest

from smallhardface_amd import caffemodel, weights
from tests import helpers as H


def test_roundtrip_cpu(tmp_path):
    msg = H.detector_msg(True)
    params = weights.synth_params(msg, seed=3)
    p = caffemodel.write_caffemodel(str(tmp_path / "m.caffemodel"), params)
    for name in ("conv1_1", "conv4_3", "head_1", "cls_score_2", "conv5_256_up"):
        for i, arr in enumerate(params[name]):
            got = caffemodel.read_blob(p, name, i)
            assert got.shape == arr.shape
            np.testing.assert_array_equal(got, arr)
    with pytest.raises(Exception, match="no layer named"):
        caffemodel.read_blob(p, "nope", 0)


@pytest.mark.gpu
def test_net_loads_caffemodel(tmp_path):
    """caffe.Net(proto, weights, TEST) copies blobs by layer NAME (net.cpp:733-768)."""
    from smallhardface_amd import caffe, prototxt as P
    msg = H.detector_msg(True)
    params = weights.synth_params(msg, seed=3)
    mp = caffemodel.write_caffemodel(str(tmp_path / "m.caffemodel"), params)
    pp = str(tmp_path / "test.prototxt")
    open(pp, "w").write(P.dumps(msg))
    a = caffe.Net(pp, mp, caffe.TEST)
    b = caffe.Net(pp, None, caffe.TEST)
    H.load_params(b, params)
    np.testing.assert_array_equal(a.params["head_4"][0].data, params["head_1"][0])
    data = H.synth_image_blob(48, 64, seed=2)
    info = np.array([[48, 64, 1.0]], np.float32)
    outs = []
    for net in (a, b):
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(1, 3)
        o = net.forward(data=data, im_info=info)
        outs.append((o["boxes"].copy(), o["cls_prob"].copy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception, match="Could not open|could not open"):
        caffe.Net(pp, str(tmp_path / "missing.caffemodel"), caffe.TEST)


def _caffe_pb2_subset():
    """NetParameter / LayerParameter / V1LayerParameter / BlobProto / BlobShape with the field numbers of
    caffe/src/caffe/proto/caffe.proto:5-22,64-96,306-330,1247-1290, built at run time with the official protobuf
    runtime (no protoc in the image), plus a few extra fields the reader must skip."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="caffe_subset_for_tests.proto", package="caffe_t", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, no, typ, label, tname, packed in fields:
            f = m.field.add(name=fname, number=no, type=typ, label=label)
            if tname:
                f.type_name = ".caffe_t." + tname
            if packed:
                f.options.packed = True

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, REP, None, True)])
    msg("BlobProto", [("shape", 7, F.TYPE_MESSAGE, OPT, "BlobShape", False),
                      ("data", 5, F.TYPE_FLOAT, REP, None, True), ("diff", 6, F.TYPE_FLOAT, REP, None, True),
                      ("num", 1, F.TYPE_INT32, OPT, None, False), ("channels", 2, F.TYPE_INT32, OPT, None, False),
                      ("height", 3, F.TYPE_INT32, OPT, None, False), ("width", 4, F.TYPE_INT32, OPT, None, False)])
    msg("ParamSpec", [("name", 1, F.TYPE_STRING, OPT, None, False), ("lr_mult", 3, F.TYPE_FLOAT, OPT, None, False)])
    msg("LayerParameter", [("name", 1, F.TYPE_STRING, OPT, None, False), ("type", 2, F.TYPE_STRING, OPT, None, False),
                           ("bottom", 3, F.TYPE_STRING, REP, None, False), ("top", 4, F.TYPE_STRING, REP, None, False),
                           ("phase", 10, F.TYPE_INT32, OPT, None, False),
                           ("param", 6, F.TYPE_MESSAGE, REP, "ParamSpec", False),
                           ("blobs", 7, F.TYPE_MESSAGE, REP, "BlobProto", False)])
    msg("V1LayerParameter", [("bottom", 2, F.TYPE_STRING, REP, None, False), ("top", 3, F.TYPE_STRING, REP, None, False),
                             ("name", 4, F.TYPE_STRING, OPT, None, False), ("type", 5, F.TYPE_INT32, OPT, None, False),
                             ("blobs", 6, F.TYPE_MESSAGE, REP, "BlobProto", False),
                             ("blobs_lr", 7, F.TYPE_FLOAT, REP, None, False)])
    msg("NetParameter", [("name", 1, F.TYPE_STRING, OPT, None, False), ("input", 3, F.TYPE_STRING, REP, None, False),
                         ("force_backward", 5, F.TYPE_BOOL, OPT, None, False),
                         ("layers", 2, F.TYPE_MESSAGE, REP, "V1LayerParameter", False),
                         ("layer", 100, F.TYPE_MESSAGE, REP, "LayerParameter", False)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe_t." + n))
            for n in ("NetParameter", "LayerParameter", "V1LayerParameter", "BlobProto")}


def test_reads_models_serialised_by_the_protobuf_runtime(tmp_path):
    """The runtime's wire reader against files written by google.protobuf itself: the modern ``layer`` form with
    BlobShape, the legacy 4-D num/channels/height/width blobs, the V1 ``layers`` form (what published VGG-16
    caffemodels use before upgrade_proto), unknown/extra fields, and Caffe's name-matched copy semantics."""
    pb = _caffe_pb2_subset()
    rng = np.random.default_rng(5)
    w1 = rng.normal(0, 1, (8, 3, 3, 3)).astype(np.float32)
    b1 = rng.normal(0, 1, (8,)).astype(np.float32)
    w2 = rng.normal(0, 1, (4, 8, 1, 1)).astype(np.float32)
    net = pb["NetParameter"](name="n", force_backward=True)
    net.input.append("data")
    l1 = net.layer.add(name="conv_a", type="Convolution", phase=1)
    l1.bottom.append("data"); l1.top.append("conv_a")
    l1.param.add(name="w", lr_mult=0.0)
    for arr in (w1, b1):
        bp = l1.blobs.add()
        bp.shape.dim.extend(arr.shape)
        bp.data.extend(arr.ravel().tolist())
        bp.diff.extend([0.0] * 3)                      # a field the reader skips
    l2 = net.layer.add(name="conv_legacy", type="Convolution")
    bp = l2.blobs.add(num=4, channels=8, height=1, width=1)   # pre-BlobShape blob
    bp.data.extend(w2.ravel().tolist())
    v1 = net.layers.add(name="conv_v1", type=4)                 # V1LayerParameter.CONVOLUTION
    v1.blobs_lr.extend([1.0, 2.0])
    bp = v1.blobs.add(num=1, channels=1, height=1, width=8)    # V1 bias blobs are (1,1,1,N)
    bp.data.extend(b1.tolist())
    path = str(tmp_path / "pb.caffemodel")
    open(path, "wb").write(net.SerializeToString())
    got = caffemodel.read_blob(path, "conv_a", 0)
    assert got.shape == w1.shape
    np.testing.assert_array_equal(got, w1)
    np.testing.assert_array_equal(caffemodel.read_blob(path, "conv_a", 1), b1)
    got = caffemodel.read_blob(path, "conv_legacy", 0)
    assert got.shape == (4, 8, 1, 1)
    np.testing.assert_array_equal(got, w2)
    got = caffemodel.read_blob(path, "conv_v1", 0)
    assert got.shape == (1, 1, 1, 8)
    np.testing.assert_array_equal(got.ravel(), b1)
    # and our own writer's files parse back with the protobuf runtime
    p2 = caffemodel.write_caffemodel(str(tmp_path / "own.caffemodel"), {"conv_a": [w1, b1]})
    back = pb["NetParameter"]()
    back.ParseFromString(open(p2, "rb").read())
    assert back.layer[0].name == "conv_a" and list(back.layer[0].blobs[0].shape.dim) == list(w1.shape)
    np.testing.assert_array_equal(np.array(back.layer[0].blobs[0].data, np.float32).reshape(w1.shape), w1)

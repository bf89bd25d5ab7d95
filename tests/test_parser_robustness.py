"""Hostile input for the two hand-written parsers of the model formats (csrc/proto_text.h: prototxt text format and the
.caffemodel wire format -- there is no protoc / libprotobuf in the image; the formats are
/root/reference/caffe/src/caffe/proto/caffe.proto:10-22 BlobProto, :64-96 NetParameter, LayerParameter{name=1, type=2,
blobs=7}).  Both read files a user hands to ``caffe.Net(prototxt, caffemodel, TEST)`` / ``--amend TEST.MODEL``: they must
answer garbage with their own error (``caffemodel: ...`` / ``prototxt: ...``), never with a crash or an out-of-bounds
read.  The header needs no GPU, so it is compiled here into a small host executable with
``g++ -fsanitize=address,undefined`` (tests/native/parser_harness.cpp copies every input into an exact-size heap buffer
first) and fed:

  * a small valid .caffemodel truncated at EVERY byte boundary;
  * a length varint of 2^63, an 11-byte varint, a ten-byte varint with payload in its top byte;
  * 1 000-deep nested length-delimited messages in an unknown field, wire types 3 / 4 / 6 / 7;
  * a blob whose shape product overflows int, a negative dimension, data that does not match the shape, 40 axes,
    a packed float array of 5 bytes;
  * prototxt with unbalanced braces, unterminated strings and lists, 10^5 nested braces, stray closers, NUL bytes.
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "smallhardface_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("parser_harness") / "parser_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", CSRC, os.path.join(ROOT, "tests", "native", "parser_harness.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def _run(harness, mode, path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([harness, mode, str(path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (mode, str(path), r.returncode, r.stdout[-500:], r.stderr[-3000:])
    return r.stdout.splitlines()


def _v(n):
    from smallhardface_amd.caffemodel import _varint
    return _varint(n)


def _f(no, wire, payload):
    from smallhardface_amd.caffemodel import _field
    return _field(no, wire, payload)


def _small_model(tmp_path):
    from smallhardface_amd.caffemodel import write_caffemodel
    rng = np.random.default_rng(0)
    layers = {"conv_a": [rng.normal(size=(2, 3, 1, 1)).astype(np.float32), rng.normal(size=(2,)).astype(np.float32)],
              "fc_b": [rng.normal(size=(3, 2)).astype(np.float32)]}
    p = tmp_path / "small.caffemodel"
    write_caffemodel(str(p), layers)
    return p, layers


def test_valid_model_parses_and_every_truncation_is_an_error_or_a_shorter_model(harness, tmp_path):
    p, layers = _small_model(tmp_path)
    n_floats = sum(a.size for bl in layers.values() for a in bl)
    assert _run(harness, "model", p) == ["OK layers=2 floats=%d" % n_floats]
    size = os.path.getsize(p)
    assert size < 400
    lines = _run(harness, "truncs", p)
    assert len(lines) == size + 1
    n_err = 0
    for k, line in enumerate(lines):
        n, verdict = line.split(" ", 1)
        assert int(n) == k
        if verdict.startswith("ERR "):
            n_err += 1
            assert verdict.startswith("ERR caffemodel: "), line
        else:
            # a cut exactly between top-level fields leaves a valid, shorter file (protobuf has no end marker)
            assert verdict.startswith("OK layers="), line
            assert int(verdict.split("floats=")[1]) <= n_floats
    assert lines[0].endswith("OK layers=0 floats=0") and lines[-1].endswith("OK layers=2 floats=%d" % n_floats)
    assert n_err >= size - 8


def _blob(shape_dims, data_bytes, packed=True):
    shape = _f(1, 2, b"".join(_v(d) for d in shape_dims)) if packed else b"".join(_f(1, 0, _v(d)) for d in shape_dims)
    return _f(7, 2, shape) + _f(5, 2, data_bytes)


def _model_with(blob_payload):
    return _f(100, 2, _f(1, 2, b"x") + _f(2, 2, b"Convolution") + _f(7, 2, blob_payload))


HOSTILE_MODELS = {
    "length_2_63": (_v((100 << 3) | 2) + _v(1 << 63) + b"abc", "caffemodel: truncated"),
    "length_max_u64": (_v((100 << 3) | 2) + b"\xff" * 9 + b"\x01", "caffemodel: truncated"),
    "varint_11_bytes": (b"\xff" * 10 + b"\x01", "caffemodel: bad varint"),
    "varint_never_ends": (b"\x80" * 5, "caffemodel: bad varint"),
    "varint_10_bytes_top_payload": (_v((100 << 3) | 2) + b"\x80" * 9 + b"\x7f", "caffemodel: truncated"),
    "nested_1000_unknown": (None, None),           # built below: valid wire data, skipped as one length-delimited field
    "wire_type_3_group": (_v((9 << 3) | 3), "caffemodel: unsupported wire type"),
    "wire_type_4": (_v((9 << 3) | 4), "caffemodel: unsupported wire type"),
    "wire_type_6": (_v((9 << 3) | 6) + b"1234", "caffemodel: unsupported wire type"),
    "wire_type_7_in_blob": (_model_with(_v((3 << 3) | 7)), "caffemodel: unsupported wire type"),
    "shape_overflows_int": (_model_with(_blob([65536, 65536, 4], b"")), "caffemodel: blob size exceeds INT_MAX"),
    "shape_overflows_int64": (_model_with(_blob([1 << 40, 1 << 40], b"")), "caffemodel: blob size exceeds INT_MAX"),
    "negative_dim": (_model_with(_blob([(1 << 64) - 2, 3], b"")), "caffemodel: negative blob dimension"),
    "unpacked_negative_dim": (_model_with(_blob([2, (1 << 64) - 1], b"", packed=False)), "caffemodel: negative blob dimension"),
    "data_shorter_than_shape": (_model_with(_blob([2, 3], struct.pack("<5f", 1, 2, 3, 4, 5))),
                                "caffemodel: blob data does not match its shape"),
    "data_longer_than_shape": (_model_with(_blob([2], struct.pack("<3f", 1, 2, 3))),
                               "caffemodel: blob data does not match its shape"),
    "forty_axes": (_model_with(_blob([1] * 40, b"")), "caffemodel: blob with more than 32 axes"),
    "packed_floats_5_bytes": (_model_with(_f(5, 2, b"\x00\x00\x80\x3f\x01")), "caffemodel: truncated"),
    "fixed32_cut": (_model_with(_v((5 << 3) | 5) + b"\x00\x00"), "caffemodel: truncated"),
    "fixed64_cut": (_v((9 << 3) | 1) + b"\x00" * 7, "caffemodel: truncated"),
    "layer_longer_than_file": (_v((100 << 3) | 2) + _v(50) + b"\x0a\x01x", "caffemodel: truncated"),
    "blob_longer_than_layer": (_f(100, 2, _f(1, 2, b"x") + _v((7 << 3) | 2) + _v(99) + b"\x00"), "caffemodel: truncated"),
    "legacy_v1_layer_ok": (_f(2, 2, _f(4, 2, b"old") + _f(6, 2, _f(1, 0, _v(1)) + _f(2, 0, _v(1)) + _f(3, 0, _v(1)) + _f(4, 0, _v(2)) +
                                   _f(5, 2, struct.pack("<2f", 1, 2)))), "OK layers=1 floats=2"),
    "zero_dim_ok": (_model_with(_blob([0, 7], b"")), "OK layers=1 floats=0"),
    "empty_file_ok": (b"", "OK layers=0 floats=0"),
}
_deep = b"\x08\x01"
for _ in range(1000):
    _deep = _f(15, 2, _deep)
HOSTILE_MODELS["nested_1000_unknown"] = (_deep + _model_with(_blob([1], struct.pack("<f", 1.0))), "OK layers=1 floats=1")


@pytest.mark.parametrize("name", sorted(HOSTILE_MODELS))
def test_hostile_caffemodel(harness, tmp_path, name):
    payload, want = HOSTILE_MODELS[name]
    p = tmp_path / (name + ".caffemodel")
    p.write_bytes(payload)
    out = _run(harness, "model", p)
    assert out == [want if want.startswith("OK") else "ERR " + want], (name, out)


HOSTILE_TEXT = {
    "unbalanced_open": ('layer { name: "a" convolution_param { num_output: 3 ', "prototxt: unterminated message"),
    "unbalanced_close": ('layer { name: "a" } }', "prototxt: unexpected character"),
    "angle_open": ('layer < name: "a" ', "prototxt: unterminated message"),
    "unterminated_string": ('name: "abc', "prototxt: unterminated string"),
    "unterminated_string_escape_at_end": ('name: "abc\\', "prototxt: unterminated string"),
    "unterminated_list": ("dim: [1, 2, 3", "prototxt: unterminated list"),
    "list_of_unterminated_string": ('top: ["a", "b', "prototxt: unterminated string"),
    "value_missing": ("name:", "prototxt: value expected after name"),
    "colon_only": (":", "prototxt: unexpected character"),
    "nested_100000": ("a {" * 100000, "prototxt: messages nested deeper than 100"),
    "nested_101_closed": ("a {" * 101 + "}" * 101, "prototxt: messages nested deeper than 100"),
    "nested_100_ok": ("a {" * 100 + "b: 1 " + "}" * 100, "OK fields=101"),
    "nul_bytes": ("name: \x00\x00", "prototxt: unexpected character"),
    "only_comment_ok": ("# nothing here", "OK fields=0"),
    "stray_quote_in_atom": ('name: ab"c', "prototxt: unexpected character"),
    "mismatched_closer": ('layer { name: "a" >', "prototxt: unexpected character"),
}


@pytest.mark.parametrize("name", sorted(HOSTILE_TEXT))
def test_hostile_prototxt(harness, tmp_path, name):
    text, want = HOSTILE_TEXT[name]
    p = tmp_path / (name + ".prototxt")
    p.write_bytes(text.encode("latin-1"))
    out = _run(harness, "text", p)
    assert len(out) == 1
    if want.startswith("OK"):
        assert out[0] == want, (name, out)
    else:
        assert out[0].startswith("ERR " + want), (name, out)


def test_the_shipped_library_reports_the_same_errors(tmp_path):
    """The same header inside libshf_hip.so (no GPU needed for the reader): a hostile file comes back as a Python exception
    carrying the parser's message, the process survives."""
    from smallhardface_amd import _lib, caffemodel
    for name in ("length_2_63", "shape_overflows_int", "data_shorter_than_shape", "varint_11_bytes"):
        payload, want = HOSTILE_MODELS[name]
        p = tmp_path / (name + ".caffemodel")
        p.write_bytes(payload)
        with pytest.raises(_lib.ShfError, match=want.replace("caffemodel: ", "")):
            caffemodel.read_blob(str(p), "x", 0)
    p, layers = _small_model(tmp_path)
    np.testing.assert_array_equal(caffemodel.read_blob(str(p), "fc_b", 0), layers["fc_b"][0])

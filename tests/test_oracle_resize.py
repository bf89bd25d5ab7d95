"""oracle/resize.py -- the independent restatement of ``cv2.resize(.., fx, fy, INTER_LINEAR)`` on a float64 image
(/root/reference/lib/utils/test_utils.py:29-46) -- and what is checked AGAINST it.

a4 / f3 of SURVEY.md 8 stay **parity-unpinned**: OpenCV is not installable here and the reference holds no vector for
its resize, so nothing below compares with a cv2-produced number.  What these tests do establish:

  * the oracle follows the published coefficient rule (the hand-derived tables of tests/test_resize_opencv_rule.py,
    which were computed outside any implementation) and the published border rules -- columns: weight zeroed, rows: the
    two row indices clipped and the weights kept;
  * the PRODUCT's host mirror (smallhardface_amd/test_utils.py) and the device kernel (csrc/pre.hip, -m gpu) agree bit
    for bit with that oracle -- a comparator the product does not share a file or a reading with (until round 5 the
    device kernel was only ever compared with the product's own mirror).
"""
import numpy as np
import pytest

from oracle import resize as R
from tests.test_resize_opencv_rule import TABLES

MEANS = [[[102.9801, 115.9465, 122.7717]]]

# the six shapes of tests/test_gpu_parity.py::test_device_preprocessing_bit_exact, plus two more
SHAPES = [
    ((97, 131), [1.0, 0.5, 1.7, 2.25, 0.3125], True),
    ((64, 64), [0.25, 3.0], True),
    ((33, 250), [0.0625, 1.0 / 3.0, 1.28], False),
    ((96, 130), [0.5], True),
    ((97, 130), [0.5], True),
    ((99, 135), [0.5], True),
    ((120, 90), [1.3671875, 0.9765625], True),      # the bench pyramid's up-scaling factor: clipped rows top and bottom
    ((7, 5), [4.0, 0.6], False),
]


def test_x_table_follows_the_hand_derived_coefficients():
    for f, (n_src, n_dst, rows) in TABLES.items():
        assert R.dsize_of(n_src, f) == n_dst
        tab = R.x_table(n_src, n_dst, f)
        for d, (sx, w0, w1) in rows.items():
            assert tab[d][:3] == (sx, w0, w1), (f, d, tab[d])
        assert all(0 <= t[0] <= n_src - 1 and 0.0 <= t[2] < 1.0 for t in tab)
        assert all(t[3] == (t[0] + 1 >= n_src) for t in tab)


def test_y_table_clips_rows_and_keeps_the_weights():
    """Up-scaling by 1.3671875: destination row 0 lies above source row 0 (sy = -1) and the last one below the last
    source row -- both rows of the blend are the border row, and the weights are those of the unclamped fraction."""
    tab = R.y_table(1024, 1400, 1.3671875)
    r0, r1, b0, b1 = tab[0]
    assert (r0, r1) == (0, 0) and b1 > 0.0 and b0 < 1.0
    assert b0 == R._f32(1.0 - b1)
    r0, r1, b0, b1 = tab[-1]
    assert (r0, r1) == (1023, 1023) and b1 > 0.0
    xt = R.x_table(1024, 1400, 1.3671875)
    assert xt[0][:3] == (0, 1.0, 0.0) and xt[-1][:3] == (1023, 1.0, 0.0)     # the column rule zeroes the weight instead
    for (ra, rb, ba, bb), (sx, a0, a1, _) in zip(tab[1:-1], xt[1:-1]):         # away from the border the two tables agree
        assert (ra, ba, bb) == (sx, a0, a1) and rb == ra + 1


def test_cv_round_is_half_to_even():
    assert [R.cv_round(v) for v in (0.5, 1.5, 2.5, 48.5, 49.5, -0.5)] == [0, 2, 2, 48, 50, 0]
    assert R.dsize_of(97, 0.5) == 48 and R.dsize_of(99, 0.5) == 50 and R.dsize_of(135, 0.5) == 68


def test_area_shortcut_only_at_exactly_two():
    assert R.is_area_fast_2x(0.5, 0.5)
    assert not R.is_area_fast_2x(0.5, 0.25) and not R.is_area_fast_2x(0.5000001, 0.5) and not R.is_area_fast_2x(1.0 / 3, 1.0 / 3)
    im = np.arange(5 * 7 * 1, dtype=np.float64).reshape(5, 7, 1) * 1.25 - 3.0
    out = R.cv_resize_linear_f64(im, 0.5, 0.5)
    assert out.shape == (2, 4, 1)                                   # cvRound(2.5) = 2 rows, cvRound(3.5) = 4 columns
    assert out[0, 0, 0] == (((im[0, 0, 0] + im[0, 1, 0]) + im[1, 0, 0]) + im[1, 1, 0]) * 0.25
    # last column: only source column 6 is inside -> two taps, (float)sum / 2
    assert out[1, 3, 0] == float(np.float32(im[2, 6, 0] + im[3, 6, 0]) / np.float32(2))


@pytest.mark.parametrize("shape,scales,flip", SHAPES)
def test_product_host_mirror_equals_the_oracle(shape, scales, flip):
    """smallhardface_amd.test.pyramid_units (the product's _get_image_blob + flip + pad) == oracle.resize.pyramid_level,
    bit for bit, on the device test's shapes."""
    from smallhardface_amd import test as T
    from smallhardface_amd.config import cfg
    cfg.TEST.FLIP = flip
    im = np.random.default_rng(shape[0]).integers(0, 256, shape + (3,)).astype(np.uint8)
    got = list(T.pyramid_units(im, scales))
    k = 0
    for s in scales:
        for fl in ([False, True] if flip else [False]):
            want, lh, lw = R.pyramid_level(im, s, fl, cfg.PIXEL_MEANS, cfg.MAX_RESOLUTION)
            data, Hh, Ww, ih, iw, gs, gf = got[k]
            k += 1
            assert (Hh, Ww, ih, iw, gs, gf) == (want.shape[2], want.shape[3], lh, lw, s, fl)
            np.testing.assert_array_equal(data.view(np.uint32), want.view(np.uint32))
    assert k == len(got)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,scales,flip", SHAPES)
def test_device_pyramid_level_equals_the_oracle(shape, scales, flip):
    """C ABI shf_make_pyramid_level (csrc/pre.hip) == oracle/resize.py, bit for bit: mean subtraction and interpolation in
    float64, the level narrowed to fp32 last, flip of the unpadded level, zero padding.  (Still parity-UNPINNED against
    cv2 itself -- see the module docstring; this replaces the self-comparison with the product's host mirror.)"""
    from smallhardface_amd import test as T
    from smallhardface_amd.config import cfg
    from tests import helpers as H
    cfg.TEST.FLIP = flip
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    im = np.random.default_rng(shape[0]).integers(0, 256, shape + (3,)).astype(np.uint8)
    dp = T.DevicePyramid(gnet)
    got = dp.units(im, scales)
    gnet.sync()
    base = dp._slots[0][0]
    k = 0
    for s in scales:
        for fl in ([False, True] if flip else [False]):
            want, lh, lw = R.pyramid_level(im, s, fl, cfg.PIXEL_MEANS, cfg.MAX_RESOLUTION)
            ptr, Hh, Ww, ih, iw, gs, gf = got[k]
            k += 1
            assert (Hh, Ww, ih, iw, gs, gf) == (want.shape[2], want.shape[3], lh, lw, s, fl)
            off = (ptr - base.data_ptr()) // 4
            dev = base[off:off + 3 * Hh * Ww].cpu().numpy().reshape(1, 3, Hh, Ww)
            np.testing.assert_array_equal(dev.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("h,w,f", [(64, 96, 0.5), (64, 96, 0.25), (64, 96, 0.75), (64, 96, 1.5), (64, 96, 2.5), (40, 56, 1.375),
                                   (48, 80, 0.625)])
def test_oracle_against_a_third_implementation_of_half_pixel_bilinear(h, w, f):
    """Not a pin (cv2 is the only thing that could pin a4 / f3), a sanity bound: torch's ``interpolate(mode="bilinear",
    align_corners=False, recompute_scale_factor=False)`` is the same half-pixel two-tap filter with ``scale = 1 / f`` computed
    in float64 throughout, so it must agree with the OpenCV restatement up to what OpenCV's FLOAT narrowing of the source
    coordinate and of the weights does: a coordinate near 100 keeps 2^-17, i.e. the fraction is off by up to ~4e-6 and a
    blend of pixels up to 255 apart by up to ~1e-3; where the coordinates are exact in float (scale 2 and 4: also OpenCV's
    2x INTER_AREA shortcut) the two agree to the last double bits.  (Shapes chosen so that floor(n f) == cvRound(n f).)"""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(int(h * 1000 + w + 100 * f))
    im = rng.integers(0, 256, (h, w, 3)).astype(np.float64) - 115.0
    a = R.cv_resize_linear_f64(im, f, f)
    t = torch.from_numpy(im.transpose(2, 0, 1)[None].copy())
    b = F.interpolate(t, scale_factor=f, mode="bilinear", align_corners=False, recompute_scale_factor=False)
    b = b.numpy()[0].transpose(1, 2, 0)
    assert a.shape == b.shape == (R.dsize_of(h, f), R.dsize_of(w, f), 3)
    err = float(np.abs(a - b).max())
    if f in (0.5, 0.25):
        assert err < 1e-10, err
    else:
        assert 0.0 < err < 2e-3, err       # not zero: the float narrowing is really there; small: the same filter

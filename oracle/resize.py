"""CPU oracle for the image pre-processing of one pyramid level -- TEST INFRASTRUCTURE ONLY (imported by tests/,
tests/golden/make_golden.py and nothing in the product path).

What it restates: ``_get_image_blob`` (/root/reference/lib/utils/test_utils.py:29-46) --

    im_copy = im.astype(np.float32, copy=True) - cfg.PIXEL_MEANS        # float64: the means are a float64 array
    im_copy = cv2.resize(im_copy, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)   # skipped when s == 1.0

-- then ``im_list_to_blob`` (lib/utils/blob.py:16-32: HWC -> (1, 3, h, w) float32), the flip of the UNPADDED level
(lib/test.py:149-150 ``im_blobs[i]['data'][..., ::-1]``) and the zero padding to a multiple of MAX_RESOLUTION
(lib/test.py:35-38).

PARITY UNPINNED for the resize itself: the algorithm lives in OpenCV (``opencv-python``, requirements.txt:1,
version unpinned), which is neither under /root/reference nor installable here, and the reference has no test or
golden vector for it.  This file restates the PUBLISHED algorithm of OpenCV 4.x for exactly this call --
``cv::resize`` -> ``hal::resize`` -> ``resizeGeneric_Invoker<HResizeLinear<double,double,float,1,HResizeNoVec>,
VResizeLinear<double,double,float,Cast<double,double>,VResizeNoVec>>`` in modules/imgproc/src/resize.cpp -- and was
written from that description alone, WITHOUT reading or importing the product's host mirror
(smallhardface_amd/test_utils.py), so that the device kernel and its comparator no longer share an author's one
reading of it.  Scalar Python on purpose: every rounding is spelled out (``_f32`` = one IEEE narrowing).

The published rules, as restated below
  * dsize:      ``Size(saturate_cast<int>(src_w * fx), saturate_cast<int>(src_h * fy))``; saturate_cast<int>(double) is
                cvRound = round-half-to-even;
  * scale:      ``scale_x = 1. / fx`` ONCE, in double (NOT src / dst);
  * x table:    ``fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx`` (float arithmetic);
                ``sx < 0 -> fx = 0, sx = 0``; ``sx >= src_w - 1 -> fx = 0, sx = src_w - 1`` (and dx >= xmax: the
                horizontal pass copies ``S[sx] * 1`` instead of filtering); ``alpha = {1.f - fx, fx}`` as floats;
  * y table:    the same three statements for fy / sy, ``beta = {1.f - fy, fy}`` -- but NO clamp of fy: the invoker
                clips the two source ROW INDICES instead (``clip(sy + k, 0, src_h)``, k = 0, 1), so a destination row
                above the first / below the last source row blends a row WITH ITSELF with weights that need not sum to
                exactly 1 in float (unlike the x direction, where the weight is zeroed);
  * filtering:  rows first: ``D[dx] = S[sx] * a0 + S[sx + 1] * a1`` in double (float weights widened), then
                ``dst = S0[x] * b0 + S1[x] * b1``; every product and sum is its own rounding (no FMA contraction in the
                generic, non-vectorised CV_64F instantiation);
  * 2x shortcut: ``iscale = saturate_cast<int>(scale)``; ``is_area_fast = |scale_x - iscale_x| < DBL_EPSILON &&
                |scale_y - iscale_y| < DBL_EPSILON``; INTER_LINEAR becomes INTER_AREA when is_area_fast and both iscale ==
                2 -> ``resizeAreaFast_Invoker<double, double, ResizeAreaFastNoVec>``: ``sum += S[o0] + S[o1] + S[o2] +
                S[o3]`` (row-major offsets, one left-to-right chain), ``D = sum * scale`` with ``float scale = 1.f /
                area`` widened to double; windows that leave the source (dx >= dwidth1, or a row with sy0 + 2 > src_h):
                the in-image taps one by one and ``(float)sum / count`` -- a float division; sy0 >= src_h -> 0.
(An IPP-enabled OpenCV build may route CV_64F INTER_LINEAR through ippiResizeLinear instead, whose arithmetic is not
published: one more reason this stays "unpinned".)
"""
import math
import struct

import numpy as np

DBL_EPSILON = 2.220446049250313e-16


def _f32(x):
    """One IEEE-754 narrowing double -> float (round to nearest even), returned as a Python float."""
    return struct.unpack("<f", struct.pack("<f", x))[0]


def cv_round(x):
    """cvRound / saturate_cast<int>(double): lrint in the default rounding mode = round half to even."""
    return int(round(x))      # Python's round() on a float is round-half-to-even


def dsize_of(src, f):
    return cv_round(src * f)


def _index_and_fraction(d, scale):
    """``fx = (float)((d + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx`` -> (sx, fx), fx a float32 value."""
    fx = _f32((d + 0.5) * scale - 0.5)
    sx = int(math.floor(fx))
    fx = _f32(fx - float(sx))          # float - (float)int: exact operands, one float rounding
    return sx, fx


def x_table(n_src, n_dst, f):
    """Per destination column: (sx, a0, a1, copy) -- ``copy``: dx >= xmax, the pass writes S[sx] * 1."""
    scale = 1.0 / f
    out = []
    for dx in range(n_dst):
        sx, fx = _index_and_fraction(dx, scale)
        if sx < 0:
            fx, sx = 0.0, 0
        copy = sx + 1 >= n_src
        if sx >= n_src - 1:
            fx, sx = 0.0, n_src - 1
        out.append((sx, _f32(1.0 - fx), fx, copy))
    return out


def y_table(n_src, n_dst, f):
    """Per destination row: (row0, row1, b0, b1) with the ROW INDICES clipped to [0, n_src) and the weights left alone."""
    scale = 1.0 / f
    clip = lambda v: 0 if v < 0 else (v if v < n_src else n_src - 1)
    out = []
    for dy in range(n_dst):
        sy, fy = _index_and_fraction(dy, scale)
        out.append((clip(sy), clip(sy + 1), _f32(1.0 - fy), fy))
    return out


def is_area_fast_2x(fx, fy):
    sx, sy = 1.0 / fx, 1.0 / fy
    ix, iy = cv_round(sx), cv_round(sy)
    return abs(sx - ix) < DBL_EPSILON and abs(sy - iy) < DBL_EPSILON and ix == 2 and iy == 2


def _area_fast_2x(im, nh, nw):
    h, w, cn = im.shape
    out = np.zeros((nh, nw, cn), np.float64)
    dwidth1 = w // 2
    quarter = float(np.float32(1.0) / np.float32(4))        # float scale = 1.f / area, widened
    for dy in range(nh):
        sy0 = 2 * dy
        if sy0 >= h:
            continue                                         # D[dx] = 0
        wfull = dwidth1 if sy0 + 2 <= h else 0
        for dx in range(nw):
            sx0 = 2 * dx
            for c in range(cn):
                if dx < wfull:
                    s = 0.0
                    s += ((float(im[sy0, sx0, c]) + float(im[sy0, sx0 + 1, c])) + float(im[sy0 + 1, sx0, c])) + float(im[sy0 + 1, sx0 + 1, c])
                    out[dy, dx, c] = s * quarter
                else:
                    s, count = 0.0, 0
                    for ky in range(2):
                        if sy0 + ky >= h:
                            break
                        for kx in range(2):
                            if sx0 + kx >= w:
                                break
                            s += float(im[sy0 + ky, sx0 + kx, c])
                            count += 1
                    out[dy, dx, c] = float(np.float32(s) / np.float32(count)) if count else 0.0
    return out


def cv_resize_linear_f64(im, fx, fy):
    """``cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR)`` of an (h, w, c) float64 image."""
    im = np.asarray(im, dtype=np.float64)
    h, w, cn = im.shape
    nh, nw = dsize_of(h, fy), dsize_of(w, fx)
    if is_area_fast_2x(fx, fy):
        return _area_fast_2x(im, nh, nw)
    xt = x_table(w, nw, fx)
    yt = y_table(h, nh, fy)
    # horizontal pass of every source row that is needed (the invoker keeps two at a time; the values are the same)
    need = sorted(set(r for r0, r1, _, _ in yt for r in (r0, r1)))
    hrow = {}
    for r in need:
        row = np.empty((nw, cn), np.float64)
        for dx, (sx, a0, a1, copy) in enumerate(xt):
            for c in range(cn):
                if copy:
                    row[dx, c] = float(im[r, sx, c]) * 1.0
                else:
                    row[dx, c] = float(im[r, sx, c]) * a0 + float(im[r, sx + 1, c]) * a1
        hrow[r] = row
    out = np.empty((nh, nw, cn), np.float64)
    for dy, (r0, r1, b0, b1) in enumerate(yt):
        out[dy] = hrow[r0] * b0 + hrow[r1] * b1              # elementwise: product, product, sum -- three roundings
    return out


def pyramid_level(im_u8, scale, flip, pixel_means, max_resolution=16):
    """uint8 BGR (h, w, 3) -> the (1, 3, H, W) float32 blob forward_net hands the net for one (scale, flip) unit, and
    the unpadded level size: mean subtraction, resize (skipped at scale == 1.0), HWC -> CHW, flip of the unpadded
    level, zero padding right / bottom to multiples of ``max_resolution``."""
    x = im_u8.astype(np.float32) - np.asarray(pixel_means, dtype=np.float64).reshape(1, 1, 3)   # float64
    if scale != 1.0:
        x = cv_resize_linear_f64(x, scale, scale)
    lvl = x.astype(np.float32).transpose(2, 0, 1)            # im_list_to_blob: float32 blob
    if flip:
        lvl = lvl[:, :, ::-1]
    lh, lw = lvl.shape[1:]
    H = -(-lh // max_resolution) * max_resolution
    W = -(-lw // max_resolution) * max_resolution
    out = np.zeros((1, 3, H, W), np.float32)
    out[0, :, :lh, :lw] = lvl
    return out, lh, lw

"""Image pre-processing helpers for the pyramid driver.

Mirrors /root/reference/lib/utils/test_utils.py:8-46 (``_compute_scaling_factor``,
``_get_image_blob``) and lib/utils/blob.py:16-32 (``im_list_to_blob``).

The reference resizes with ``cv2.resize(im, None, None, fx=s, fy=s, interpolation=INTER_LINEAR)`` on a float64
image (test_utils.py:43-44).  OpenCV is not in this image and no reference test covers it, so resize parity stays
UNPINNED -- but ``resize_bilinear`` is not an arbitrary bilinear filter: it restates, operation for
operation, what OpenCV 4.x publishes for this call (modules/imgproc/src/resize.cpp; line numbers cannot be checked
offline, the functions are named instead):

* ``cv::resize``: with an empty dsize, ``dsize = (cvRound(src_w * fx), cvRound(src_h * fy))`` and the given factors
  are used as they are (``inv_scale = fx``, NOT dst / src);
* ``hal::resize`` coefficient loop: ``scale = 1. / inv_scale`` once, in double; per destination index
  ``fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx;`` -- the source coordinate is narrowed to
  FLOAT before the floor is subtracted, in float; ``sx < 0 -> (fx, sx) = (0, 0)``; ``sx >= src_w - 1 -> (fx, sx) =
  (0, src_w - 1)``; ``cbuf[0] = 1.f - fx; cbuf[1] = fx`` (a float subtraction);
* the VERTICAL table has no such border rule (``_row_coeffs``, round 6 -- found by restating the call a second time,
  independently, as oracle/resize.py): the invoker clips the two source row indices and keeps ``beta = {1.f - fy, fy}``;
* ``HResizeLinear<double, double, float, 1>`` / ``VResizeLinear<double, double, float>`` (the CV_64F instantiation):
  ``S[sx] * a0 + S[sx + cn] * a1`` with the float coefficients widened to double, rows first, then
  ``S0[x] * b0 + S1[x] * b1``; products and sums are separate roundings.


One exception to the two-tap path, also restated (round 4): ``cv::resize`` turns INTER_LINEAR into INTER_AREA when both
scale factors are exactly 2x down -- ``iscale = saturate_cast<int>(scale)`` (= cvRound), ``is_area_fast = |scale_x -
iscale_x| < DBL_EPSILON && |scale_y - iscale_y| < DBL_EPSILON``, ``if (interpolation == INTER_LINEAR && is_area_fast &&
iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`` -- and then runs ``resizeAreaFast_Invoker<double, double,
ResizeAreaFastNoVec>``: interior outputs are ``(S[y][x] + S[y][x+1] + S[y+1][x] + S[y+1][x+1]) * (double)0.25f`` summed
left to right in one chain (the unrolled ``sum += S[ofs[0]] + S[ofs[1]] + S[ofs[2]] + S[ofs[3]]``, offsets row-major);
outputs whose 2x2 window leaves an odd-sized source (the last column when the width is odd and rounds up, every output
of the last row when the height is) take the border loop: the in-image taps added one by one, row-major, and
``(float)sum / count`` -- a FLOAT division, widened back to double.  A level scale of exactly 0.5 is reachable
(TEST.SCALES 600 on a 1200-short-side image).

The benchmarks and GPU parity tests start from blobs *after* the resize; csrc/pre.hip is kept bit-equal to this file.
"""
import numpy as np

from .config import cfg


def _compute_scaling_factor(im_shape, target_size, max_size):
    """lib/utils/test_utils.py:8-26."""
    if cfg.TEST.ORIG_SIZE:
        return 1.0
    im_size_min = np.min(im_shape[0:2])
    im_size_max = np.max(im_shape[0:2])
    im_scale = float(target_size) / float(im_size_min)
    # Prevent the biggest axis from being more than MAX_SIZE
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    return im_scale


def pyramid_scales(im_shape):
    """lib/test.py:131-137: the per-level scales of the test pyramid."""
    base_scale = _compute_scaling_factor(im_shape, cfg.TEST.PYRAMID_BASE_SIZE[0],
                                         cfg.TEST.PYRAMID_BASE_SIZE[1])
    return [float(scale) / cfg.TEST.PYRAMID_BASE_SIZE[0] * base_scale
            for scale in cfg.TEST.SCALES]


def _axis_coeffs(n_src, n_dst, f):
    """OpenCV's INTER_LINEAR coefficient table for one axis (hal::resize, see the module docstring):
    -> (i0, i1, a0, a1) with float32 weights a0 = 1.f - fx, a1 = fx."""
    scale = 1.0 / float(f)                                    # double scale_x = 1. / inv_scale_x
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)         # fx = (float)((dx + 0.5) * scale_x - 0.5)
    i0 = np.floor(fx).astype(np.int64)                        # sx = cvFloor(fx)
    fx = fx - i0.astype(np.float32)                           # fx -= sx   (float arithmetic)
    lo = i0 < 0                                               # if (sx < 0) fx = 0, sx = 0
    i0[lo] = 0
    fx[lo] = 0.0
    hi = i0 >= n_src - 1                                      # if (sx >= src_width - 1) fx = 0, sx = src_width - 1
    i0[hi] = n_src - 1
    fx[hi] = 0.0
    i1 = np.minimum(i0 + 1, n_src - 1)
    a0 = (np.float32(1.0) - fx).astype(np.float32)            # cbuf[0] = 1.f - fx
    return i0, i1, a0, fx.astype(np.float32)


def _row_coeffs(n_src, n_dst, f):
    """The VERTICAL table of the same call (round 6): hal::resize forms ``fy = (float)((dy + 0.5) * scale_y - 0.5); sy =
    cvFloor(fy); fy -= sy; beta = {1.f - fy, fy}`` WITHOUT the horizontal table's border rule -- instead
    resizeGeneric_Invoker clips the two source ROW INDICES, ``clip(sy + k, 0, src_h)`` for k = 0, 1, so a destination row
    above the first / below the last source row blends that row with itself under weights that need not add up to
    exactly 1 in float: ``S * b0 + S * b1`` is not always ``S`` in the last double bit.  -> (i0, i1, b0, b1)."""
    scale = 1.0 / float(f)
    d = np.arange(n_dst, dtype=np.float64)
    fy = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(np.float32)).astype(np.float32)
    i0 = np.clip(sy, 0, n_src - 1)
    i1 = np.clip(sy + 1, 0, n_src - 1)
    return i0, i1, (np.float32(1.0) - fy).astype(np.float32), fy


def is_area_fast_2x(fx, fy):
    """cv::resize's switch from INTER_LINEAR to the INTER_AREA fast path (module docstring): both scales exactly 2x down."""
    eps = np.finfo(np.float64).eps
    sx, sy = 1.0 / float(fx), 1.0 / float(fy)
    ix, iy = int(np.round(sx)), int(np.round(sy))             # saturate_cast<int>(double) = cvRound
    return abs(sx - ix) < eps and abs(sy - iy) < eps and ix == 2 and iy == 2


def _resize_area_fast_2x(im, nh, nw):
    """resizeAreaFast_Invoker<double, double, NoVec> with scale_x = scale_y = 2 (module docstring)."""
    h, w = im.shape[:2]
    out = np.zeros((nh, nw) + im.shape[2:], dtype=im.dtype)
    full_w = w // 2                                           # dwidth1 / cn
    full_h = h // 2                                           # rows with sy0 + 2 <= ssize.height
    fh, fw = min(full_h, nh), min(full_w, nw)
    a = im[0:2 * fh:2, 0:2 * fw:2]
    b = im[0:2 * fh:2, 1:2 * fw:2]
    c = im[1:2 * fh:2, 0:2 * fw:2]
    d = im[1:2 * fh:2, 1:2 * fw:2]
    out[:fh, :fw] = (((a + b) + c) + d) * im.dtype.type(np.float32(0.25))
    for dy in range(nh):
        for dx in (range(nw) if dy >= full_h else range(full_w, nw)):
            sy0, sx0 = 2 * dy, 2 * dx
            if sy0 >= h:
                continue                                      # D[dx] = 0
            s = np.zeros(im.shape[2:], dtype=im.dtype)
            count = 0
            for sy in range(2):
                if sy0 + sy >= h:
                    break
                for sx in range(2):
                    if sx0 + sx >= w:
                        break
                    s = s + im[sy0 + sy, sx0 + sx]
                    count += 1
            if count:
                out[dy, dx] = (s.astype(np.float32) / np.float32(count)).astype(im.dtype)   # (float)sum / count
    return out


def resize_bilinear(im, fx, fy):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) of an HxWxC float image, restated
    (module docstring).  Computes in the image's own dtype: the reference hands cv2 a float64 image
    (uint8.astype(f32) - float64 PIXEL_MEANS -> f64, test_utils.py:36)."""
    im = np.asarray(im)
    if im.dtype not in (np.float32, np.float64):
        im = im.astype(np.float32)
    h, w = im.shape[:2]
    nh = int(np.round(h * fy))  # cvRound: round-half-to-even, same as np.round
    nw = int(np.round(w * fx))
    if is_area_fast_2x(fx, fy):
        return _resize_area_fast_2x(im, nh, nw)
    y0, y1, b0, b1 = _row_coeffs(h, nh, fy)
    x0, x1, a0, a1 = _axis_coeffs(w, nw, fx)
    a0 = a0[None, :, None].astype(im.dtype)
    a1 = a1[None, :, None].astype(im.dtype)
    b0 = b0[:, None, None].astype(im.dtype)
    b1 = b1[:, None, None].astype(im.dtype)
    top = im[y0][:, x0] * a0 + im[y0][:, x1] * a1             # HResizeLinear on source row sy
    bot = im[y1][:, x0] * a0 + im[y1][:, x1] * a1             # ... and sy + 1 (replicated at the border)
    return top * b0 + bot * b1                                # VResizeLinear


def im_list_to_blob(ims):
    """lib/utils/blob.py:16-32: list of HxWx3 images -> (N,3,Hmax,Wmax) fp32."""
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    num_images = len(ims)
    blob = np.zeros((num_images, max_shape[0], max_shape[1], ims[0].shape[2]),
                    dtype=np.float32)
    for i in range(num_images):
        im = ims[i]
        blob[i, 0:im.shape[0], 0:im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2))


def _get_image_blob(im, im_scales):
    """lib/utils/test_utils.py:29-46: mean-subtract, resize per scale, NCHW."""
    im_copy = im.astype(np.float32, copy=True) - np.array(cfg.PIXEL_MEANS)
    blobs = []
    for scale in im_scales:
        if scale == 1.0:
            blobs.append({'data': im_list_to_blob([im_copy])})
        else:
            blobs.append({'data': im_list_to_blob(
                [resize_bilinear(im_copy, scale, scale)])})
    return blobs


def _get_image_blob_device(im, im_scales):
    """``_get_image_blob`` with the arithmetic on the GPU (C ABI shf_image_blobs -> csrc/pre.hip, bit-equal to the host
    mirror above: tests/test_gpu_parity.py) and HOST blobs out, for the literal drop-in path -- ``detect()`` /
    ``forward_net`` with one Net.forward() per unit (lib/test.py:109-178).  The reference spends this step inside a native
    library too (cv2.resize); the numpy mirror takes ~190 ms for a 1024 x 1024 image's five levels, this ~10 ms."""
    import ctypes as C
    from . import _lib, caffe
    lib = _lib.load()
    im = np.ascontiguousarray(im, dtype=np.uint8)
    h, w = im.shape[:2]
    n = len(im_scales)
    shapes = [caffe.pyramid_level_shape(h, w, s, 1)[:2] for s in im_scales]
    outs = [np.empty((1, 3, lh, lw), dtype=np.float32) for lh, lw in shapes]
    pm = (C.c_double * 3)(*[float(v) for v in np.asarray(cfg.PIXEL_MEANS).reshape(-1)[:3]])
    sc = (C.c_double * n)(*[float(s) for s in im_scales])
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    lh = (C.c_int * n)(*[s_[0] for s_ in shapes])
    lw = (C.c_int * n)(*[s_[1] for s_ in shapes])
    _lib.check(lib.shf_image_blobs(im.ctypes.data_as(C.c_void_p), h, w, n, sc, pm, ptrs, lh, lw), "image_blobs")
    return [{'data': o} for o in outs]

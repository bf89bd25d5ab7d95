"""Image pre-processing helpers for the pyramid driver.

Mirrors /root/reference/lib/utils/test_utils.py:8-46 (``_compute_scaling_factor``,
``_get_image_blob``) and lib/utils/blob.py:16-32 (``im_list_to_blob``).

The reference resizes with ``cv2.resize(..., INTER_LINEAR)``; OpenCV is not in
this image and no reference test covers it, so ``resize_bilinear`` below is our
own restatement of OpenCV's documented INTER_LINEAR rule (half-pixel centres,
``dsize = round(src * f)``, source coordinate ``(d + 0.5) / f - 0.5``, replicated
border).  Resize parity is therefore UNPINNED; the benchmarks and
GPU parity tests start from blobs *after* the resize.
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
    d = np.arange(n_dst, dtype=np.float64)
    s = (d + 0.5) / f - 0.5
    i0 = np.floor(s).astype(np.int64)
    frac = (s - i0).astype(np.float32)
    # replicate border the way OpenCV does: clamp the index, zero the weight
    lo = i0 < 0
    i0[lo] = 0
    frac[lo] = 0.0
    hi = i0 >= n_src - 1
    i0[hi] = n_src - 1
    frac[hi] = 0.0
    i1 = np.minimum(i0 + 1, n_src - 1)
    return i0, i1, frac


def resize_bilinear(im, fx, fy):
    """Bilinear resize of an HxWxC float image by factors (fx, fy).

    Computes in the image's own dtype: the reference hands cv2 a float64 image
    (uint8.astype(f32) - float64 PIXEL_MEANS -> f64, test_utils.py:36)."""
    im = np.asarray(im)
    if im.dtype not in (np.float32, np.float64):
        im = im.astype(np.float32)
    h, w = im.shape[:2]
    nh = int(np.round(h * fy))  # cvRound: round-half-to-even, same as np.round
    nw = int(np.round(w * fx))
    y0, y1, wy = _axis_coeffs(h, nh, fy)
    x0, x1, wx = _axis_coeffs(w, nw, fx)
    wx = wx[None, :, None].astype(im.dtype)
    wy = wy[:, None, None].astype(im.dtype)
    one = im.dtype.type(1)
    top = im[y0][:, x0] * (one - wx) + im[y0][:, x1] * wx
    bot = im[y1][:, x0] * (one - wx) + im[y1][:, x1] * wx
    return top * (one - wy) + bot * wy


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

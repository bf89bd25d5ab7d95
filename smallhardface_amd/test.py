"""Multi-GPU multi-scale inference driver.

Mirrors /root/reference/lib/test.py (same function names, arguments, return values
and error behaviour) on top of the pycaffe-compatible shim:

  forward_net       test.py:21-106   pad to MAX_RESOLUTION, im_info, forward, flip fix, unscale, tile
  detect            test.py:109-178  pyramid x flip loop, >thresh cut, bbox_vote / NMS
  bbox_vote         test.py:181-217  (device implementation, see nms.py)
  inference_worker  test.py:220-267  one worker per GPU
  test_net          test.py:290-356  shard the image range over cfg.TEST.GPU_ID, gather

plus ``detect_fused`` -- the same computation with the pyramid resident in HBM and all
per-unit post-processing on the device (C ABI shf_detect_*), which is what bench.py times.
The reference's ``"NMS"`` branch calls a name it never imports; here it
is wired to lib/nms's GPU semantics.
"""
from __future__ import print_function

import logging
import os
import pickle
import sys

import numpy as np

from . import _lib, caffe
from .config import cfg
from .nms import bbox_vote, nms
from .test_utils import _compute_scaling_factor, _get_image_blob, _get_image_blob_device, pyramid_scales
from .timer import Timer

logger = logging.getLogger(__name__)


def _imread(path):
    """cv2.imread stand-in (BGR uint8) -- OpenCV is not in the image."""
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()


def forward_net(net, blob, im_scale, pyramid=False, flip=False):
    """Run one (scale, flip) unit; returns ([probs (R,2)], [pred_boxes (R,8)])."""
    blob['im_info'] = np.array([[blob['data'].shape[2], blob['data'].shape[3], im_scale]], dtype=np.float32)

    h, w = blob['data'].shape[2:]
    new_h = int(np.ceil(1.0 * h / cfg.MAX_RESOLUTION) * cfg.MAX_RESOLUTION)
    new_w = int(np.ceil(1.0 * w / cfg.MAX_RESOLUTION) * cfg.MAX_RESOLUTION)
    n, c = blob['data'].shape[:2]
    net.blobs['data'].reshape(n, c, new_h, new_w)
    net.blobs['im_info'].reshape(*(blob['im_info'].shape))
    # test.py:35-38 zero-pads with np.pad and hands the copy to forward(), which copies it again into the blob: here the
    # (possibly flipped, i.e. negatively strided) level is written ONCE, straight into the blob's host mirror, the pad rows /
    # columns zeroed around it -- the same (n, c, new_h, new_w) tensor, two 24-MB copies per 1408 x 1408 unit fewer
    data = net.blobs['data'].data
    data[:, :, :h, :w] = blob['data']
    if new_h > h:
        data[:, :, h:, :] = 0
    if new_w > w:
        data[:, :, :h, w:] = 0
    net_args = {'data': data,
                'im_info': blob['im_info'].astype(np.float32, copy=False)}
    blobs_out = net.forward(**net_args)

    if flip:
        for i in [k for k in blobs_out.keys() if k.startswith('boxes')]:
            blobs_out[i][:, [1, 3]] = w - blobs_out[i][:, [3, 1]]

    pred_boxes = []
    probs = []
    if 'boxes' in net.blobs:
        if not pyramid:
            raise NotImplementedError("Please complete this part!")  # test.py:84-88
        levels = [None]
    else:
        levels = [k.split('_')[-1] for k in net.blobs.keys() if k.startswith('boxes')]
        if len(cfg.TEST.LEVEL) > 0:
            logger.warning('Subset of levels selected for evaluation: {}'.format(cfg.TEST.LEVEL))
            levels = cfg.TEST.LEVEL
    for level in levels:
        suffix = '' if level is None else '_{}'.format(level)
        cur_boxes = net.blobs['boxes' + suffix].data
        cur_boxes = cur_boxes[:, 1:5] / im_scale  # back to raw image space
        cur_probs = net.blobs['cls_prob' + suffix].data
        pred_boxes.append(np.tile(cur_boxes, (1, cur_probs.shape[1])))
        probs.append(cur_probs)
    return probs, pred_boxes


def _merge_class_dets(probs, boxes, thresh):
    """The per-class >thresh cut and box merging of detect() (test.py:161-176)."""
    cls_dets = [None] * (probs.shape[1] - 1)
    for class_i in range(1, probs.shape[1]):
        inds = np.where(probs[:, class_i] > thresh)[0]
        probs_i = probs[inds, class_i]
        boxes_i = boxes[inds, :]
        dets = np.hstack((boxes_i, probs_i[:, np.newaxis])).astype(np.float32, copy=False)
        if cfg.TEST.NMS_METHOD == "BBOX_VOTE":
            cls_dets[class_i - 1] = bbox_vote(dets)
        elif cfg.TEST.NMS_METHOD == "NMS":
            keep = nms(dets, cfg.TEST.NMS_THRESH)
            cls_dets[class_i - 1] = dets[keep, :]
        else:
            raise NotImplementedError("Unknown NMS method: {}".format(cfg.TEST.NMS_METHOD))
    assert all([_ is not None for _ in cls_dets]), 'None in detection results'
    return cls_dets


def detect(net, im_path, thresh=0.05, timers=None, pyramid=False, im=None):
    if not timers:
        timers = {'detect': Timer(), 'misc': Timer()}
    if im is None:
        im = _imread(im_path)
    sys.stdout.flush()
    timers['detect'].tic()

    if not pyramid:
        im_scale = _compute_scaling_factor(im.shape, cfg.TEST.SCALES[0], cfg.TEST.MAX_SIZE)
        im_blob = _get_image_blob(im, [im_scale])
        probs, boxes = forward_net(net, im_blob[0], im_scale, pyramid=False)
        if isinstance(probs, list):
            probs = np.vstack(probs)
            boxes = np.vstack(boxes)
        boxes = boxes[:, 0:4]
    else:
        all_probs = []
        all_boxes = []
        base_scale = _compute_scaling_factor(im.shape, cfg.TEST.PYRAMID_BASE_SIZE[0],
                                             cfg.TEST.PYRAMID_BASE_SIZE[1])
        pyramid_scales = [float(scale) / cfg.TEST.PYRAMID_BASE_SIZE[0] * base_scale for scale in cfg.TEST.SCALES]
        # (the reference: cv2.resize per level, a native library; here the same step on the GPU with host blobs out -- bit-equal
        # to the numpy mirror _get_image_blob, which SHF_HOST_PREPROCESS=1 selects)
        im_blobs = (_get_image_blob if os.environ.get("SHF_HOST_PREPROCESS") == "1" or im.dtype != np.uint8 else
                    _get_image_blob_device)(im, pyramid_scales)
        for i in range(len(pyramid_scales)):
            probs, boxes = forward_net(net, im_blobs[i], pyramid_scales[i], pyramid=True)
            for j in range(len(probs)):
                all_boxes.append(boxes[j][:, 0:4])
                all_probs.append(probs[j].copy())
            if cfg.TEST.FLIP:
                probs, boxes = forward_net(net, {'data': im_blobs[i]['data'][..., ::-1]}, pyramid_scales[i],
                                           pyramid=True, flip=True)
                for j in range(len(probs)):
                    all_boxes.append(boxes[j][:, 0:4])
                    all_probs.append(probs[j].copy())
        probs = np.concatenate(all_probs)
        boxes = np.concatenate(all_boxes)
    timers['detect'].toc()
    timers['misc'].tic()
    cls_dets = _merge_class_dets(probs, boxes, thresh)
    timers['misc'].toc()
    return cls_dets, timers


def pyramid_units(im, scales=None):
    """The (blob, scale, flip) units detect() feeds the net for one image, padded to
    MAX_RESOLUTION (test.py:35-38,141-155).  Yields (data (1,3,H,W) f32, H, W, im_h, im_w, scale, flip)."""
    if scales is None:
        base_scale = _compute_scaling_factor(im.shape, cfg.TEST.PYRAMID_BASE_SIZE[0],
                                             cfg.TEST.PYRAMID_BASE_SIZE[1])
        scales = [float(scale) / cfg.TEST.PYRAMID_BASE_SIZE[0] * base_scale for scale in cfg.TEST.SCALES]
    im_blobs = _get_image_blob(im, scales)
    m = cfg.MAX_RESOLUTION
    for blob, s in zip(im_blobs, scales):
        for flip in ([False, True] if cfg.TEST.FLIP else [False]):
            d = blob['data'][..., ::-1] if flip else blob['data']
            h, w = d.shape[2:]
            nh, nw = int(np.ceil(1.0 * h / m) * m), int(np.ceil(1.0 * w / m) * m)
            d = np.pad(d, ((0, 0), (0, 0), (0, nh - h), (0, nw - w)), 'constant')
            yield np.ascontiguousarray(d, dtype=np.float32), nh, nw, h, w, s, flip


class DevicePyramid(object):
    """pyramid_units with the pre-processing on the device (C ABI shf_make_pyramid_level): the raw
    BGR uint8 image goes up once (3 bytes/pixel instead of 4 bytes x 3 channels x every level x
    flip) and each (scale, flip) unit is resized / flipped / padded into a device blob on
    ``net``'s stream.  ``units(im)`` returns the same tuples as pyramid_units with a device pointer
    in place of the host array (use with on_device=True); the blobs stay alive in this object
    until the slot comes round again (``n_slots`` calls later: collect that image first)."""

    def __init__(self, net, n_slots=2):
        self.net = net
        self._slots = [None] * max(1, n_slots)
        self._k = 0

    def units(self, im, scales=None, im_dev=None, net=None):
        return self.window_units([im], None, scales=[scales], im_devs=[im_dev], net=net)

    def window_units(self, ims, picks=None, scales=None, im_devs=None, net=None):
        """The units ``picks`` = [(image index in ``ims``, unit index)] (default: every unit of every image, image-major)
        of several images in ONE slot: what a rank of the pyramid-sharded schedule runs of a window
        (pyramid.ShardedDetector) -- an image none of whose units is picked is not uploaded."""
        import torch
        net = net or self.net
        geos = []
        for k, im in enumerate(ims):
            sc = scales[k] if scales is not None and scales[k] is not None else pyramid_scales(im.shape)
            im_h, im_w = im.shape[:2]
            geo = []
            for s in sc:
                lh, lw, H, W = caffe.pyramid_level_shape(im_h, im_w, s, cfg.MAX_RESOLUTION)
                for flip in ([False, True] if cfg.TEST.FLIP else [False]):
                    geo.append((lh, lw, H, W, s, flip))
            geos.append(geo)
        if picks is None:
            picks = [(k, u) for k in range(len(ims)) for u in range(len(geos[k]))]
        devs = {}
        for k in sorted(set(k for k, _ in picks)):
            d = im_devs[k] if im_devs is not None else None
            if d is None:
                d = torch.from_numpy(np.ascontiguousarray(ims[k], dtype=np.uint8)).to("cuda", non_blocking=False)
            devs[k] = d
        total = sum(3 * geos[k][u][2] * geos[k][u][3] for k, u in picks)
        slot = self._slots[self._k]
        if slot is None or slot[0].numel() < total:
            slot = [torch.empty(max(total, 1), dtype=torch.float32, device="cuda"), None]
        slot[1] = devs  # keep the images alive while the kernels are in flight
        self._slots[self._k] = slot
        self._k = (self._k + 1) % len(self._slots)
        out, off = [], 0
        for k, u in picks:
            lh, lw, H, W, s, flip = geos[k][u]
            im_h, im_w = ims[k].shape[:2]
            ptr = slot[0].data_ptr() + 4 * off
            net.make_pyramid_level(devs[k].data_ptr(), im_h, im_w, s, flip, cfg.PIXEL_MEANS, ptr, H, W, lh, lw)
            out.append((ptr, H, W, lh, lw, s, flip))
            off += 3 * H * W
        return out


def lane_ranges(costs, n_lanes):
    """Split the unit sequence into ``n_lanes`` CONTIGUOUS ranges of roughly equal cost
    (contiguous so that concatenating the lanes' detection lists in lane order is the
    reference's unit order, test.py:141-158)."""
    n_lanes = max(1, min(n_lanes, len(costs)))
    total = float(sum(costs))
    ranges, start, acc = [], 0, 0.0
    for i, c in enumerate(costs):
        acc += c
        left_units = len(costs) - (i + 1)
        left_lanes = n_lanes - len(ranges) - 1
        if left_lanes > 0 and (acc >= total * (len(ranges) + 1) / n_lanes or left_units == left_lanes):
            ranges.append((start, i + 1))
            start = i + 1
    ranges.append((start, len(costs)))
    return ranges


GROUP_UNITS = 16   # units per grouped pass (csrc/conv_common.h MAX_GROUP = csrc/tail.hip TG)


class FusedDetector(object):
    """detect() with every per-unit step on the device (C ABI shf_detect_*), the units of an
    image spread over execution lanes (own HIP stream + activations, shared weights) so the
    latency-bound small pyramid levels overlap with the large ones."""

    def __init__(self, net, n_lanes=4, mode="streams", lane_sets=1):
        """mode "streams": units spread over lanes running on their own HIP streams;
        mode "group": the lanes only lend activation buffers and every MFMA conv layer runs as
        ONE grid over all units of the image (shf_detect_add_levels).
        ``lane_sets`` (group mode, submit / collect): 1 = the shipped pipeline -- consecutive images share one lane set and one
        in-order convolution stream, so a kernel's duration is its own; 2 = a MEASUREMENT mode (bench.py `overlapped_pipeline`):
        the two head lanes get a lane set and a stream each, consecutive images' convolutions overlap on the GPU and fill each
        other's partial last rounds -- the same kernels, the same bits, more images/s, but every kernel's duration then contains
        its neighbour's blocks, so no per-kernel roofline can be read off such a run."""
        self.net = net
        self.mode = mode
        self.lane_sets = 2 if int(lane_sets) >= 2 else 1
        self.lanes = [net] + [net.clone() for _ in range(max(1, n_lanes) - 1)]
        self._lanes_b = None
        self._xbuf = None

    # -- software pipeline over images (group mode): the box merging + read-back of image k overlaps
    #    the convolutions of image k+1.  Two head lanes own the image lists and streams; the member
    #    lanes only lend activation buffers, so image k+1 waits for image k's appends, not its merge.
    def next_head(self):
        """The head lane the next ``submit`` will enqueue on (e.g. to pre-process its image on the
        same stream: DevicePyramid.units(im, net=fd.next_head()))."""
        if not hasattr(self, "_heads"):
            self._heads = [self.net.clone(), self.net.clone()]
            if self.lane_sets == 2:
                # measurement mode: no predecessor, no shared convolution stream -- each head's whole pass on its own stream
                # over its own lane set (a lane is only ever re-used by the same head, two images later)
                self._lanes_b = [self.net.clone() for _ in self.lanes]
            else:
                # one lane set: image k + 1's convolutions queue behind image k's (both images' convolutions overlapped on two
                # lane sets measure +3 % images/s, but then no kernel's duration is its own: not the shipped pipeline)
                self._heads[0].set_predecessor(self._heads[1])
                self._heads[1].set_predecessor(self._heads[0])
                if os.environ.get("SHF_PIPE_SHARED_CONV_STREAM", "1") != "0":
                    # convolutions of consecutive images on one in-order stream, tails + merge on the heads' own
                    # high-priority streams: does not depend on how the runtime maps streams to hardware queues
                    for h in self._heads:
                        h.set_pipeline(True)
            self._turn = 0
            self._inflight = []
        return self._heads[self._turn]

    # -- fp16 range guard of the split-fp16 mode: a pass whose convolutions left the fp16 range fails in
    #    detect_finish with a distinct message (C ABI shf_net_range_fallbacks); the image is then redone on the
    #    exact fp32 kernels -- the reference computes in fp32 everywhere -- never a silent inf/NaN.
    range_fallbacks = 0

    @staticmethod
    def _is_range_error(e):
        return "split-fp16 range exceeded" in str(e)

    def _redo_fp32(self, units, thresh, on_device):
        self.range_fallbacks += 1
        logger.warning("split-fp16 range exceeded: image redone on the exact fp32 kernels")
        mode = self.net.conv_mode
        self.net.set_conv_mode("fp32")      # shared by every lane
        try:
            return self.detect(units, thresh, on_device=on_device)
        finally:
            self.net.set_conv_mode(mode)

    def submit(self, units, thresh=0.05, on_device=False):
        units = list(units)
        assert self.mode == "group"
        head = self.next_head()
        lanes = self._lanes_b if (self.lane_sets == 2 and self._turn == 1) else self.lanes
        while len(lanes) < len(units):
            lanes.append(self.net.clone())
        head.detect_begin()
        # no wait here: add_levels starts as soon as the previous image's logits kernels have consumed
        # the member lanes' feature maps and only awaits its appends before this image's own tails.
        # One grouped pass holds GROUP_UNITS units (one kernel-argument member table); a longer unit list runs as
        # several passes into the same image list, each on lanes of its own (a pass's tails run on the head's
        # stream beside the next pass's convolutions: they must not share tail workspaces).
        for a in range(0, len(units), GROUP_UNITS):
            b = min(a + GROUP_UNITS, len(units))
            head.detect_add_levels(lanes[a:b], units[a:b], thresh, on_device=on_device)
        head.record_event()
        self._inflight.append((head, units, thresh, on_device))
        self._turn = 1 - self._turn
        return head

    def collect(self):
        """Detections of the oldest submitted image (blocks until its merge is done)."""
        if getattr(self, "_done", None):
            return self._done.pop(0)
        head, units, thresh, on_device = self._inflight.pop(0)
        try:
            return [head.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)]
        except _lib.ShfError as e:
            if not self._is_range_error(e):
                raise
        # drain what is in flight (the redo reuses the lanes' buffers), then redo the flagged image(s) exactly
        rest = []
        while self._inflight:
            h, u, t, od = self._inflight.pop(0)
            try:
                rest.append([h.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)])
            except _lib.ShfError as e:
                if not self._is_range_error(e):
                    raise
                rest.append((u, t, od))
        out = self._redo_fp32(units, thresh, on_device)
        self._done = [r if isinstance(r, list) else self._redo_fp32(*r) for r in rest]
        return out

    def pending(self):
        return len(getattr(self, "_inflight", [])) + len(getattr(self, "_done", None) or [])

    def detect(self, units, thresh=0.05, on_device=False):
        """``units``: list of (data, H, W, im_h, im_w, scale, flip); data = host array or device pointer."""
        units = list(units)
        if self.mode == "group":
            # every pass of a longer unit list on lanes of ITS OWN, like submit(): a pass's tail phase 2 may run (pipelined
            # head) beside the next pass's convolutions and logits, and the two must not share tail workspaces
            while len(self.lanes) < len(units):
                self.lanes.append(self.net.clone())
            head = self.lanes[0]
            head.detect_begin()
            for a in range(0, len(units), GROUP_UNITS):
                b = min(a + GROUP_UNITS, len(units))
                head.detect_add_levels(self.lanes[a:b], units[a:b], thresh, on_device=on_device)
            try:
                return [head.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)]
            except _lib.ShfError as e:
                if not self._is_range_error(e):
                    raise
            return self._redo_fp32(units, thresh, on_device)
        ranges = lane_ranges([u[1] * u[2] for u in units], len(self.lanes))
        used = []
        # issue the most expensive ranges first
        order = sorted(range(len(ranges)), key=lambda k: -sum(u[1] * u[2] for u in units[ranges[k][0]:ranges[k][1]]))
        for k in range(len(ranges)):
            self.lanes[k].detect_begin()
        for k in order:
            a, b = ranges[k]
            for data, H, W, im_h, im_w, s, flip in units[a:b]:
                self.lanes[k].detect_add_level(data, H, W, im_h, im_w, s, flip, thresh, on_device=on_device)
            used.append(k)
        head = self.lanes[0]
        if len(ranges) > 1:
            import torch  # device staging buffer for the lane -> lane hand-off
            cap = sum(cfg.TEST.N_DETS_PER_MODULE for _ in units)
            if self._xbuf is None or self._xbuf.shape[0] < cap:
                self._xbuf = torch.empty((cap, 5), dtype=torch.float32, device="cuda")
            for k in range(1, len(ranges)):
                n = self.lanes[k].detect_export(self._xbuf.data_ptr(), cap)
                head.detect_import(self._xbuf.data_ptr(), min(n, cap))
        return [head.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)]


def detect_fused(net, units, thresh=0.05, on_device=False):
    """Single-lane form of FusedDetector.detect."""
    net.detect_begin()
    for data, H, W, im_h, im_w, s, flip in units:
        net.detect_add_level(data, H, W, im_h, im_w, s, flip, thresh, on_device=on_device)
    return [net.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)]


def fused_image_loop(net, paths, thresh=0.05, timers=None, progress=None, prefetch=None, stats=None, fd=None, dp=None):
    """The hot loop of inference_worker (lib/test.py:239-244: imread -> detect, image after image) on the device-resident
    path: image FILE -> decode -> upload -> pyramid on the device (DevicePyramid) -> grouped pass -> merge, two images in
    flight on the GPU (FusedDetector.submit / collect).  The decodes of the next ``prefetch`` images (default 2; env
    SHF_DECODE_PREFETCH, 0 = decode synchronously like the reference) run on reader threads while image i is being
    submitted and image i - 1 collected: a JPEG decode is 5-15 ms, the GPU's share of an image 11-15 ms, so a
    synchronous decode on the submitting thread would be exposed.  Returns the per-image (n, 5) detection arrays in
    order.  ``stats`` (a dict) receives per-image mean milliseconds: decode (on the reader threads), decode_wait / submit /
    collect_wait (on this thread)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    if timers is None:
        timers = {'detect': Timer(), 'misc': Timer()}
    # decode-ahead depth: SHF_DECODE_PREFETCH / ``prefetch`` fix it; otherwise it starts at 2 and is re-sized every few images
    # from what the loop itself measures -- decodes in flight needed = decode time / time per image of the loop, + 1 of
    # slack, within [2, 8] (a slow decoder -- large JPEGs, a busy host -- then gets more reader threads instead of
    # starving the GPU; stats["decode_prefetch"] reports where it ended)
    adaptive = prefetch is None and "SHF_DECODE_PREFETCH" not in os.environ
    if prefetch is None:
        prefetch = int(os.environ.get("SHF_DECODE_PREFETCH", "2"))
    prefetch = max(0, int(prefetch))
    max_prefetch = 8 if adaptive else prefetch
    n_units = len(cfg.TEST.SCALES) * (2 if cfg.TEST.FLIP else 1)
    fd = fd or FusedDetector(net, n_lanes=n_units, mode="group")
    dp = dp or DevicePyramid(net, n_slots=2)
    acc = {"decode": 0.0, "decode_wait": 0.0, "submit": 0.0, "collect_wait": 0.0}
    n_decoded = [0]
    t_loop0 = time.perf_counter()

    import threading
    acc_lock = threading.Lock()

    def read(path):
        t0 = time.perf_counter()
        im = _imread(path)
        with acc_lock:
            acc["decode"] += time.perf_counter() - t0
            n_decoded[0] += 1
        if im is None:
            raise IOError("cannot read image %s" % path)
        return im

    out = [None] * len(paths)
    queued = []
    # (one reader thread per decode in flight: a pool of eight mostly idle threads measured 8 % slower than a pool of two on
    # the bench's files -- GIL hand-overs --, so the pool is replaced when the depth grows, not over-provisioned)
    pool = ThreadPoolExecutor(max_workers=prefetch) if prefetch and len(paths) > 1 else None
    retired = []
    ahead, nxt_i = [], 0     # futures of the images after the current one, in order
    try:
        for i in list(range(len(paths))) + [None]:
            if i is not None:
                timers['misc'].tic()
                t0 = time.perf_counter()
                if pool:
                    if adaptive and i >= 4 and i % 4 == 0 and n_decoded[0] > 0:
                        with acc_lock:
                            dec = acc["decode"] / n_decoded[0]
                        per_image = (time.perf_counter() - t_loop0) / i
                        want = int(min(max_prefetch, max(2, np.ceil(dec / max(per_image, 1e-6)) + 1)))
                        if want > prefetch:        # grow only: a larger pool for the decodes submitted from here on
                            retired.append(pool)
                            pool = ThreadPoolExecutor(max_workers=want)
                            prefetch = want
                    while nxt_i < len(paths) and nxt_i <= i + prefetch:
                        ahead.append(pool.submit(read, paths[nxt_i]))
                        nxt_i += 1
                    im = ahead.pop(0).result()
                else:
                    im = read(paths[i])
                t1 = time.perf_counter()
                acc["decode_wait"] += t1 - t0
                timers['misc'].toc()
                timers['detect'].tic()
                fd.submit(dp.units(im, net=fd.next_head()), thresh, on_device=True)
                acc["submit"] += time.perf_counter() - t1
                queued.append(i)
            if queued and (i is None or fd.pending() > 1):
                j = queued.pop(0)
                t0 = time.perf_counter()
                out[j] = fd.collect()[0]
                acc["collect_wait"] += time.perf_counter() - t0
                timers['detect'].toc()
                if progress:
                    progress(j)
        while queued:
            j = queued.pop(0)
            out[j] = fd.collect()[0]
            if progress:
                progress(j)
    except BaseException:
        # a reader or a submit failed with images still in flight: drain the detector (a caller-provided ``fd`` would
        # otherwise hand the NEXT user detections of these images) and drop the decodes that have not started
        for f in ahead:
            f.cancel()
        while fd.pending():
            try:
                fd.collect()
            except Exception:
                pass
        raise
    finally:
        for p_ in retired + ([pool] if pool else []):
            p_.shutdown(wait=True)
    if stats is not None and paths:
        stats.update({k + "_ms": 1000.0 * v / len(paths) for k, v in acc.items()})
        stats["decode_threads"] = prefetch if pool else 0
        stats["decode_prefetch"] = prefetch if pool else 0
        stats["decode_prefetch_adaptive"] = bool(adaptive and pool)
        stats["decode_over_step"] = (acc["decode"] / max(1, n_decoded[0])) / max((time.perf_counter() - t_loop0) / max(1, len(paths)), 1e-9)
    return out


def inference_worker(rank, imdb, target_test, start, end, thresh, result_queue=None, fused=None):
    """lib/test.py:220-253.  ``fused`` (default: env SHF_FUSED_DETECT, on): pyramid images go through
    the device-resident path (DevicePyramid -> FusedDetector.submit/collect, two images in flight)
    instead of one Net.forward() per unit; same detections, the blobs never visit the host."""
    cfg.GPU_ID = cfg.TEST.GPU_ID[rank]
    caffe.set_mode_gpu()
    caffe.set_device(cfg.GPU_ID)
    if not str(cfg.TEST.MODEL):
        # Caffe aborts in CopyTrainedLayersFrom on an empty path (net.cpp:733-768); an all-zero net would
        # otherwise write detection files without a word
        raise IOError("TEST.MODEL is empty: pass --amend TEST.MODEL <file>.caffemodel")
    net = caffe.Net(str(target_test), str(cfg.TEST.MODEL), caffe.TEST)
    if "SHF_CONV_MODE" not in os.environ:
        # split-fp16 matrix-core arithmetic (fp32-class, same parity bars, 3x the throughput of the exact
        # fp32 MFMA path); SHF_CONV_MODE=0 keeps the library default (exact fp32)
        net.set_conv_mode("f16x3")

    timers = {'detect': Timer(), 'misc': Timer()}
    pyramid = True if len(cfg.TEST.SCALES) > 1 else False
    dets = [[[] for _ in range(start, end)] for _ in range(imdb.num_classes)]
    if fused is None:
        fused = os.environ.get("SHF_FUSED_DETECT", "1") != "0"
    fused = fused and pyramid and len(cfg.TEST.LEVEL) == 0

    def progress(i):
        if rank == 0:
            print('\r{:02d}% detect-time: {:.3f}s, misc-time:{:.3f}s, remain-time: {:.3f}s'.format(
                int(100 * (i + 1 - start) / (end - start)), timers['detect'].average_time,
                timers['misc'].average_time,
                (end - i - 1) * (timers['detect'].average_time + timers['misc'].average_time)), end='')

    if fused:
        paths = [imdb.image_path_at(i) for i in range(start, end)]
        dets[1] = fused_image_loop(net, paths, thresh, timers=timers, progress=lambda j: progress(start + j))
    else:
        for i in range(start, end):
            im_path = imdb.image_path_at(i)
            dets_, _ = detect(net, im_path, thresh, timers=timers, pyramid=pyramid)
            for c in range(imdb.num_classes - 1):
                dets[c + 1][i - start] = dets_[c]
            progress(i)
    if result_queue:
        result_queue.put((rank, dets))
        return
    return dets


def _worker_entry(cfg_state, rank, imdb, target_test, start, end, thresh, result_queue):
    cfg.clear()
    cfg.update(cfg_state)
    inference_worker(rank, imdb, target_test, start, end, thresh, result_queue)


def _gather_results(result_queue, procs, poll_seconds=0.5, grace_polls=20):
    """One (rank, dets) per worker from the queue -- lib/test.py:339 is a bare ``result_queue.get()`` per worker and
    waits forever for a worker that died (out of memory, unreadable image); here a worker that has exited without
    delivering ends the run with an exception naming its rank, and the surviving workers are stopped."""
    import queue as _queue
    got, quiet = {}, {}
    while len(got) < len(procs):
        try:
            rank, d = result_queue.get(timeout=poll_seconds)
            got[rank] = (rank, d)
            continue
        except _queue.Empty:
            pass
        for rank, p in enumerate(procs):
            if rank in got or p.exitcode is None:
                continue
            # (an exit code 0 with the result still in the pipe is possible for a moment: allow a few more polls)
            quiet[rank] = quiet.get(rank, 0) + 1
            if p.exitcode != 0 or quiet[rank] > grace_polls:
                for q in procs:
                    if q.exitcode is None:
                        q.terminate()
                raise RuntimeError("inference worker %d (GPU %s) exited with code %s before delivering its detections"
                                   % (rank, cfg.TEST.GPU_ID[rank] if rank < len(cfg.TEST.GPU_ID) else "?", p.exitcode))
    return [got[r] for r in sorted(got)]


def dist_env():
    """(rank, world, local_rank) of a process started by torch.distributed.run / torchrun; (0, 1, 0) otherwise."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def pyramid_sharded_inference(imdb, target_test, thresh=0.05, strict=False, progress_every=0):
    """``TEST.SHARD = "pyramid"`` (or "pyramid_strict"): the north star's split of the work -- the PYRAMID of every image is
    sharded over the ranks of a torch.distributed job (one process per GPU, started by ``torch.distributed.run``), not the
    image list (lib/test.py:327-344, which ``TEST.SHARD = "images"`` keeps).  Windows of ``world`` images go through
    pyramid.ShardedDetector: every rank decodes the window's images, builds ITS units on the device (DevicePyramid),
    runs them as grouped passes, and the > thresh rows travel to the image's owner rank in ONE all_to_all per window
    (RCCL; ``SHF_DIST_BACKEND=gloo`` for validation with several ranks on one GPU), which merges them.  At the end the
    owners' results are all-gathered so that every rank returns the full ``dets[class][image]`` lists.

    Device of rank r: ``TEST.GPU_ID[r]`` when the list has one entry per rank (``[0,0]``: two ranks on one GPU, gloo),
    LOCAL_RANK otherwise."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from .pyramid import ShardedDetector
    rank, world, local = dist_env()
    ids = cfg.TEST.GPU_ID if not isinstance(cfg.TEST.GPU_ID, int) else [cfg.TEST.GPU_ID]
    dev_id = int(ids[rank]) if len(ids) == world else local
    if len(cfg.TEST.SCALES) <= 1 or len(cfg.TEST.LEVEL) > 0:
        raise ValueError("TEST.SHARD pyramid needs the pyramid path (several TEST.SCALES, no TEST.LEVEL subset)")
    if not str(cfg.TEST.MODEL):
        raise IOError("TEST.MODEL is empty: pass --amend TEST.MODEL <file>.caffemodel")
    torch.cuda.set_device(dev_id)
    dist, own_group, finished = None, False, False
    if world > 1:
        import datetime
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get("SHF_DIST_BACKEND", "nccl")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = dict(rank=rank, world_size=world,
                      timeout=datetime.timedelta(seconds=float(os.environ.get("SHF_DIST_TIMEOUT", "600"))))
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_id), **kw)
            else:
                dist.init_process_group(backend, **kw)
            own_group = True
    try:
        cfg.GPU_ID = dev_id
        caffe.set_mode_gpu()
        caffe.set_device(dev_id)
        net = caffe.Net(str(target_test), str(cfg.TEST.MODEL), caffe.TEST)
        if "SHF_CONV_MODE" not in os.environ:
            net.set_conv_mode("f16x3")
        n_flip = 2 if cfg.TEST.FLIP else 1
        n_units = len(cfg.TEST.SCALES) * n_flip
        sd = ShardedDetector(net, rank, world, n_units, units_per_level=n_flip, shard="strict" if strict else "window",
                             thresh=thresh, device=torch.device("cuda", dev_id))
        dp = DevicePyramid(net, n_slots=2)   # window k + 1 is built while window k is in flight; k - 1 is finished
        n = len(imdb)
        n_windows = (n + world - 1) // world
        owned = {}

        def window_plan(w):
            base = w * world
            n_valid = min(world, n - base)
            picks = [(i, u) for (i, u) in sd.mine if i < n_valid]
            need = sorted(set(i for i, _ in picks))
            return base, n_valid, picks, need

        def read(i):
            im = _imread(imdb.image_path_at(i))
            if im is None:
                raise IOError("cannot read image %s" % imdb.image_path_at(i))
            return im

        pool = ThreadPoolExecutor(max_workers=max(1, min(4, world)))
        try:
            ahead = None
            prev_base = None
            for w in range(n_windows):
                base, n_valid, picks, need = window_plan(w)
                futs = ahead if ahead is not None else {i: pool.submit(read, base + i) for i in need}
                if w + 1 < n_windows:                       # the next window's decodes run under this window's GPU work
                    nb, _, _, nneed = window_plan(w + 1)
                    ahead = {i: pool.submit(read, nb + i) for i in nneed}
                else:
                    ahead = None
                ims = [futs[i].result() for i in need]
                remap = {i: k for k, i in enumerate(need)}
                ls = sd.lane_sets[sd._k & 1]                # the pass's head: its pre-processing runs in front of its convolutions
                head = ls[0] if ls else net                 # (a rank without a share -- strict sharding, more ranks than levels)
                units = dp.window_units(ims, [(remap[i], u) for (i, u) in picks], net=head)
                done = sd.submit(units, picks=picks, n_valid=n_valid)
                for i, d in done.items():
                    owned[prev_base + i] = d
                prev_base = base
                if progress_every and rank == 0 and (w + 1) % progress_every == 0:
                    print('\r{:02d}% of {} windows'.format(int(100 * (w + 1) / n_windows), n_windows), end='')
            for i, d in sd.flush().items():
                owned[prev_base + i] = d
        finally:
            pool.shutdown(wait=True)
        sd.sync()
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, owned)
            owned = {}
            for p_ in parts:
                owned.update(p_)
        assert sorted(owned) == list(range(n)), "Detection result compromised"
        dets = [[[] for _ in range(n)] for _ in range(imdb.num_classes)]
        for i in range(n):
            dets[1][i] = owned[i]
        finished = True
        return dets
    finally:
        if own_group:
            if finished:                  # (after a failure the other ranks are not at this barrier: let the group's timeout end them)
                dist.barrier()
            dist.destroy_process_group()


def test_net(imdb, output_dir, target_test, thresh=0.05, no_cache=False, step=0):
    logger.info('Evaluating {} on {}'.format(cfg.NAME, imdb.name))
    run_inference = True
    dets = None
    if not no_cache:
        det_file = os.path.join(output_dir, 'detections.pkl')
        if os.path.exists(det_file):
            try:
                with open(det_file, 'rb') as f:
                    dets = pickle.load(f)
                    run_inference = False
                    logger.info('Loading detections from cache: {}'.format(det_file))
            except Exception:
                logger.warning('Could not load the cached detections file, detecting from scratch!')

    shard = str(cfg.TEST.get("SHARD", "images"))
    if shard not in ("images", "pyramid", "pyramid_strict"):
        raise ValueError("TEST.SHARD must be 'images' (the reference's image ranges), 'pyramid' or 'pyramid_strict'")
    if run_inference and shard != "images":
        # the north star's split: every image's pyramid sharded over the ranks of a torch.distributed job
        dets = pyramid_sharded_inference(imdb, target_test, thresh, strict=(shard == "pyramid_strict"))
        assert len(dets[0]) == len(imdb), "Detection result compromised"
        if not no_cache and dist_env()[0] == 0:
            with open(os.path.join(output_dir, 'detections.pkl'), 'wb') as f:
                pickle.dump(dets, f, pickle.HIGHEST_PROTOCOL)
    elif run_inference:
        if isinstance(cfg.TEST.GPU_ID, int):
            cfg.TEST.GPU_ID = [cfg.TEST.GPU_ID]
        assert len(cfg.TEST.GPU_ID) >= 1, "You must specify at least one GPU"
        if len(cfg.TEST.GPU_ID) == 1:
            dets = inference_worker(0, imdb, target_test, 0, len(imdb), thresh)
        else:
            # one process per GPU (spawn, not fork: a HIP context does not survive fork)
            import multiprocessing as mp
            ctx = mp.get_context("spawn")
            result_queue = ctx.Queue()
            procs = []
            len_per_gpu = int(np.ceil(1. * len(imdb) / len(cfg.TEST.GPU_ID)))
            for rank in range(len(cfg.TEST.GPU_ID)):
                p = ctx.Process(target=_worker_entry,
                                args=(dict(cfg), rank, imdb, target_test, len_per_gpu * rank,
                                      min(len_per_gpu * (rank + 1), len(imdb)), thresh, result_queue))
                p.daemon = True
                p.start()
                procs.append(p)
            dets = _gather_results(result_queue, procs)
            for p in procs:
                p.join()
            dets = [det[1] for det in sorted(dets, key=lambda x: x[0])]
            dets = [[_ for det in dets for _ in det[i]] for i in range(imdb.num_classes)]
        assert len(dets[0]) == len(imdb), "Detection result compromised"
        det_file = os.path.join(output_dir, 'detections.pkl')
        if not no_cache:
            with open(det_file, 'wb') as f:
                pickle.dump(dets, f, pickle.HIGHEST_PROTOCOL)

    if shard != "images" and dist_env()[0] != 0:
        return dets                           # every rank holds the full lists; rank 0 writes and evaluates
    logger.info('Evaluating detections')
    result = imdb.evaluate_detections(all_boxes=dets, output_dir=output_dir, method_name=cfg.NAME, step=step)
    logger.info(result)
    logger.info('All Done!')
    return dets
